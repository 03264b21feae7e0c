#!/bin/bash
# Round 5, call 17: layer-1 forward on the matrix cores, taps two groups ahead, weights through LDS: waves per SIMD 4 / 3 / 2
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "conv1 or frontend" 2>&1 | grep -v "amdgpu.ids" > $O/c17_pytest.log
tail -3 $O/c17_pytest.log | cut -c1-300
for occ in 4 3 2 4 3; do
NST_C1F_OCC=$occ timeout 300 python scripts/conv_bench.py > $O/c17_conv_bench_occ$occ.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c17_conv_bench_occ$occ.json')); print($occ, {k: round(v['us'],1) for k,v in d.items() if 'conv1' in k})"
done
