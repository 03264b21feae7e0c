#!/bin/bash
# Round 5, call 18: layer-1 forward on the matrix cores: ablations (1 = no stores, 2 = no tap gather, 4 = rows through LDS)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
NST_C1F_OCC=2 NST_C1F_DBG=4 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "conv1" 2>&1 | grep -v "amdgpu.ids" > $O/c18_pytest.log
tail -3 $O/c18_pytest.log | cut -c1-300
for dbg in 0 1 2 3 4 6 0 4; do
NST_C1F_OCC=2 NST_C1F_DBG=$dbg timeout 300 python scripts/conv_bench.py 2>/dev/null | grep conv1_fwd | sed "s/^/dbg $dbg /"
done
