#!/bin/bash
# Round 5, call 16: layer-1 forward on the matrix cores: parity, stand-alone timing, step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "conv1 or conv2_kernels_at or frontend" 2>&1 | grep -v "amdgpu.ids" > $O/c16_pytest.log
tail -12 $O/c16_pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -x 2>&1 | grep -v "amdgpu.ids" > $O/c16_pytest_model.log
tail -5 $O/c16_pytest_model.log | cut -c1-300
timeout 300 python scripts/conv_bench.py > $O/c16_conv_bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c16_conv_bench.json')); print({k: round(v['us'],1) for k,v in d.items()})"
scripts/gpu_profile2.sh r05c16 8 > $O/c16_profile.log 2>&1
grep -E "conv1|TOTAL" gpurun_out/r05c16_kernel_stats.csv | cut -c1-140
tail -1 gpurun_out/r05c16_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
