#!/bin/bash
# Round 5, call 24: layer-1 backward with the ReLU gate from a clamped packed FMA
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "conv1 or frontend" 2>&1 | grep -v "amdgpu.ids" > $O/c24_pytest.log
tail -3 $O/c24_pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -x 2>&1 | grep -v "amdgpu.ids" > $O/c24_pytest_model.log
tail -3 $O/c24_pytest_model.log | cut -c1-300
for i in 1 2; do timeout 300 python scripts/conv_bench.py 2>/dev/null | grep conv1; done
scripts/gpu_profile2.sh r05c24 8 > $O/c24_profile.log 2>&1
grep -E "conv1|TOTAL" gpurun_out/r05c24_kernel_stats.csv | awk -F, '{print substr($1,1,40),$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
tail -1 gpurun_out/r05c24_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
