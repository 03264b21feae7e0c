#!/bin/bash
# Round 5, call 20: layer-1 forward on the matrix cores, final form (mean from a sum fragment, packed ReLU, rows through LDS)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "conv1 or conv2_kernels_at or frontend" 2>&1 | grep -v "amdgpu.ids" > $O/c20_pytest.log
tail -3 $O/c20_pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -x 2>&1 | grep -v "amdgpu.ids" > $O/c20_pytest_model.log
tail -3 $O/c20_pytest_model.log | cut -c1-300
for i in 1 2; do timeout 300 python scripts/conv_bench.py 2>/dev/null | grep conv1_fwd; done
scripts/gpu_profile2.sh r05c20 8 > $O/c20_profile.log 2>&1
grep -E "conv1|TOTAL" gpurun_out/r05c20_kernel_stats.csv | awk -F, '{print substr($1,1,40),$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
tail -1 gpurun_out/r05c20_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
