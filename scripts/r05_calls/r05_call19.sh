#!/bin/bash
# Round 5, call 19: layer-1 forward on the matrix cores, taps gathered a block of four groups at a time, 1-2 blocks ahead
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
NST_C1F_OCC=2 NST_C1F_DBG=0 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "conv1" 2>&1 | grep -v "amdgpu.ids" > $O/c19_pytest.log
tail -3 $O/c19_pytest.log | cut -c1-300
NST_C1F_OCC=2 NST_C1F_DBG=4 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "conv1" 2>&1 | grep -v "amdgpu.ids" | tail -2
for dbg in 0 1 2 3 4 0 4; do
NST_C1F_OCC=2 NST_C1F_DBG=$dbg timeout 300 python scripts/conv_bench.py 2>/dev/null | grep conv1_fwd | sed "s/^/dbg $dbg /"
done
NST_C1F_OCC=2 NST_C1F_DBG=4 scripts/gpu_profile2.sh r05c19 8 > $O/c19_profile.log 2>&1
grep -E "conv1|TOTAL" gpurun_out/r05c19_kernel_stats.csv | awk -F, '{print substr($1,1,40),$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
tail -1 gpurun_out/r05c19_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
