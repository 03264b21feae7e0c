#!/bin/bash
# Round 5, call 21: role-split stream GEMM (four waves multiply + store, two move operands): parity, stand-alone A/B, step A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
NST_GEMM_V3S=4096 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "gemm" 2>&1 | grep -v "amdgpu.ids" > $O/c21_pytest.log
tail -3 $O/c21_pytest.log | cut -c1-300
for v in 0 4096 0 4096; do
NST_GEMM_V3S=$v timeout 300 python scripts/r05_experiments/gemm_v3s_bench.py 2>/dev/null | tail -1 > $O/c21_gemm_v3s_$v.json; cat $O/c21_gemm_v3s_$v.json | cut -c1-1800
done
NST_GEMM_V3S=256 scripts/gpu_profile2.sh r05c21a 8 > $O/c21_profile_a.log 2>&1
tail -1 gpurun_out/r05c21a_prof_bench.json | python -c 'import sys,json; print("v3s<=256 step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
NST_GEMM_V3S=4096 scripts/gpu_profile2.sh r05c21b 8 > $O/c21_profile_b.log 2>&1
tail -1 gpurun_out/r05c21b_prof_bench.json | python -c 'import sys,json; print("v3s all step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
grep -E "dense_gemm|TOTAL" gpurun_out/r05c21b_kernel_stats.csv | awk -F, '{print substr($1,1,75),$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
