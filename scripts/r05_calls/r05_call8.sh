#!/bin/bash
# Round 5, call 8: small-launch GEMM with the whole reduction in flight (dense_gemm_kernel_v2d): parity, stand-alone timing, step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" --tb=short 2>&1 | grep -v "amdgpu.ids" > $O/c8_pytest.log
tail -12 $O/c8_pytest.log
for v in 0 1; do NST_GEMM_V2D=$v timeout 300 python scripts/gemm_small_bench.py 2>/dev/null | tail -n 1 > $O/c8_small_bench_$v.json; done
python - <<'PY'
import json
a=json.load(open("gpurun_out/r05/c8_small_bench_0.json")); b=json.load(open("gpurun_out/r05/c8_small_bench_1.json"))
for k in a:
    if k!="NST_GEMM_V2D": print(f"{k:40s} stream {a[k]:8.2f} us   whole-K {b[k]:8.2f} us")
PY
for v in 0 1; do
  NST_GEMM_V2D=$v scripts/gpu_profile2.sh r05c8_v2d$v 8 > $O/c8_profile_v2d$v.log 2>&1
  grep -E "TOTAL" gpurun_out/r05c8_v2d${v}_kernel_stats.csv | cut -c1-100
  tail -1 gpurun_out/r05c8_v2d${v}_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
done
