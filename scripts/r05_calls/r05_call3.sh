#!/bin/bash
# Round 5, call 3: row-panel K = 256 GEMM: parity, stand-alone timing against the stream kernel, step A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "row_panel or rowdot" --tb=short 2>&1 | grep -v "amdgpu.ids" > $O/c3_pytest.log
tail -25 $O/c3_pytest.log
for v in 0 1; do NST_GEMM_RP=$v timeout 300 python scripts/gemm_rp_bench.py 2>/dev/null | tail -n 1 > $O/c3_rp_bench_$v.json; done
python - <<'PY'
import json
a=json.load(open("gpurun_out/r05/c3_rp_bench_0.json")); b=json.load(open("gpurun_out/r05/c3_rp_bench_1.json"))
for k in a:
    if k!="NST_GEMM_RP": print(f"{k:44s} stream {a[k]:8.2f} us   row-panel {b[k]:8.2f} us")
PY
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do for v in 0 1; do
  echo "NST_GEMM_RP=$v  $(NST_GEMM_RP=$v step) ms/step"
done; done | tee $O/c3_ab_rp.log
