#!/bin/bash
# RECORD of a round-6 experiment: the NST_* switch(es) this script sets existed only in the working tree of that experiment
# (removed with it; the library now warns about them).  Kept for the log under profiles/r06_history/; it does not re-run.
echo "$0: record of a removed experiment (see the header); not runnable against this tree" >&2; exit 1
# Round 6, call 12: the two-ring form of the whole-row products (A and weight streams in separate rings / waves): parity with the
# form forced, cold-cache timings against the first form, step A/B; the split feed-forward pair's tests again
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_ffn.py -q -m gpu --tb=short -k "layernorm" 2>&1 | tail -5 | tee $O/c12_pytest_ffn.log
NST_ROWGEMM_V2=1 timeout 900 python -m pytest tests/test_gpu_rowgemm.py -x -q -m gpu --tb=short 2>&1 | tail -12 | tee $O/c12_pytest_rowgemm_v2.log
for v in 0 1; do
  NST_ROWGEMM_V2=$v timeout 400 python scripts/rowgemm_bench.py r06_c12_v$v --cold 2>/dev/null | tail -45 > $O/c12_cold_v$v.json
  NST_ROWGEMM_V2=$v NST_ROWGEMM_DBG=6 timeout 400 python scripts/rowgemm_bench.py r06_c12_v${v}_dbg6 --cold 2>/dev/null | tail -45 > $O/c12_cold_v${v}_dbg6.json
done
python - <<'PY'
import json
r={}
for n in ("v0","v1","v0_dbg6","v1_dbg6"):
    try: r[n]=json.load(open(f"gpurun_out/r06/c12_cold_{n}.json"))
    except Exception as e: print(n, e)
keys=[k for k in next(iter(r.values())) if k.endswith("fused_us") or "rows_us" in k]
print("%-36s"%"case (cold)"+"".join("%10s"%c for c in r))
for k in keys: print("%-36s"%k+"".join("%10.2f"%r[c].get(k,float('nan')) for c in r))
PY
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do for v in 0 1; do
  echo "NST_ROWGEMM_V2=$v  $(NST_ROWGEMM_V2=$v step) ms/step"
done; done | tee $O/c12_ab_v2.log
