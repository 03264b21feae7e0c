#!/bin/bash
# Round 6, call 6: the eight-wave 128-row form of the whole-row products (one workgroup per CU, three stages) for the encoder's
# row count: parity with the form forced on every call, timings, step A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
NST_ROWGEMM_CFG=128,3 timeout 900 python -m pytest tests/test_gpu_rowgemm.py -q -m gpu --tb=short 2>&1 | tail -5 | tee $O/c6_pytest_rowgemm_128.log
for cfg in "" "128,2"; do
  NST_ROWGEMM_CFG=$cfg timeout 300 python scripts/rowgemm_bench.py r06_c6_${cfg/,/_} 2>/dev/null | tail -45 > $O/c6_bench_${cfg/,/_}.json
done
python - <<'PY'
import json,glob
r={}
for f in sorted(glob.glob("gpurun_out/r06/c6_bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, e); continue
    r[d.get("NST_ROWGEMM_CFG") or "default"]=d
keys=[k for k in next(iter(r.values())) if k.endswith("_us")]
print("%-36s"%"case"+"".join("%10s"%c for c in r))
for k in keys: print("%-36s"%k+"".join("%10.2f"%r[c].get(k,float('nan')) for c in r))
PY
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do for v in "" "128,2"; do
  echo "NST_ROWGEMM_CFG=$v  $(NST_ROWGEMM_CFG=$v step) ms/step"
done; done | tee $O/c6_ab_rows.log
NST_ROWGEMM_CFG=128,2 scripts/gpu_profile2.sh r06c6_graph 8 > $O/c6_profile.log 2>&1; tail -2 $O/c6_profile.log
