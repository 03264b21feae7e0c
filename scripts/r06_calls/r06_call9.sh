#!/bin/bash
# Round 6, call 9: the whole-row products with COLD caches (a 768 MB fill in front of every launch): what the forms cost inside
# the step, where operands come from HBM
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
for cfg in "" "128,2" "64,3" "32,2"; do
  NST_ROWGEMM_CFG=$cfg timeout 400 python scripts/rowgemm_bench.py r06_c9_${cfg/,/_} --cold 2>/dev/null | tail -45 > $O/c9_cold_${cfg/,/_}.json
done
python - <<'PY'
import json,glob
r={}
for f in sorted(glob.glob("gpurun_out/r06/c9_cold_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, e); continue
    r[d.get("NST_ROWGEMM_CFG") or "default"]=d
keys=[k for k in next(iter(r.values())) if k.endswith("_us")]
print("%-36s"%"case (cold)"+"".join("%10s"%c for c in r))
for k in keys: print("%-36s"%k+"".join("%10.2f"%r[c].get(k,float('nan')) for c in r))
PY
