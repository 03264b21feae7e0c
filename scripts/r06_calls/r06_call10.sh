#!/bin/bash
# RECORD of a round-6 experiment: the NST_* switch(es) this script sets existed only in the working tree of that experiment
# (removed with it; the library now warns about them).  Kept for the log under profiles/r06_history/; it does not re-run.
echo "$0: record of a removed experiment (see the header); not runnable against this tree" >&2; exit 1
# Round 6, call 10: ablation of the whole-row products with cold caches (NST_ROWGEMM_DBG: 1 = no K loop, 2 = no row phase, 4 = no x prefetch)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
for dbg in 0 1 2 3 4 6 7; do
  NST_ROWGEMM_DBG=$dbg timeout 400 python scripts/rowgemm_bench.py r06_c10_dbg$dbg --cold 2>/dev/null | tail -45 > $O/c10_cold_dbg$dbg.json
done
python - <<'PY'
import json,glob
r={}
for d_ in (0,1,2,3,4,6,7):
    try: r[f"dbg{d_}"]=json.load(open(f"gpurun_out/r06/c10_cold_dbg{d_}.json"))
    except Exception as e: print(d_, e)
keys=[k for k in next(iter(r.values())) if k.endswith("fused_us")]
print("%-36s"%"case (cold)"+"".join("%9s"%c for c in r))
for k in keys: print("%-36s"%k+"".join("%9.2f"%r[c].get(k,float('nan')) for c in r))
PY
