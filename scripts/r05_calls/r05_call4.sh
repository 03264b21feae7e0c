#!/bin/bash
# Round 5, call 4: timing ablations of the row-panel kernel (what bounds a K step?)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
for d in 0 1 2 4 6 7 8 15; do NST_RP_DBG=$d timeout 300 python scripts/gemm_rp_bench.py 2>/dev/null | tail -n 1 > $O/c4_rp_dbg_$d.json; done
python - <<'PY'
import json
r={d: json.load(open(f"gpurun_out/r05/c4_rp_dbg_{d}.json")) for d in (0,1,2,4,6,7,8,15)}
print("%-44s" % "dbg: 1 no DMA, 2 no MFMA, 4 no frag reads, 8 no epilogue", *["%7d" % d for d in r])
for k in r[0]:
    if k!="NST_GEMM_RP": print("%-44s" % k, *["%7.2f" % r[d][k] for d in r])
PY
