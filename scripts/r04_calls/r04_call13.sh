#!/bin/bash
# Round 4, call 13: timing ablations of attn_bwd_head8_kernel (results wrong on purpose)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
for m in 0 1 2 4 8 16 32 3 7 15 31 63; do
  NST_ATTN_DBG=$m timeout 300 python scripts/attn_bench.py 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dbg $m', {k: round(v, 1) for k, v in d.items() if k.endswith('bwd_us')})"
done | tee gpurun_out/r04/c13_attn_ablation.log
