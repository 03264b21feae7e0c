#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 900 python -m pytest tests/test_gpu_ffn.py -x -q 2>&1 | tail -5
for r in 1 2; do timeout 300 python scripts/ffn_bench.py --rows 28800 --iters 30 2>/dev/null | python -c "
import sys,json
l=sys.stdin.readline(); r=json.loads(l[l.index('{'):])
print({k:round(v['fused_us'],1) for k,v in r.items() if 'fused_us' in v})"; done
bash scripts/ab_env.sh 2 NST_FFN_GATE_BITS 0 1 -- --steps 20 --warmup 5
