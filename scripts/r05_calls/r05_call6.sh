#!/bin/bash
# Round 5, call 6: A/B of the recomputed ReLU gate in the front end's LayerNorm backward
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do for v in 0 1; do
  echo "NST_LN_REGATE=$v  $(NST_LN_REGATE=$v step) ms/step"
done; done | tee $O/c6_ab_regate.log
