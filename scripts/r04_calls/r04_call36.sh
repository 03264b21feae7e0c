#!/bin/bash
# Round 4, call 36: where the conv2 weight gradient runs now that all three front-end backward kernels own their CUs alone
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do for w in side after_dgrad last; do
  echo "NST_CONV2_WGRAD_AT=$w  $(NST_CONV2_WGRAD_AT=$w step) ms/step"
done; done | tee $O/c36_ab_step.log
