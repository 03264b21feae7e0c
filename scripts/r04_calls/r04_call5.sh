#!/bin/bash
# Round 4, call 5: which products join the grouped launch (all / encoder + decoder / encoder only)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
for r in 1 2; do
  for cfg in "NST_WGRAD_GROUP_SET=all" "NST_WGRAD_GROUP_SET=enc+dec" "NST_WGRAD_GROUP_MIN_ROWS=20000" "NST_WGRAD_GROUP=0"; do
    ms=$(env $cfg timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>$O/c5_bench_err.log | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
    echo "$cfg  $ms ms/step"
  done
done | tee $O/c5_ab_step.log
