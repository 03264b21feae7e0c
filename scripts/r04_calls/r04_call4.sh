#!/bin/bash
# Round 4, call 4: where the grouped weight-gradient launches go (end / stack / side)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
for r in 1 2; do
  for cfg in "NST_WGRAD_GROUP=0" "NST_WGRAD_GROUP_AT=end" "NST_WGRAD_GROUP_AT=stack" "NST_WGRAD_GROUP_AT=side"; do
    ms=$(env $cfg timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>$O/c4_bench_err.log | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
    echo "$cfg  $ms ms/step"
  done
done | tee $O/c4_ab_step.log
tail -5 $O/c4_bench_err.log
NST_WGRAD_GROUP_AT=side timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --eager 2>/dev/null | tail -n 1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("eager side", d["ms_per_step"], d.get("host_issue_ms_per_step"))
for n, f in d["roofline_families"].items(): print("  ", n, round(f["ms_per_step"], 3), round(f.get("frac", 0), 4), f["launches_per_step"])
'
