#!/bin/bash
# A/B of one environment switch on the benchmark step, interleaved rounds (single runs move +-1 % with the box's clocks):
#   scripts/ab_env.sh ROUNDS VAR value1 value2 ... [-- extra bench.py arguments]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROUNDS=$1; VAR=$2; shift 2
VALS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq 1 "$ROUNDS"); do
  for v in "${VALS[@]}"; do
    ms=$(env "$VAR=$v" python bench.py --no-cpu-baseline --roofline-steps 0 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
    echo "$VAR=$v  $ms ms/step"
  done
done
