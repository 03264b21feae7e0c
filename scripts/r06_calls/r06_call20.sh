#!/bin/bash
# RECORD of a round-6 experiment: the NST_* switch(es) this script sets existed only in the working tree of that experiment
# (removed with it; the library now warns about them).  Kept for the log under profiles/r06_history/; it does not re-run.
echo "$0: record of a removed experiment (see the header); not runnable against this tree" >&2; exit 1
# call 20: the grouped weight-gradient launch on the weight-gradient stream behind the encoder (temporary NST_WGRAD_GROUP_SIDE)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  echo "end $(one)   side $(NST_WGRAD_GROUP_SIDE=1 one)"
done | tee gpurun_out/r06/c20_group_side.log
