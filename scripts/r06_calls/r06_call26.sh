#!/bin/bash
# RECORD of a round-6 experiment: the NST_* switch(es) this script sets existed only in the working tree of that experiment
# (removed with it; the library now warns about them).  Kept for the log under profiles/r06_history/; it does not re-run.
echo "$0: record of a removed experiment (see the header); not runnable against this tree" >&2; exit 1
# call 26: how many weight-gradient calls make a layer boundary cut the captured step (temporary NST_MIN_DEFERRED)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 2>/dev/null | ms; }
for i in 1 2 3; do
  echo "6: $(NST_MIN_DEFERRED=6 one)  3: $(NST_MIN_DEFERRED=3 one)  12: $(NST_MIN_DEFERRED=12 one)  24: $(NST_MIN_DEFERRED=24 one)  1000: $(NST_MIN_DEFERRED=1000 one)"
done | tee gpurun_out/r06/c26_min_deferred.log
