#!/bin/bash
# RECORD of a round-6 experiment: the NST_* switch(es) this script sets existed only in the working tree of that experiment
# (removed with it; the library now warns about them).  Kept for the log under profiles/r06_history/; it does not re-run.
echo "$0: record of a removed experiment (see the header); not runnable against this tree" >&2; exit 1
# call 22: the decoder stack's weight-gradient group launched behind the decoder (1: on the weight-gradient stream, 2: on the
# compute stream), the encoder's at the end (temporary NST_WGRAD_DEC_SIDE)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  echo "end $(one)   dec-side $(NST_WGRAD_DEC_SIDE=1 one)   dec-main $(NST_WGRAD_DEC_SIDE=2 one)"
done | tee gpurun_out/r06/c22_group_dec.log
