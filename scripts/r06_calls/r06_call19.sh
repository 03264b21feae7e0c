#!/bin/bash
# RECORD of a round-6 experiment: the NST_* switch(es) this script sets existed only in the working tree of that experiment
# (removed with it; the library now warns about them).  Kept for the log under profiles/r06_history/; it does not re-run.
echo "$0: record of a removed experiment (see the header); not runnable against this tree" >&2; exit 1
# call 19: the front end's LayerNorm + ReLU backward (576 000 rows) against the number of workgroups (NST_LN_BWD_CAP, temporary)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
for cap in 512 768 1024 1280 2048; do
  echo "cap $cap: $(NST_LN_BWD_CAP=$cap timeout 300 python scripts/conv_bench.py 2>/dev/null | grep -E 'conv2_ln_relu_bwd' | tr '\n' ' ')"
done | tee gpurun_out/r06/c19_ln_bwd_cap.log
