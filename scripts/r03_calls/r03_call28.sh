#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for m in 0 2 3 0 2 3; do echo -n "ISSUE_LATE=$m "; NST_CONV_ISSUE_LATE=$m timeout 300 python scripts/conv_bench.py 2>/dev/null | grep -E "conv2_fwd|conv2_dgrad" | sed 's/, .tflops.*//' | tr '\n' ' '; echo; done
