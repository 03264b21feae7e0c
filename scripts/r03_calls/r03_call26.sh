#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
bash scripts/ab_env.sh 2 NST_CONV1_BWD_V2 0 1 -- --steps 20 --warmup 5
