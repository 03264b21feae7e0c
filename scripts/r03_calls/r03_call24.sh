#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
bash scripts/ab_env.sh 2 NST_FFN_MIN_ROWS 16384 8192 -- --steps 20 --warmup 5
NST_FFN_MIN_ROWS=8192 bash scripts/ab_env.sh 1 NST_FFN_NW 2 4 -- --steps 20 --warmup 5
