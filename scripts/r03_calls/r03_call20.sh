#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export NST_GEMM_RING=0
bash scripts/ab_env.sh 2 NST_CONV2_WGRAD_AT side after_dgrad last -- --steps 20 --warmup 5
