#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "xent" 2>&1 | tail -3
bash scripts/ab_env.sh 2 NST_XENT_VEC 0 1 -- --steps 20 --warmup 5
