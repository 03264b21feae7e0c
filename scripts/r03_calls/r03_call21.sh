#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export NST_GEMM_RING=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv1" 2>&1 | tail -3
timeout 600 python scripts/conv_bench.py 2>&1 | tail -12
