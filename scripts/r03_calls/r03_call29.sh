#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv2" 2>&1 | tail -3
for m in 0 1 0 1; do echo -n "PATCH256=$m "; NST_CONV2_PATCH256=$m timeout 300 python scripts/conv_bench.py 2>/dev/null | grep -E "conv2_fwd" | sed 's/, .tflops.*//'; done
