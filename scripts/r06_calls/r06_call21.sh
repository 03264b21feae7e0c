#!/bin/bash
# call 21: lean DMA issue + running chunk indices in the feed-forward loop: tests, same-box A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ffn.py -m gpu -q -x --tb=short > gpurun_out/r06/c21_pytest_ffn.log 2>&1; tail -n 2 gpurun_out/r06/c21_pytest_ffn.log
bash scripts/ab_libs.sh c21 3
