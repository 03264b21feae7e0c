#!/bin/bash
# Round 6, call 1: the whole-row products (ABI 10): kernel parity against the float64 contract and the unfused pairs, timings
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_rowgemm.py -x -q -m gpu --tb=short 2>&1 | tail -25 | tee $O/c1_pytest_rowgemm.log
timeout 300 python scripts/rowgemm_bench.py r06_c1 2>&1 | tail -40 | tee $O/c1_rowgemm_bench.log
