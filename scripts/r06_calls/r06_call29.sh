#!/bin/bash
# call 29: attention forward skips query blocks past Tq and key blocks past Tk / above the diagonal: tests, timings, same-box A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "attention or attn" > gpurun_out/r06/c29_pytest_attn.log 2>&1; tail -n 3 gpurun_out/r06/c29_pytest_attn.log
echo "prev: $(NST_LIBRARY=$PWD/neurst_amd/lib/libneurst_hip_prev.so python scripts/attn_bench.py 2>/dev/null | tail -1)"
echo "new:  $(python scripts/attn_bench.py 2>/dev/null | tail -1)"
AB_NO_FFN=1 bash scripts/ab_libs.sh c29 3
