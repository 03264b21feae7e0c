#!/bin/bash
# Round 6, call 15: compile-time epilogues for the fp32 products (BASELINE config #2): GEMM parity in fp32, the fp32 model cases, fp32 step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -k "gemm" 2>&1 | tail -5 | tee $O/c15_pytest_gemm.log
timeout 1500 python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -k "float32" 2>&1 | tail -5 | tee $O/c15_pytest_model_fp32.log
for r in 1 2; do timeout 400 python bench.py --dtype fp32 --steps 6 --warmup 2 --no-cpu-baseline --roofline-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("fp32", d["ms_per_step"])'; done | tee $O/c15_fp32_step.log
