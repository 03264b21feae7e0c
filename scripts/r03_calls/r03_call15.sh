#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "attention" > $O/r03_attn_tests_call15.log 2>&1
echo "attention tests rc=$? $(tail -n 1 $O/r03_attn_tests_call15.log)"; grep -E "^FAILED|^ERROR" $O/r03_attn_tests_call15.log | head -20; grep -E "^E  " $O/r03_attn_tests_call15.log | head -10
bash scripts/ab_env.sh 3 NST_ATTN_FUSED_BWD 0 1 -- --steps 20 --warmup 5 2>&1 | tee $O/r03_ab_attn_fused_bwd.log
