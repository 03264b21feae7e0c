#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "conv2" > $O/r03_conv_tests_call14.log 2>&1
echo "conv tests rc=$? $(tail -n 1 $O/r03_conv_tests_call14.log)"; grep -E "^FAILED|^ERROR|^E  " $O/r03_conv_tests_call14.log | head
for r in 1 2 3; do for v in 0 1; do
  echo -n "NST_CONV_ISSUE_LATE=$v: "; NST_CONV_ISSUE_LATE=$v timeout 300 python scripts/conv_bench.py --iters 20 2>/dev/null | tail -1 | cut -c1-400
done; done | tee $O/r03_conv_issue_late.log
bash scripts/ab_env.sh 3 NST_CONV_ISSUE_LATE 0 1 -- --steps 20 --warmup 5 2>&1 | tee -a $O/r03_conv_issue_late.log
