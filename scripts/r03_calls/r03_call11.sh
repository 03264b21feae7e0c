#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "gemm" > $O/r03_gemm_tests_call11.log 2>&1
echo "gemm tests rc=$? $(tail -n 1 $O/r03_gemm_tests_call11.log)"; grep -E "^FAILED|^ERROR|^E  " $O/r03_gemm_tests_call11.log | head
for v in 1 0 1 0; do
  echo "NST_GEMM_SPLIT_ISSUE=$v"; NST_GEMM_SPLIT_ISSUE=$v timeout 300 python scripts/gemm_iso.py 2>/dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("   " + "  ".join("%s=%.1f" % (k.split()[0], v) for k, v in d["us"].items()))'
done
for r in 1 2 3; do for v in 0 1; do
  ms=$(NST_GEMM_SPLIT_ISSUE=$v python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "NST_GEMM_SPLIT_ISSUE=$v  $ms ms/step"
done; done | tee $O/r03_ab_split_issue.log
