#!/bin/bash
# Round 4, call 2: fewer, longer units on the 256 x 256 weight-gradient kernel (CU time instead of stand-alone latency)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
for u in 64 32 96; do
  ms=$(NST_WGRAD256_UNITS=$u timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "NST_WGRAD256_UNITS=$u  $ms ms/step"
done | tee $O/c2_ab_step.log
ms=$(NST_GEMM256=0 timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
echo "NST_GEMM256=0  $ms ms/step" | tee -a $O/c2_ab_step.log
