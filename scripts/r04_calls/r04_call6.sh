#!/bin/bash
# Round 4, call 6: conv2 forward on the phase-staggered 256 x 256 kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "conv2" > $O/c6_conv_tests.log 2>&1
echo "conv tests rc=$? $(tail -n 1 $O/c6_conv_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c6_conv_tests.log | head
for v in 1 0 1 0; do
  echo "NST_CONV2_G256=$v"; NST_CONV2_G256=$v timeout 300 python scripts/conv_bench.py --iters 10 2>&1 | grep conv2_
done | tee $O/c6_conv_bench.log
for r in 1 2; do for v in 0 1; do
  ms=$(NST_CONV2_G256=$v timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "NST_CONV2_G256=$v  $ms ms/step"
done; done | tee $O/c6_ab_step.log
