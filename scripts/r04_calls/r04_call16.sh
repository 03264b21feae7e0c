#!/bin/bash
# Round 4, call 16: sequence-mask / criterion-reduction kernels; whole GPU tier; step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1800 python -m pytest tests -m gpu -q --tb=short --durations=10 > $O/c16_gpu_tests.log 2>&1
echo "gpu tests rc=$? $(tail -n 1 $O/c16_gpu_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c16_gpu_tests.log | head -20
for r in 1 2 3; do
  ms=$(timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "default  $ms ms/step"
done | tee $O/c16_step.log
