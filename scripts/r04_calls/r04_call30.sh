#!/bin/bash
# Round 4, call 30: conv2 weight gradient on the 256 x 256 tile core; the pruned library through the full GPU suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "conv2" > $O/c30_conv2_tests.log 2>&1
echo "conv2 tests rc=$? $(tail -n 1 $O/c30_conv2_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c30_conv2_tests.log | head
for v in 0 1; do
  NST_CONV2_WGRAD_G256=$v timeout 300 python scripts/conv_bench.py --iters 20 --out $O/c30_conv_bench_g$v.json 2>&1 | grep -i "conv2" | head -4
done
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do for v in 0 1; do
  echo "NST_CONV2_WGRAD_G256=$v  $(NST_CONV2_WGRAD_G256=$v step) ms/step"
done; done | tee $O/c30_ab_step.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -x > $O/c30_gpu_tests.log 2>&1
echo "gpu suite rc=$? $(grep -E 'passed|failed' $O/c30_gpu_tests.log | tail -n 1)"; grep -E "^FAILED|^ERROR|^E  " $O/c30_gpu_tests.log | head
