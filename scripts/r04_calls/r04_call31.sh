#!/bin/bash
# Round 4, call 31: conv1 forward with two pixels per wavefront pass (parity, stand-alone, in-step A/B)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "conv1" > $O/c31_conv1_tests.log 2>&1
echo "conv1 tests rc=$? $(tail -n 1 $O/c31_conv1_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c31_conv1_tests.log | head
for v in 0 1; do
  NST_CONV1_FWD_PAIR=$v timeout 300 python scripts/conv_bench.py --iters 20 --out $O/c31_conv_bench_p$v.json 2>&1 | grep -i "conv1" | head -4
done
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do for v in 0 1; do
  echo "NST_CONV1_FWD_PAIR=$v  $(NST_CONV1_FWD_PAIR=$v step) ms/step"
done; done | tee $O/c31_ab_step.log
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -x > $O/c31_model_tests.log 2>&1
echo "model tests rc=$? $(tail -n 1 $O/c31_model_tests.log)"
