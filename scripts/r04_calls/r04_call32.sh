#!/bin/bash
# Round 4, call 32: conv1 forward pair kernel with the next row prefetched; two rows in flight for the 576 000-row LayerNorm backward
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "conv1 or layernorm or test_ln" > $O/c32_tests.log 2>&1
echo "conv1 / LayerNorm tests rc=$? $(tail -n 1 $O/c32_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c32_tests.log | head
for v in 1 2; do
  echo "NST_LN_BWD_BIG_U=$v"; NST_LN_BWD_BIG_U=$v timeout 300 python scripts/conv_bench.py --iters 20 --out $O/c32_conv_bench_u$v.json 2>&1 | grep -i "conv1_fwd\|ln_relu" | head -4
done
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do
  echo "old conv1 fwd, big LN U=1     $(NST_CONV1_FWD_PAIR=0 NST_LN_BWD_BIG_U=1 step) ms/step"
  echo "pair conv1 fwd, big LN U=1    $(NST_LN_BWD_BIG_U=1 step) ms/step"
  echo "pair conv1 fwd, big LN U=2    $(step) ms/step"
done | tee $O/c32_ab_step.log
