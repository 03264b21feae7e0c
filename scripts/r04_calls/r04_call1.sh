#!/bin/bash
# Round 4, call 1: the phase-staggered 256 x 256 weight-gradient kernel (nst_gemm256.h): parity, stand-alone timings against
# the 128 x 128 stream kernel, timing ablations, then the step A/B.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "gemm" > $O/c1_gemm_tests.log 2>&1
echo "gemm tests rc=$? $(tail -n 1 $O/c1_gemm_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c1_gemm_tests.log | head
timeout 600 python scripts/gemm256_bench.py > $O/c1_g256_new.json 2> $O/c1_g256_new.err; echo "g256 bench rc=$?"; tail -c 3000 $O/c1_g256_new.json
NST_GEMM256=0 timeout 600 python scripts/gemm256_bench.py > $O/c1_g256_old.json 2> $O/c1_g256_old.err; echo "old bench rc=$?"; tail -c 2500 $O/c1_g256_old.json
for m in 11 12 14; do
  NST_GEMM256=$m G256_ONLY=enc.ffn1 timeout 300 python scripts/gemm256_bench.py 2>/dev/null | tail -n 1 | python -c '
import sys, json
d = json.loads(sys.stdin.read()); print("mode", d["NST_GEMM256"], d["us"])'
done | tee $O/c1_g256_ablation.log
for r in 1 2; do for v in 0 1; do
  ms=$(NST_GEMM256=$v timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>$O/c1_bench_err_$v.log | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "NST_GEMM256=$v  $ms ms/step"
done; done | tee $O/c1_ab_step.log
for u in 128 192; do
  ms=$(NST_WGRAD256_UNITS=$u timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "NST_WGRAD256_UNITS=$u  $ms ms/step"
done | tee -a $O/c1_ab_step.log
