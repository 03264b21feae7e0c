#!/bin/bash
# Round 4, call 34: conv1 forward pair kernel at four waves per SIMD (128 registers, no prefetch)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "conv1" > $O/c34_conv1_tests.log 2>&1
echo "conv1 tests rc=$? $(tail -n 1 $O/c34_conv1_tests.log)"
timeout 300 python scripts/conv_bench.py --iters 20 --out $O/c34_conv_bench.json 2>&1 | grep -i "conv1\|ln_relu" | head -4
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do echo "step $(step) ms"; done
