#!/bin/bash
# One GPU session: parity tests, kernel micro-benchmarks, train-step bench.   usage: scripts/gpu_round.sh <tag> [bench groups] [pytest -k expr]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
TAG=${1:-r}
GROUPS_=${2:-attn}
KEXPR=${3:-}
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short -k "$KEXPR" > gpurun_out/${TAG}_tests.log 2>&1
else
  timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --tb=short > gpurun_out/${TAG}_tests.log 2>&1
fi
echo "gpu tests rc=$? $(tail -n 1 gpurun_out/${TAG}_tests.log)"
grep -E "^FAILED|^ERROR" gpurun_out/${TAG}_tests.log | head -20
if [ "$GROUPS_" != "none" ]; then
  timeout 600 python scripts/kernel_bench.py --only ${GROUPS_} > gpurun_out/${TAG}_kb.log 2>&1; grep -E "us " gpurun_out/${TAG}_kb.log | head -80
fi
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -n 1 gpurun_out/${TAG}_bench.json | cut -c1-420; tail -n 3 gpurun_out/${TAG}_bench.err | cut -c1-300
