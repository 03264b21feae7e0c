#!/bin/bash
# Round 5, call 5: LayerNorm + ReLU backward with the recomputed gate (front end): parity and step time
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "layernorm or frontend or forward_backward" --tb=short 2>&1 | grep -v "amdgpu.ids" > $O/c5_pytest.log
tail -8 $O/c5_pytest.log
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do echo "regate  $(step) ms/step"; done | tee $O/c5_step.log
