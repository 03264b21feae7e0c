#!/bin/bash
# Round 5, call 1: fp32 residual stream (nst_add_layernorm_fwd / nst_layernorm_bwd_mixed): kernel parity, model parity against the
# oracle fixtures, and what the stream costs in the step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm" 2>&1 | tail -6 | tee $O/c1_pytest_ln.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "oracle_fixture or forward_backward" 2>&1 | tail -8 | tee $O/c1_pytest_model.log
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do for v in 0 1; do
  echo "NST_STREAM32=$v  $(NST_STREAM32=$v step) ms/step"
done; done | tee $O/c1_ab_stream32.log
