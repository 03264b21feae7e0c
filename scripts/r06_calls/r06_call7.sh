#!/bin/bash
# Round 6, call 7: the feed-forward pair with the wrapper's row stages (nst_ffn_add_layernorm_fwd / nst_ffn_layernorm_bwd): kernel
# parity, the model / graph suites on the fused path, step A/B, kernel trace
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ffn.py tests/test_gpu_rowgemm.py -q -m gpu --tb=short 2>&1 | tail -15 | tee $O/c7_pytest_ffn_rows.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py -x -q -m gpu --tb=short 2>&1 | tail -15 | tee $O/c7_pytest_model.log
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do for v in 1 0; do
  echo "NST_ROW_FUSION=$v  $(NST_ROW_FUSION=$v step) ms/step"
done; done | tee $O/c7_ab_rows.log
scripts/gpu_profile2.sh r06c7_graph 8 > $O/c7_profile.log 2>&1; tail -2 $O/c7_profile.log
