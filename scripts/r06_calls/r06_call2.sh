#!/bin/bash
# Round 6, call 2: whole-row products inside the model: full kernel parity of the new entries, model / graph / ffn suites, step time
# A/B (fused rows on / off), kernel trace of the graph-replayed step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_rowgemm.py -q -m gpu --tb=short 2>&1 | tail -15 | tee $O/c2_pytest_rowgemm.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py -x -q -m gpu --tb=short 2>&1 | tail -15 | tee $O/c2_pytest_model.log
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do for v in 1 0; do
  echo "NST_ROW_FUSION=$v  $(NST_ROW_FUSION=$v step) ms/step"
done; done | tee $O/c2_ab_rows.log
scripts/gpu_profile2.sh r06c2_graph 8 > $O/c2_profile.log 2>&1; tail -2 $O/c2_profile.log
