#!/bin/bash
# Round 6, call 11: the feed-forward pair with its hidden dimension split over workgroups for the decoder's row count (+ row launch):
# kernel parity, model suites, step A/B and trace
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ffn.py -x -q -m gpu --tb=short 2>&1 | tail -15 | tee $O/c11_pytest_ffn.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py -x -q -m gpu --tb=short 2>&1 | tail -15 | tee $O/c11_pytest_model.log
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do echo "step  $(step) ms/step"; done | tee $O/c11_step.log
scripts/gpu_profile2.sh r06c11_graph 8 > $O/c11_profile.log 2>&1; tail -2 $O/c11_profile.log
