#!/bin/bash
# Round 4, call 17: the cross-attention k|v and front dense weight gradients as small side-stream groups (A/B), graph tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_multi.py -m gpu -q --tb=short > $O/c17_graph_tests.log 2>&1
echo "graph/multi tests rc=$? $(tail -n 1 $O/c17_graph_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c17_graph_tests.log | head
for r in 1 2 3; do for v in 0 1; do
  ms=$(NST_WGRAD_SIDE_GROUP=$v timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "NST_WGRAD_SIDE_GROUP=$v  $ms ms/step"
done; done | tee $O/c17_ab_step.log
