#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_multi.py -m gpu -q --tb=short -rs > $O/c9_graph_tests.log 2>&1
echo "graph/multi tests rc=$? $(tail -n 1 $O/c9_graph_tests.log)"; grep -E "^FAILED|^ERROR|^E  |SKIP" $O/c9_graph_tests.log | head -20
