#!/bin/bash
# Round 5, call 26: the crash of call 25's second pytest command, with the full log
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -X faulthandler -m pytest tests/test_gpu_graph.py -q -m gpu --tb=short -x > $O/c26_graph.log 2>&1; echo "graph rc=$?"; grep -v "amdgpu.ids" $O/c26_graph.log | tail -5 | cut -c1-300
timeout 900 python -X faulthandler -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -x > $O/c26_model.log 2>&1; echo "model rc=$?"; grep -v "amdgpu.ids" $O/c26_model.log | tail -5 | cut -c1-300
timeout 900 python -X faulthandler -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py -q -m gpu --tb=short -x > $O/c26_both.log 2>&1; echo "both rc=$?"; grep -v "amdgpu.ids" $O/c26_both.log | tail -5 | cut -c1-300
