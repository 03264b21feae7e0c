#!/bin/bash
# Round 5, call 27: capture with the collector held off; the order that crashed (model tests, then graph tests) in one interpreter
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -X faulthandler -m pytest tests/test_gpu_graph.py -q -m gpu --tb=short -x > $O/c27_graph.log 2>&1; echo "graph rc=$?"; grep -v "amdgpu.ids" $O/c27_graph.log | tail -3 | cut -c1-300
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py -q -m gpu --tb=short -x > $O/c27_both.log 2>&1; echo "both rc=$?"; grep -v "amdgpu.ids" $O/c27_both.log | tail -3 | cut -c1-300
