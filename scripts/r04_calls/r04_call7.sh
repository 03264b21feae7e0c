#!/bin/bash
# Round 4, call 7: rocprofv3 kernel trace of the step as it stands (grouped weight gradients, conv2 forward on the 256 x 256 kernel)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
scripts/gpu_profile2.sh r04a_graph 8 > gpurun_out/r04a_graph_profile.log 2>&1; tail -3 gpurun_out/r04a_graph_profile.log
ls gpurun_out | grep r04a
