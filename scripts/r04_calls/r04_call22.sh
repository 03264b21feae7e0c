#!/bin/bash
# Round 4, call 22: which nst_comm_* call blocks the host while the producer stream is busy
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
NST_DIST_FORCE=1 timeout 300 python scripts/comm_probe.py 2>&1 | grep -v "^\[W\|amdgpu.ids" | tee $O/c22_comm_probe.log
