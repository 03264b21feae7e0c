#!/bin/bash
# Round 4, call 24: side effects of creating an RCCL communicator on unrelated kernels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
export NST_COMM_DEBUG=1
{
echo "== no torch process group"; timeout 200 python scripts/comm_side_effect.py 2>&1 | grep -v "^\[W\|amdgpu.ids"
echo "== torch process group (one forced rank) first"; timeout 200 python scripts/comm_side_effect.py --torch-pg 2>&1 | grep -v "^\[W\|amdgpu.ids"
echo "== torch process group + a second torch group, no native"; timeout 200 python scripts/comm_side_effect.py --torch-pg --second-torch-group --no-native 2>&1 | grep -v "^\[W\|amdgpu.ids"
echo "== no torch process group, RCCL of /opt/rocm"; NST_RCCL_PATH=/opt/rocm/lib/librccl.so.1 timeout 200 python scripts/comm_side_effect.py 2>&1 | grep -v "^\[W\|amdgpu.ids"
} | tee $O/c24_side_effect.log
