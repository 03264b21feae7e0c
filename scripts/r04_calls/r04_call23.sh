#!/bin/bash
# Round 4, call 23: what slows the step down with the library's communicator (diagnostic switches)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
export NST_DIST_FORCE=1
{
echo "torch.distributed                                   $(step) ms/step"
echo "native                                              $(NST_DIST_NATIVE=1 step) ms/step"
echo "communicator created, torch path used               $(NST_DIST_NATIVE=1 NST_DIST_NATIVE_IDLE=1 step) ms/step"
echo "native, ncclAllReduce skipped (events + fence only) $(NST_DIST_NATIVE=1 NST_COMM_SKIP_NCCL=1 step) ms/step"
echo "native, 16-bit wire                                 $(NST_DIST_NATIVE=1 step --wire bf16) ms/step"
echo "torch, 16-bit wire                                  $(step --wire bf16) ms/step"
} | tee $O/c23_native.log
rocm-smi --showclocks 2>/dev/null | head -20
