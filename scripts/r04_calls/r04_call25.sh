#!/bin/bash
# Round 4, call 25: queues / threads of the process with and without the library's communicator
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>$O/c25_err.log | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; grep "\[diag\]" $O/c25_err.log | cut -c1-400; }
export NST_DIST_FORCE=1 NST_BENCH_DIAG=1
nproc
{
echo "torch.distributed                      $(step)"
echo "communicator idle                      $(NST_DIST_NATIVE=1 NST_DIST_NATIVE_IDLE=1 step)"
echo "communicator created and destroyed     $(NST_DIST_NATIVE=1 NST_DIST_NATIVE_IDLE=2 step)"
echo "communicator idle, RCCL of /opt/rocm   $(NST_DIST_NATIVE=1 NST_DIST_NATIVE_IDLE=1 NST_RCCL_PATH=/opt/rocm/lib/librccl.so.1 step)"
echo "no torch group, no exchange            $(NST_DIST_FORCE=0 step)"
} | tee $O/c25_diag.log
