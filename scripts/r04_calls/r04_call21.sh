#!/bin/bash
# Round 4, call 21: why the forced exchange through nst_comm_* costs 9 ms per step in graph mode (stream priority, hardware queues,
# eager mode), and a kernel trace of it
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
export NST_DIST_FORCE=1
{
echo "torch.distributed                         $(step) ms/step"
echo "native, high-priority stream              $(NST_DIST_NATIVE=1 step) ms/step"
echo "native, default priority                  $(NST_DIST_NATIVE=1 NST_COMM_STREAM_PRIORITY=0 step) ms/step"
echo "native, default priority, 8 hw queues     $(NST_DIST_NATIVE=1 NST_COMM_STREAM_PRIORITY=0 GPU_MAX_HW_QUEUES=8 step) ms/step"
echo "native, high priority, 8 hw queues        $(NST_DIST_NATIVE=1 GPU_MAX_HW_QUEUES=8 step) ms/step"
echo "torch.distributed, eager                  $(step --eager) ms/step"
echo "native, eager                             $(NST_DIST_NATIVE=1 step --eager) ms/step"
} | tee $O/c21_native.log
cd /tmp && export TMPDIR=/tmp
NST_DIST_NATIVE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/c21_trace -o native -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --roofline-steps 0 --steps 4 --warmup 3 > $O/c21_trace.log 2>&1
ls -la $O/c21_trace | head
