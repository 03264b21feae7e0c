#!/bin/bash
# Round 4, call 28: fewer hardware queues per priority class (GPU_MAX_HW_QUEUES) against the 22-31 ms steps
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
{
for q in 1 2; do
echo "hwq=$q one GPU, three classes                 $(GPU_MAX_HW_QUEUES=$q step)"
echo "hwq=$q one GPU, old streams                   $(GPU_MAX_HW_QUEUES=$q NST_STEP_PRIORITY=0 NST_WGRAD_PRIORITY=0 step)"
for k in 0 2 3; do echo "hwq=$q forced native, $k pads, three classes   $(GPU_MAX_HW_QUEUES=$q NST_DIST_FORCE=1 NST_DIST_NATIVE=1 NST_DIST_PAD_STREAMS=$k step)"; done
echo "hwq=$q forced torch, three classes            $(GPU_MAX_HW_QUEUES=$q NST_DIST_FORCE=1 step)"
done
echo "hwq=1 forced native, 3 pads, eager, 3 classes $(GPU_MAX_HW_QUEUES=1 NST_DIST_FORCE=1 NST_DIST_NATIVE=1 NST_DIST_PAD_STREAMS=3 step --eager)"
echo "default hwq, forced native, 3 pads, eager, single stream (no weight-gradient stream) $(NST_WGRAD_STREAM=0 NST_DIST_FORCE=1 NST_DIST_NATIVE=1 NST_DIST_PAD_STREAMS=3 step --eager)"
} | tee $O/c28_hwq.log
dmesg 2>/dev/null | tail -5
