#!/bin/bash
# Round 4, call 26: is it WHEN the communicator is created (stream -> hardware queue assignment of the step's streams)?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
export NST_DIST_FORCE=1 NST_DIST_NATIVE=1
{
echo "native, communicator created with the reducer   $(step)"
echo "native, communicator created in init_distributed $(NST_DIST_NATIVE_EARLY=1 step)"
for k in 1 2 3 4 5; do echo "native, late, $k padding streams               $(NST_DIST_PAD_STREAMS=$k step)"; done
echo "native, early, 16-bit wire                       $(NST_DIST_NATIVE_EARLY=1 step --wire bf16)"
} | tee $O/c26_order.log
