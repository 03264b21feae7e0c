#!/bin/bash
# Round 4, call 27: the step / weight-gradient / exchange streams in three priority classes: A/B, robustness against the number of
# streams created before, full GPU suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
{
for r in 1 2; do
echo "one GPU: step high, wgrad low (default)   $(step)"
echo "one GPU: step high, wgrad default class   $(NST_WGRAD_PRIORITY=0 step)"
echo "one GPU: caller's stream, wgrad low       $(NST_STEP_PRIORITY=0 step)"
echo "one GPU: caller's stream, wgrad default   $(NST_STEP_PRIORITY=0 NST_WGRAD_PRIORITY=0 step)"
done
echo "one GPU, eager: default                   $(step --eager)"
echo "one GPU, eager: old streams               $(NST_STEP_PRIORITY=0 NST_WGRAD_PRIORITY=0 step --eager)"
export NST_DIST_FORCE=1
echo "forced exchange, torch.distributed        $(step)"
for k in 0 1 2 3 4 5; do echo "forced exchange, native, $k padding streams  $(NST_DIST_NATIVE=1 NST_DIST_PAD_STREAMS=$k step)"; done
for k in 0 3; do echo "forced exchange, native, $k pads, OLD streams $(NST_STEP_PRIORITY=0 NST_WGRAD_PRIORITY=0 NST_DIST_NATIVE=1 NST_DIST_PAD_STREAMS=$k step)"; done
} | tee $O/c27_streams.log
unset NST_DIST_FORCE
timeout 2400 python -m pytest tests -m gpu -q --tb=short -x > $O/c27_gpu_tests.log 2>&1
echo "gpu suite rc=$? $(grep -E 'passed|failed' $O/c27_gpu_tests.log | tail -n 1)"; grep -E "^FAILED|^ERROR|^E  " $O/c27_gpu_tests.log | head
