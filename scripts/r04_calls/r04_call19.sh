#!/bin/bash
# Round 4, call 19: native RCCL entry points (one-rank communicator), loss scale + clipping on the device, forced-exchange bench
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_graph.py -m gpu -q --tb=short > $O/c19_tests.log 2>&1
echo "multi/graph tests rc=$? $(tail -n 1 $O/c19_tests.log)"; grep -E "^FAILED|^ERROR|^E  |SKIPPED" $O/c19_tests.log | head -20
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>$O/c19_bench_err.log | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do
  echo "one GPU, no exchange:            $(step) ms/step"
  echo "forced exchange, torch.distributed: $(NST_DIST_FORCE=1 step) ms/step"
  echo "forced exchange, nst_comm_*:        $(NST_DIST_FORCE=1 NST_DIST_NATIVE=1 step) ms/step"
done | tee $O/c19_ab_native.log
tail -5 $O/c19_bench_err.log
