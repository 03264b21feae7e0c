#!/bin/bash
# Model-level parity + smoke + first bench + rocprof kernel stats.  Logs land in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short > gpurun_out/model_tests.log 2>&1
echo "[model tests] rc=$? : $(tail -n 1 gpurun_out/model_tests.log)"
grep -E "^FAILED|^ERROR" gpurun_out/model_tests.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "[smoke] rc=$? : $(tail -n 1 gpurun_out/smoke.log)"
timeout 900 python bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-2} ${BENCH_ARGS} > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "[bench] rc=$? : $(tail -c 1500 gpurun_out/bench.log)"
tail -n 5 gpurun_out/bench.err
if [ -n "$PROFILE" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof.log 2>&1
  echo "[rocprof] rc=$?"
  cd $R
  find gpurun_out/prof -name "*kernel_stats*" | head -3
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -n 25 "$f"
  # keep the merge small: drop the per-dispatch trace if it is huge
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
