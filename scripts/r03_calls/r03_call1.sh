#!/bin/bash
# Round 3, first GPU call:  gpurun --timeout 1500 -- 'bash scripts/r03_call1.sh'
#  1. the whole -m gpu tier (incl. the round-3 parity cases on the timed path), 2. smoke, 3. the bench line (graph replay,
#  the default) and the eager one, 4. the two-rank rehearsal started by bench.py itself (host-staged gloo, shared device).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x --durations=15 > $O/r03_gpu_tests_call1.log 2>&1
echo "gpu tests rc=$? $(tail -n 1 $O/r03_gpu_tests_call1.log)"; grep -E "^FAILED|^ERROR" $O/r03_gpu_tests_call1.log | head -20
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_ffn.py -m gpu -q --tb=short -k "noise or benchmark or bench_" > $O/r03_new_tests_call1.log 2>&1
echo "new tests rc=$? $(tail -n 1 $O/r03_new_tests_call1.log)"; grep -E "^FAILED|^ERROR" $O/r03_new_tests_call1.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke.log 2>&1; echo "smoke rc=$? $(tail -n 1 $O/r03_smoke.log)"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r03_bench_graph_call1.json 2> $O/r03_bench_graph_call1.err; echo "bench(graph) rc=$?"; tail -n 1 $O/r03_bench_graph_call1.json | cut -c1-420
timeout 600 python bench.py --steps 20 --warmup 5 --eager --no-cpu-baseline > $O/r03_bench_eager_call1.json 2> $O/r03_bench_eager_call1.err; echo "bench(eager) rc=$?"; tail -n 1 $O/r03_bench_eager_call1.json | cut -c1-420
NST_DIST_BACKEND=gloo NST_BENCH_HANG_DUMP_S=240 timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --roofline-steps 1 > $O/r03_two_rank_rehearsal.log 2>&1
echo "two-rank rehearsal rc=$?"; grep -E '^\{' $O/r03_two_rank_rehearsal.log | cut -c1-500
NST_DIST_FORCE=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --wire bf16 > $O/r03_bench_forced_bf16wire.json 2> $O/r03_bench_forced_bf16wire.err; echo "forced exchange bf16 wire rc=$?"; tail -n 1 $O/r03_bench_forced_bf16wire.json | cut -c1-300
