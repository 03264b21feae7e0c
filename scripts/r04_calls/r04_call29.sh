#!/bin/bash
# Round 4, call 29: full GPU suite and the bench variants with the new process defaults (one hardware queue per priority class)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q --tb=short -x > $O/c29_gpu_tests.log 2>&1
echo "gpu suite rc=$? $(grep -E 'passed|failed' $O/c29_gpu_tests.log | tail -n 1)"; grep -E "^FAILED|^ERROR|^E  " $O/c29_gpu_tests.log | head
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
{
for r in 1 2; do
echo "one GPU                              $(step)"
echo "one GPU, eager                       $(step --eager)"
echo "forced exchange, torch.distributed   $(NST_DIST_FORCE=1 step)"
echo "forced exchange, nst_comm_*          $(NST_DIST_FORCE=1 NST_DIST_NATIVE=1 step)"
echo "forced exchange, nst_comm_*, bf16 wire $(NST_DIST_FORCE=1 NST_DIST_NATIVE=1 step --wire bf16)"
done
} | tee $O/c29_bench.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/c29_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/c29_smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/c29_bench_default.json 2> $O/c29_bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/c29_bench_default.json
