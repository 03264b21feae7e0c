#!/bin/bash
# Round 3, call 10: whole GPU tier with the eight-wave feed-forward kernels and the LayerNorm finalize, then the step A/B.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short > $O/r03_gpu_tests_call10.log 2>&1
echo "gpu tests rc=$? $(tail -n 1 $O/r03_gpu_tests_call10.log)"; grep -E "^FAILED|^ERROR" $O/r03_gpu_tests_call10.log | head -20
for r in 1 2 3; do for v in 0 1; do
  ms=$(NST_FFN_V2=$v NST_FFN_V2_BWD=$v python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "NST_FFN_V2=$v  $ms ms/step"
done; done | tee $O/r03_ab_ffn_v2.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_call10.json 2> $O/r03_bench_call10.err; tail -n 1 $O/r03_bench_call10.json | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print(d["ms_per_step"], d["host_issue_ms_per_step"])
for n, f in d["roofline_families"].items(): print("  ", n, round(f["ms_per_step"], 3), round(f.get("frac", 0), 4))
'
