#!/bin/bash
# Round 4, call 3: grouped weight gradients (nst_gemm_wgrad_group): parity, stand-alone timings + ablations, model parity, step A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "gemm" > $O/c3_gemm_tests.log 2>&1
echo "gemm tests rc=$? $(tail -n 1 $O/c3_gemm_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c3_gemm_tests.log | head
for m in 0 11 12 14; do
  NST_GEMM256=$m timeout 300 python scripts/wgrad_group_bench.py 2>$O/c3_group_bench_$m.err | tail -n 1 | tee $O/c3_group_bench_$m.json | cut -c1-1500
done
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -x > $O/c3_model_tests.log 2>&1
echo "model tests rc=$? $(tail -n 1 $O/c3_model_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c3_model_tests.log | head
for r in 1 2; do
  for cfg in "NST_WGRAD_GROUP=0" "NST_WGRAD_GROUP_AT=end" "NST_WGRAD_GROUP_AT=stack" "NST_WGRAD_GROUP_AT=decoder"; do
    ms=$(env $cfg timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>$O/c3_bench_err.log | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
    echo "$cfg  $ms ms/step"
  done
done | tee $O/c3_ab_step.log
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/c3_bench.json 2> $O/c3_bench.err; tail -n 1 $O/c3_bench.json | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print(d["ms_per_step"], d.get("host_issue_ms_per_step"))
for n, f in d["roofline_families"].items(): print("  ", n, round(f["ms_per_step"], 3), round(f.get("frac", 0), 4), f["launches_per_step"])
'
