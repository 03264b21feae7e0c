#!/bin/bash
# Round 5, call 23: front dense split-K reduce in the same weight-gradient graph as its GEMM; cuts per captured step (min_deferred)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_graph.py -q -m gpu --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -2
for rep in 1 2 3; do
for md in 6 12 1000; do
timeout 300 python scripts/r05_experiments/bench_min_deferred.py $md --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('min_deferred $md', round(d['ms_per_step'],3))"
done; done
scripts/gpu_profile2.sh r05c23 8 > $O/c23_profile.log 2>&1
tail -1 gpurun_out/r05c23_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05c23_timeline_trace.csv')))
e=lambda r: float(r['start_us'])+float(r['dur_us'])
print("span", round(max(e(r) for r in rows),1), "main end", round(max(e(r) for r in rows if r['stream']=='2'),1), "side end", round(max(e(r) for r in rows if r['stream']=='1'),1))
PY
