#!/bin/bash
# call 17: hidden-tile store split into an early read and a later store (feed-forward pair): tests, timings, cost model
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_ffn.py -m gpu -q -x --tb=short > gpurun_out/r06/c17_pytest_ffn.log 2>&1; tail -n 3 gpurun_out/r06/c17_pytest_ffn.log
timeout 300 python scripts/ffn_bench.py --rows 28800,9600 --out gpurun_out/r06/c17_ffn_bench.json > gpurun_out/r06/c17_ffn_bench.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06/c17_ffn_bench.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k,{a:round(b['fused_us'],1) for a,b in v.items()})
PY
timeout 600 python scripts/ffn_cost_model.py r06c17 > gpurun_out/r06/c17_cost_model.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06c17_ffn_cost_model.json'))
for k,v in d['cases'].items():
    for dd,x in v.items():
        print(k,dd,x['us_by_chunks'],x['fixed_us'],x['per_chunk_us'])
PY
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms_per_step', round(d['ms_per_step'],3))"; done
