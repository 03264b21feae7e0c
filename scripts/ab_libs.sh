#!/bin/bash
# Same-box A/B of two builds of the library (boxes of the pool differ by +-2 % under load, more than most kernel changes):
# alternates bench.py between neurst_amd/lib/libneurst_hip_prev.so (A) and the product library (B), then the feed-forward cost
# model for both.   bash scripts/ab_libs.sh <tag> [rounds]
cd "${GRAFT_REPO_ROOT:-.}"
T=${1:-ab}; R=${2:-3}
mkdir -p gpurun_out/r06
PREV=$PWD/neurst_amd/lib/libneurst_hip_prev.so
one() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for i in $(seq $R); do
  echo "A(prev) $(NST_LIBRARY=$PREV one)   B(new) $(one)"
done | tee gpurun_out/r06/${T}_ab_bench.log
if [ -z "$AB_NO_FFN" ]; then
NST_LIBRARY=$PREV timeout 600 python scripts/ffn_cost_model.py ${T}_prev > /dev/null 2>&1
timeout 600 python scripts/ffn_cost_model.py ${T}_new > /dev/null 2>&1
python - <<PY
import json
for w in ("prev","new"):
    d=json.load(open("gpurun_out/${T}_%s_ffn_cost_model.json" % w))
    for k,v in d['cases'].items():
        for dd,x in v.items():
            print(w,k,dd,x['us_by_chunks'],x['fixed_us'],x['per_chunk_us'])
PY
fi
