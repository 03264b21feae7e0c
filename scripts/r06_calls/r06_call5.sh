#!/bin/bash
# Round 6, call 5: whole-row products, 48 rows x 4 stages for the decoder row counts: (rows per workgroup, stages) sweep, graph-replayed timings;
# parity of the new entries; the vectorised LayerNorm finalize; step A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_rowgemm.py -q -m gpu --tb=short 2>&1 | tail -5 | tee $O/c5_pytest_rowgemm.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -k "layernorm" 2>&1 | tail -5 | tee $O/c5_pytest_ln.log
for cfg in "" "32,2" "64,3" "48,4"; do
  NST_ROWGEMM_CFG=$cfg timeout 300 python scripts/rowgemm_bench.py r06_c5_${cfg/,/_} 2>/dev/null | tail -45 > $O/c5_bench_${cfg/,/_}.json
done
python - <<'PY'
import json,glob
r={}
for f in sorted(glob.glob("gpurun_out/r06/c5_bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, e); continue
    r[d.get("NST_ROWGEMM_CFG") or "default"]=d
keys=[k for k in next(iter(r.values())) if k.endswith("_us")]
print("%-36s"%"case"+"".join("%10s"%c for c in r))
for k in keys: print("%-36s"%k+"".join("%10.2f"%r[c].get(k,float('nan')) for c in r))
PY
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do for v in 1 0; do
  echo "NST_ROW_FUSION=$v  $(NST_ROW_FUSION=$v step) ms/step"
done; done | tee $O/c5_ab_rows.log
scripts/gpu_profile2.sh r06c5_graph 8 > $O/c5_profile.log 2>&1; tail -2 $O/c5_profile.log
