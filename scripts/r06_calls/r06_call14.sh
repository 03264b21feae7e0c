#!/bin/bash
# Round 6, call 14: conv2 forward (and data gradient) with the taps of a channel slice on consecutive K steps (L2 re-use of the shifted dy windows):
# parity, stand-alone timing, HBM counters; the improved slab-row kernels in the step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -k "conv2" 2>&1 | tail -5 | tee $O/c14_pytest_conv2.log
timeout 300 python scripts/conv_bench.py --out $O/c14_conv_bench.json 2>&1 | tail -12
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  scripts/pmc_kernel.sh $O/c14_pmc_conv2.json conv2_ scripts/conv_bench.py --iters 3 > $O/c14_pmc_conv2.log 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06/c14_pmc_conv2.json"))
for k,v in d.get("kernels",{}).items():
    print(k[:60], {c:(sum(x)/len(x)) for c,x in v.items() if c in ("FETCH_SIZE","WRITE_SIZE","SQ_VALU_MFMA_BUSY_CYCLES","SQ_BUSY_CYCLES")})
PY
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 "$@" 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2; do echo "step  $(step) ms/step"; done | tee $O/c14_step.log
