#!/bin/bash
# Round 4, call 39: PMC counters of the conv1 kernels (what the 2.5-2.9 TB/s are bound by)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS;FETCH_SIZE;WRITE_SIZE" \
  scripts/pmc_kernel.sh gpurun_out/r04_pmc_conv1.json conv1_ scripts/conv_bench.py --iters 3 > gpurun_out/r04_pmc_conv1.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_pmc_conv1.json"))
for k, v in d["summary"].items():
    print(k[:70])
    for c, x in sorted(v.items()):
        print("   ", c, round(x["mean"], 1))
PY
