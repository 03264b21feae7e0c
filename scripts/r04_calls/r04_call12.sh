#!/bin/bash
# Round 4, call 12: PMC counters of the per-head attention backward (where does its time go)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM;GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES" \
  scripts/pmc_kernel.sh gpurun_out/r04/c12_pmc_attn_head8.json attn_bwd_head8 scripts/attn_bench.py > gpurun_out/r04/c12_pmc.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/c12_pmc_attn_head8.json"))
for name, k in d["summary"].items():
    print(name[:60])
    for c, v in sorted(k.items()):
        print("   ", c, round(v["mean"], 1), v["n"])
PY
