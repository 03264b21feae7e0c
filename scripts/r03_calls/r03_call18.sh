#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export NST_GEMM_RING=0
export PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
bash scripts/pmc_kernel.sh gpurun_out/r03_pmc_ffn1_wgrad_tcc.json dense_gemm_kernel_v3 scripts/pmc_target.py ffn1_wgrad 2>&1 | tail -16
