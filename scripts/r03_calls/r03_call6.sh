#!/bin/bash
# Round 3, call 6: LN finalize rewrite (parity), FFN v2 forward with the residual prefetch (parity + time), PMC of the v2 kernel.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ffn.py tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "ffn or layernorm or ln" > $O/r03_tests_call6.log 2>&1
echo "tests rc=$? $(tail -n 1 $O/r03_tests_call6.log)"; grep -E "^FAILED|^ERROR|^E  " $O/r03_tests_call6.log | head -20
for v in 1 0; do
  NST_FFN_V2=$v timeout 300 python scripts/ffn_bench.py --rows 28800 --iters 30 2>/dev/null | grep -o '"fwd_p0.[01]": {"fused_us": [0-9.]*' | tr '\n' ' '; echo " V2=$v"
done
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" NST_FFN_V2=1 bash scripts/pmc_kernel.sh $O/r03_pmc_ffn_v2_raw.json ffn_pair scripts/ffn_bench.py --rows 28800 --iters 3 > $O/r03_pmc_ffn_v2.log 2>&1
grep -A22 "ffn_pair8_kernel<0, 0" $O/r03_pmc_ffn_v2.log | head -50
