#!/bin/bash
# Round 5, call 25: xent_reduce with eight lanes per row, 16-byte transpose launch: tests + step trace; PMC of the conv1 kernels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_ffn.py tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "transpos or xent or train_steps or ls_xent" 2>&1 | grep -v "amdgpu.ids" | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py -q -m gpu --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -2
scripts/gpu_profile2.sh r05c25 8 > $O/c25_profile.log 2>&1
grep -E "xent_reduce|transpose_bf16|TOTAL" gpurun_out/r05c25_kernel_stats.csv | awk -F, '{print substr($1,1,40),$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
tail -1 gpurun_out/r05c25_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
  scripts/pmc_kernel.sh gpurun_out/r05_pmc_conv1.json conv1_ scripts/conv_bench.py --iters 3 > gpurun_out/r05_pmc_conv1.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r05_pmc_conv1.json')); print(json.dumps(d)[:1500])"
