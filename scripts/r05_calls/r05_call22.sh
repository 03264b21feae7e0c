#!/bin/bash
# Round 5, call 22: LayerNorm backward with the next pass's operands requested ahead of the stores
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "layernorm or add_layernorm" 2>&1 | grep -v "amdgpu.ids" > $O/c22_pytest.log
tail -3 $O/c22_pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_widening.py -q -m gpu --tb=short -x 2>&1 | grep -v "amdgpu.ids" > $O/c22_pytest_model.log
tail -3 $O/c22_pytest_model.log | cut -c1-300
timeout 300 python scripts/conv_bench.py 2>/dev/null | grep ln_relu
scripts/gpu_profile2.sh r05c22 8 > $O/c22_profile.log 2>&1
grep -E "ln_bwd|ln_fwd|conv1|TOTAL" gpurun_out/r05c22_kernel_stats.csv | awk -F, '{print substr($1,1,80),$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}'
tail -1 gpurun_out/r05c22_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
