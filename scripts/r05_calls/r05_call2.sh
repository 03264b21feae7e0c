#!/bin/bash
# Round 5, call 2: the add + LayerNorm kernel test with its failure text
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "add_layernorm" --tb=short 2>&1 | grep -v "amdgpu.ids" > $O/c2_pytest_ln.log
tail -60 $O/c2_pytest_ln.log
