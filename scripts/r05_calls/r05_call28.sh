#!/bin/bash
# Round 5, call 28: LayerNorm backward of the front end with up to 2048 workgroups (512 before)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "layernorm or conv2_kernels_at or frontend" 2>&1 | grep -v "amdgpu.ids" | tail -2
for i in 1 2; do timeout 300 python scripts/conv_bench.py 2>/dev/null | grep ln_relu_bwd; done
