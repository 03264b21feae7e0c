#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ffn.py -m gpu -q --tb=short -x > gpurun_out/r03_ffn_tests_call8.log 2>&1
echo "ffn tests rc=$? $(tail -n 1 gpurun_out/r03_ffn_tests_call8.log)"; grep -E "^FAILED|^ERROR|^E  " gpurun_out/r03_ffn_tests_call8.log | head
for r in 1 2 3; do for v in 1 0; do
  NST_FFN_V2=$v timeout 300 python scripts/ffn_bench.py --rows 28800 --iters 30 2>/dev/null | grep -o '"fwd_p0.[01]": {"fused_us": [0-9.]*' | tr '\n' ' '; echo " V2=$v round $r"
done; done 2>&1 | tee gpurun_out/r03_ffn_v2_dma_interleave.log
