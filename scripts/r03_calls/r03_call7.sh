#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
for r in 1 2 3; do for m in 0 1 2; do
  NST_FFN_ROT=$m NST_FFN_V2=1 timeout 300 python scripts/ffn_bench.py --rows 28800 --iters 30 2>/dev/null | grep -o '"fwd_p0.[01]": {"fused_us": [0-9.]*' | tr '\n' ' '; echo " ROT=$m round $r"
done; done 2>&1 | tee gpurun_out/r03_ffn_v2_rot.log
