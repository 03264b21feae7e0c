#!/bin/bash
# Round 3, call 4: ablations of the eight-wave feed-forward forward kernel (no dropout, M = 28800), interleaved rounds.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
for r in 1 2; do
for d in 0 1 2 4 8 12 16 32 31 64; do
  us=$(NST_FFN_DBG=$d NST_FFN_V2=1 timeout 120 python scripts/ffn_bench.py --rows 28800 --iters 30 2>/dev/null | grep -o '"fwd_p0.0": {"fused_us": [0-9.]*' | grep -o '[0-9.]*$')
  echo "round $r DBG=$d fwd_p0.0 fused_us=$us"
done
done 2>&1 | tee $O/r03_ffn_v2_ablation.log
