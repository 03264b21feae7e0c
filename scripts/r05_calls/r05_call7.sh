#!/bin/bash
# Round 5, call 7: kernel-level A/B (rocprofv3 kernel stats of the graph-replayed step) of the recomputed ReLU gate
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
for v in 0 1; do
  NST_LN_REGATE=$v scripts/gpu_profile2.sh r05c7_regate$v 8 > gpurun_out/r05/c7_profile_regate$v.log 2>&1
  tail -1 gpurun_out/r05/c7_profile_regate$v.log
  grep -E "ln_bwd_wide|ln_fwd_wide|attn_fwd|TOTAL" gpurun_out/r05c7_regate${v}_kernel_stats.csv | cut -c1-200
done
