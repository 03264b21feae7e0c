#!/bin/bash
# Round 5, call 14: kernel stats of the fp32 step (BASELINE config #2)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
scripts/gpu_profile2.sh r05c14_fp32 4 --dtype fp32 > gpurun_out/r05/c14_profile_fp32.log 2>&1
head -22 gpurun_out/r05c14_fp32_kernel_stats.csv | cut -c1-130,180-260
tail -1 gpurun_out/r05c14_fp32_kernel_stats.csv
