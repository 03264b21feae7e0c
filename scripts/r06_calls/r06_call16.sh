#!/bin/bash
# call 16: fixed / per-chunk cost model of the feed-forward pair and its stage ablation (ablation twin of the library)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
timeout 600 python scripts/ffn_cost_model.py r06 > gpurun_out/r06/c16_cost_model.log 2>&1; tail -n 40 gpurun_out/r06/c16_cost_model.log | cut -c1-300
NST_LIBRARY=$PWD/neurst_amd/lib/libneurst_hip_ablation.so timeout 900 python scripts/ffn_ablation.py r06 > gpurun_out/r06/c16_ablation.log 2>&1; tail -n 20 gpurun_out/r06/c16_ablation.log | cut -c1-400
