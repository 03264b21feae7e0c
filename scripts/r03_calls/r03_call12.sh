#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
bash scripts/ab_env.sh 2 NST_WGRAD_UNITS 256 384 512 -- --steps 20 --warmup 5 2>&1 | tee gpurun_out/r03_ab_wgrad_units.log
bash scripts/ab_env.sh 2 NST_WGRAD_UNITS_SMALL 128 256 -- --steps 20 --warmup 5 2>&1 | tee -a gpurun_out/r03_ab_wgrad_units.log
bash scripts/ab_env.sh 2 NST_GRAPH_MIN_DEFERRED 3 6 12 -- --steps 20 --warmup 5 2>&1 | tee -a gpurun_out/r03_ab_wgrad_units.log
