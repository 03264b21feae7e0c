#!/bin/bash
# Round 4, call 35: secondary bench lines, native exchange, RCCL soak; HBM probe with non-zero fills
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
timeout 120 python scripts/hbm_probe.py --out gpurun_out/r04_hbm_probe.json 2>/dev/null | cut -c1-600
timeout 300 python scripts/conv_bench.py --iters 20 --out gpurun_out/r04_conv_bench.json > gpurun_out/r04_conv_bench.log 2>&1; tail -9 gpurun_out/r04_conv_bench.log
bash scripts/gpu_round4_secondary.sh r04
