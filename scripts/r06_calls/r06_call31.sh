#!/bin/bash
# call 31: attention forward at four waves per SIMD (122 registers, bf16 MI = 1) against three (148 + 16)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
echo "prev: $(NST_LIBRARY=$PWD/neurst_amd/lib/libneurst_hip_prev.so python scripts/attn_bench.py 2>/dev/null | tail -1)"
echo "new:  $(python scripts/attn_bench.py 2>/dev/null | tail -1)"
echo "prev: $(NST_LIBRARY=$PWD/neurst_amd/lib/libneurst_hip_prev.so python scripts/attn_bench.py 2>/dev/null | tail -1)"
echo "new:  $(python scripts/attn_bench.py 2>/dev/null | tail -1)"
AB_NO_FFN=1 bash scripts/ab_libs.sh c31 3
