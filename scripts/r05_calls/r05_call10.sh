#!/bin/bash
# Round 5, call 10: long-K / few-tile GEMMs on the four-stage v2 block against the stream kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
for v in 0 256 512; do NST_GEMM_V2L=$v timeout 300 python scripts/gemm_longk_bench.py 2>/dev/null | tail -n 1 > $O/c10_longk_$v.json; done
python - <<'PY'
import json
r={v: json.load(open(f"gpurun_out/r05/c10_longk_{v}.json")) for v in (0,256,512)}
for k in r[0]:
    if k!="NST_GEMM_V2L": print(f"{k:44s} stream {r[0][k]:8.2f} us   v2l<=256 {r[256][k]:8.2f} us   v2l<=512 {r[512][k]:8.2f} us")
PY
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" --tb=short 2>&1 | tail -3
