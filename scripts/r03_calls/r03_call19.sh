#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for p in 0.0 0.1; do ATTN_BENCH_P=$p timeout 300 python scripts/attn_bench.py 2>&1 | tail -1; done
for m in 1 2; do NST_ATTN_MI=$m ATTN_BENCH_P=0.1 timeout 300 python scripts/attn_bench.py 2>&1 | tail -1; done
