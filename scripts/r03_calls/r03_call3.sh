#!/bin/bash
# Round 3, call 3: the eight-wave feed-forward forward kernel -- parity, then timing against the one-wave-per-SIMD kernel.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ffn.py -m gpu -q --tb=short -x > $O/r03_ffn_tests_call3.log 2>&1
echo "ffn tests rc=$? $(tail -n 1 $O/r03_ffn_tests_call3.log)"; grep -E "^FAILED|^ERROR|^E  " $O/r03_ffn_tests_call3.log | head -20
for v in 1 0 1 0; do
  NST_FFN_V2=$v timeout 300 python scripts/ffn_bench.py --rows 28800 --iters 30 > $O/r03_ffn_bench_v2_$v.log 2>&1
  echo "NST_FFN_V2=$v"; grep -o '"fwd_p0.[0-9]*": {[^}]*}' $O/r03_ffn_bench_v2_$v.log | cut -c1-220
done
NST_FFN_DBG=1 NST_FFN_V2=1 timeout 300 python scripts/ffn_bench.py --rows 28800 --iters 30 2>&1 | grep -o '"fwd_p0.0": {[^}]*}' | cut -c1-200
