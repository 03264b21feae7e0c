#!/bin/bash
# Round 3, call 5: eight-wave feed-forward forward kernel with the fragment ring: parity + ablations (interleaved rounds).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_ffn.py -m gpu -q --tb=short -x > $O/r03_ffn_tests_call5.log 2>&1
echo "ffn tests rc=$? $(tail -n 1 $O/r03_ffn_tests_call5.log)"; grep -E "^FAILED|^ERROR|^E  " $O/r03_ffn_tests_call5.log | head -20
for r in 1 2; do
for d in 0 1 2 4 8 12 16 32 31; do
  us=$(NST_FFN_DBG=$d NST_FFN_V2=1 timeout 120 python scripts/ffn_bench.py --rows 28800 --iters 30 2>/dev/null | grep -o '"fwd_p0.[01]": {"fused_us": [0-9.]*' | grep -o '[0-9.]*$' | tr '\n' ' ')
  echo "round $r DBG=$d fwd_p0.0 / fwd_p0.1 fused_us = $us"
done
done 2>&1 | tee $O/r03_ffn_v2_ablation2.log
