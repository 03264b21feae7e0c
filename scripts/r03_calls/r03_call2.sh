#!/bin/bash
# Round 3, second GPU call: specialised GEMM epilogues -- parity (whole tier), stand-alone timings and the K sweep with and
# without them, and the interleaved A/B on the benchmark step.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short --durations=10 > $O/r03_gpu_tests_call2.log 2>&1
echo "gpu tests rc=$? $(tail -n 1 $O/r03_gpu_tests_call2.log)"; grep -E "^FAILED|^ERROR" $O/r03_gpu_tests_call2.log | head -20
for v in 0 1; do
  NST_GEMM_GENERIC_EPI=$v timeout 300 python scripts/gemm_iso.py > $O/r03_gemm_iso_generic$v.json 2> $O/r03_gemm_iso_generic$v.err; echo "iso generic=$v rc=$?"; cat $O/r03_gemm_iso_generic$v.json
  NST_GEMM_GENERIC_EPI=$v timeout 300 python scripts/gemm_iso.py --ksweep > $O/r03_gemm_ksweep_generic$v.json 2>> $O/r03_gemm_iso_generic$v.err; cat $O/r03_gemm_ksweep_generic$v.json
done
timeout 600 bash scripts/ab_env.sh 3 NST_GEMM_GENERIC_EPI 1 0 -- --steps 20 --warmup 5 > $O/r03_ab_generic_epi.log 2>&1; cat $O/r03_ab_generic_epi.log
