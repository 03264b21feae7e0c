#!/bin/bash
# Round 4, call 33: final evidence run (profiles, PMC passes, micro-benchmarks, bench lines) + the full GPU suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
bash scripts/gpu_round4_evidence.sh r04
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r04_gpu_tests_final.log 2>&1
echo "gpu suite rc=$? $(grep -E 'passed|failed' gpurun_out/r04_gpu_tests_final.log | tail -n 1)"
