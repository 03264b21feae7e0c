#!/bin/bash
# Round 4, call 37: final evidence at the final schedule (profiles, PMC passes, benches, GPU suite, secondary numbers, soak)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
bash scripts/gpu_round4_evidence.sh r04
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r04_gpu_tests_final.log 2>&1
echo "gpu suite rc=$? $(grep -E 'passed|failed' gpurun_out/r04_gpu_tests_final.log | tail -n 1)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/gpu_round4_secondary.sh r04
