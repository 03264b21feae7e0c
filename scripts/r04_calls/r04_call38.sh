#!/bin/bash
# Round 4, call 38: the full GPU suite, smoke() and one bench line on the final tree
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r04_gpu_tests_final.log 2>&1
echo "gpu suite rc=$? $(grep -E 'passed|failed' gpurun_out/r04_gpu_tests_final.log | tail -n 1)"; grep -E "^FAILED|^ERROR" gpurun_out/r04_gpu_tests_final.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | cut -c1-330
