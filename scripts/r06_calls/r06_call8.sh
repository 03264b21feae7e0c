#!/bin/bash
# Round 6, call 8: the whole GPU suite on the fused-row defaults; smoke; text models at the base / big shapes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q --tb=short > $O/c8_gpu_tests.log 2>&1
echo "gpu tests rc=$? $(tail -n 1 $O/c8_gpu_tests.log)"
grep -E "FAILED|ERROR" $O/c8_gpu_tests.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2 | tee $O/c8_smoke.log
