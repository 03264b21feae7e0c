#!/bin/bash
# First GPU call of the next round:  gpurun --timeout 1500 -- 'bash scripts/gpu_round2_first_call.sh'
#  1. the 28 parity cases added at the end of round 1 without GPU time (tests/test_zz_gpu_widening.py), full tracebacks kept;
#  2. the whole -m gpu tier (what the driver runs), 3. smoke(), 4. the headline bench line.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zz_gpu_widening.py -m gpu -q --tb=long > gpurun_out/r02_zz_tests.log 2>&1
echo "zz tests rc=$? $(tail -n 1 gpurun_out/r02_zz_tests.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r02_zz_tests.log | head -30
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r02_gpu_tests.log 2>&1
echo "all gpu tests rc=$? $(tail -n 1 gpurun_out/r02_gpu_tests.log)"; grep -E "^FAILED|^ERROR" gpurun_out/r02_gpu_tests.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; echo "smoke rc=$? $(tail -n 1 gpurun_out/r02_smoke.log)"
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -n 1 gpurun_out/r02_bench.json | cut -c1-600
