#!/bin/bash
# full GPU suite with a kept log (gpurun_out/r03_pytest_final.log -> profiles/r03_gpu_pytest_final.log by collect_round3_profiles.sh)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
git_rev=$(cat .git_rev 2>/dev/null)
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r03_pytest_final.log 2>&1
tail -25 gpurun_out/r03_pytest_final.log
