#!/bin/bash
# Round 4, call 20: is the 13.4 ms of call 19 the box or the code (previous commit in a worktree, alternating); comm probe
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=$PWD/gpurun_out/r04
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do
  echo "HEAD                      $(step) ms/step"
  echo "HEAD, patch dgrad         $(NST_CONV2_DGRAD_G256=0 step) ms/step"
  [ -d _prev ] && echo "previous commit           $(cd _prev && step) ms/step"
done | tee $O/c20_ab_commit.log
NST_DIST_FORCE=1 timeout 300 python scripts/comm_probe.py 2>&1 | grep -v "^\[W\|amdgpu.ids" | tee $O/c20_comm_probe.log
rocm-smi --showclocks --showpower 2>/dev/null | head -30 > $O/c20_smi.log
