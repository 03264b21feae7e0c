#!/bin/bash
# the forced one-rank exchange path (graph replay, RCCL) several times in a row: every run must print its JSON line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do
  NST_DIST_FORCE=1 timeout 200 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 5 --warmup 3 > gpurun_out/forced_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -c '^{' gpurun_out/forced_$i.log) json line(s) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/forced_$i.log)"
  if [ $rc -ne 0 ]; then tail -25 gpurun_out/forced_$i.log; fi
done
