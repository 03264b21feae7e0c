#!/bin/bash
# usage: scripts/gpu_pmc.sh <pmc_target arg> <kernel substring> <counter group> [<counter group> ...]   (a group = "A B C")
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
TARGET=$1; FILT=$2; shift 2
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$i -- python $REPO/scripts/pmc_target.py $TARGET > /tmp/pmc_$i.log 2>&1)
  python scripts/pmc_summary.py /tmp/pmc_$i "$FILT" || tail -5 /tmp/pmc_$i.log
done
