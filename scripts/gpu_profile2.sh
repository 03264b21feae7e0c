#!/bin/bash
# rocprofv3 kernel trace of the train-step benchmark -> gpurun_out/<tag>_kernel_stats.csv + <tag>_timeline.json
#   usage: scripts/gpu_profile2.sh <tag> [steps] [extra bench args...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
TAG=${1:-prof}; STEPS=${2:-6}; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run -- python $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --roofline-steps 0 "$@" > $REPO/gpurun_out/${TAG}_prof_bench.json 2> $REPO/gpurun_out/${TAG}_prof.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
cd $REPO
python scripts/export_profile.py $DB gpurun_out/${TAG}_kernel_stats.csv $((STEPS + 2)) | tail -1
python scripts/timeline.py $DB gpurun_out/${TAG}_timeline.json 4
