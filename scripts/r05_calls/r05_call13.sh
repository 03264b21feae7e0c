#!/bin/bash
# Round 5, call 13: kernel stats of the transformer_big step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
REPO=$PWD
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_big
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_big -o run -- python $REPO/scripts/bench_text.py --model transformer_big --batch 256 --steps 6 --warmup 2 > $REPO/gpurun_out/r05/c13_big_prof_bench.json 2> $REPO/gpurun_out/r05/c13_big_prof.err
DB=$(find /tmp/prof_big -name "*.db" | head -1)
cd $REPO
python scripts/export_profile.py $DB gpurun_out/r05/c13_big_kernel_stats.csv 10 | tail -1
head -24 gpurun_out/r05/c13_big_kernel_stats.csv | cut -c1-150,200-280
