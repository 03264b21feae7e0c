#!/bin/bash
# Round-end evidence run: full GPU parity suite, all kernel micro-benchmarks, the train-step bench (default N=1 line),
# and a rocprofv3 kernel trace of the same command.    usage: scripts/gpu_final.sh <tag>
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
TAG=${1:-final}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/${TAG}_tests.log 2>&1
echo "gpu tests rc=$? $(tail -n 1 gpurun_out/${TAG}_tests.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 900 python scripts/kernel_bench.py > gpurun_out/${TAG}_kb.log 2>&1; cp gpurun_out/kernel_bench_bf16.json gpurun_out/${TAG}_kernel_bench_bf16.json
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -n 1 gpurun_out/${TAG}_bench.json | cut -c1-300
(cd /tmp && rm -rf /tmp/prof_$TAG && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/${TAG}_prof_bench.json 2> $REPO/gpurun_out/${TAG}_prof.err)
python scripts/export_profile.py $(find /tmp/prof_$TAG -name "*.db" | head -1) gpurun_out/${TAG}_kernel_stats.csv 8
(cd /tmp && rm -rf /tmp/prof1_$TAG && NST_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof1_$TAG -o run -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/${TAG}_prof1_bench.json 2> $REPO/gpurun_out/${TAG}_prof1.err)
python scripts/export_profile.py $(find /tmp/prof1_$TAG -name "*.db" | head -1) gpurun_out/${TAG}_kernel_stats_single_stream.csv 8
python scripts/step_breakdown.py 2>&1 | tail -n 1
