#!/bin/bash
# call 28: the benchmark loop on the step's own stream against calls from the default stream
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  echo "caller stream: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 --caller-stream 2>/dev/null | ms)   step stream: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>/dev/null | ms)"
done | tee gpurun_out/r06/c28_loop_stream.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r06/c28_bench_full.json 2> gpurun_out/r06/c28_bench_full.err; tail -1 gpurun_out/r06/c28_bench_full.json | cut -c1-300
NST_DIST_FORCE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 2>/dev/null | ms
NST_DIST_BACKEND=gloo NST_BENCH_HANG_DUMP_S=240 timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --roofline-steps 1 2>&1 | grep -E '^\{' | cut -c1-200
