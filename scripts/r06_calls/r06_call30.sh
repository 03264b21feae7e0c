#!/bin/bash
# call 30: soak of the final schedule: 300 graph-replayed steps, 300 with a forced one-rank exchange; driver's command line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('soak', d['ms_per_step'], d['final_loss'])"
NST_DIST_FORCE=1 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('soak forced exchange', d['ms_per_step'], d['final_loss'], d.get('exchange'))"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/c30_driver_line.json 2> gpurun_out/r06/c30_driver_line.err; tail -1 gpurun_out/r06/c30_driver_line.json | cut -c1-420
