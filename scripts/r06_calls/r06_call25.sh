#!/bin/bash
# call 25: same-box A/B of the decoder group's placement in graph and eager mode (module constant pinned through a wrapper)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
run() { python -c "
import sys, runpy
import neurst_amd.models.encoder_decoder_model as M
M._WGRAD_DECODER_SIDE = $1
sys.argv = ['bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--roofline-steps', '0'] + '$2'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | ms; }
for i in 1 2 3; do
  echo "graph: end $(run False '')  dec-side $(run True '')    eager: end $(run False --eager)  dec-side $(run True --eager)"
done | tee gpurun_out/r06/c25_group_dec_ab.log
