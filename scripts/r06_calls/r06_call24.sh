#!/bin/bash
# call 24: decoder group on the weight-gradient stream as the default (rule): graph / model / multi tests, bench lines
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_model.py tests/test_gpu_multi.py tests/test_zz_gpu_widening.py -m gpu -q -x --tb=short > gpurun_out/r06/c24_pytest.log 2>&1; tail -n 2 gpurun_out/r06/c24_pytest.log
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for i in 1 2 3; do echo "default $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 2>/dev/null | ms)"; done
echo "eager $(python bench.py --eager --no-cpu-baseline --roofline-steps 0 2>/dev/null | ms)"
echo "forced exchange $(NST_DIST_FORCE=1 python bench.py --no-cpu-baseline --roofline-steps 0 2>/dev/null | ms)"
echo "text base $(python scripts/bench_text.py --model transformer_base --batch 256 2>/dev/null | ms)"
