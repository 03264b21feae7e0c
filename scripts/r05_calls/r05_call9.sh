#!/bin/bash
# Round 5, call 9: full GPU suite + smoke + default bench after the switch clean-up
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "amdgpu.ids" > $O/c9_gpu_pytest.log
tail -6 $O/c9_gpu_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 600 python bench.py > $O/c9_bench.json 2> $O/c9_bench.err; tail -n 1 $O/c9_bench.json | cut -c1-400
