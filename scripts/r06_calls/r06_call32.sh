#!/bin/bash
# RECORD of a round-6 experiment: the NST_* switch(es) this script sets existed only in the working tree of that experiment
# (removed with it; the library now warns about them).  Kept for the log under profiles/r06_history/; it does not re-run.
echo "$0: record of a removed experiment (see the header); not runnable against this tree" >&2; exit 1
# call 32: attention forward against waves per SIMD (temporary NST_ATTN_OCC_FWD): 3 (compiler's choice) .. 6
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -k "attention or attn" 2>&1 | tail -n 1
for occ in 3 4 5 6 4 3; do echo "occ $occ: $(NST_ATTN_OCC_FWD=$occ python scripts/attn_bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v,1) for k,v in d.items() if k.endswith('fwd_us')})")"; done | tee gpurun_out/r06/c32_attn_occ.log
