#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export NST_GEMM_RING=0
for u in 256 512 1024; do NST_WGRAD_UNITS=$u timeout 300 python scripts/gemm_iso.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('UNITS=$u', {k.split('[')[0].strip()+k.split(']')[1]:round(v,1) for k,v in d['us'].items() if 'wgrad' in k})"; done
