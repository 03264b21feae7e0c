#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for r in 0 5 51 52 54 58 53 55 57 515; do NST_GEMM_RING=$r timeout 300 python scripts/gemm_iso.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('RING=$r', {k.split('[')[0].strip()+k.split(']')[1]:round(v,1) for k,v in d['us'].items() if 'wgrad' in k})"; done
