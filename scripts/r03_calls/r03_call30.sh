#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or wgrad or split or colsum" 2>&1 | tail -3
for r in 0 1; do NST_GEMM_KS=$r timeout 300 python scripts/gemm_iso.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('KS=$r', {k.split('[')[0].strip()+k.split(']')[1][:9]:round(v,1) for k,v in d['us'].items() if 'wgrad' in k})"; done
for r in 0 1; do echo KS=$r; NST_GEMM_KS=$r timeout 300 python scripts/kernel_bench.py --only gemm --tag ks$r 2>&1 | grep -E "dec_ffn|front_dense|logits|ffn1.wgrad|ffn2.wgrad|attn_out.wgrad"; done
bash scripts/ab_env.sh 2 NST_GEMM_KS 0 1 -- --steps 20 --warmup 5
