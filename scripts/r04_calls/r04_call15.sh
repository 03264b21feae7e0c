#!/bin/bash
# Round 4, call 15: attention forward, one 8-wave workgroup per head (attn_fwd_head8_kernel)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "attention" > $O/c15_attn_tests.log 2>&1
echo "attention tests rc=$? $(tail -n 1 $O/c15_attn_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c15_attn_tests.log | head
for v in 1 0 1 0; do
  NST_ATTN_FUSED_FWD=$v timeout 300 python scripts/attn_bench.py 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fused_fwd $v', {k: round(v, 1) for k, v in d.items() if k.endswith('_us')})"
done | tee $O/c15_attn_bench.log
for r in 1 2; do for v in 0 1; do
  ms=$(NST_ATTN_FUSED_FWD=$v timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
  echo "NST_ATTN_FUSED_FWD=$v  $ms ms/step"
done; done | tee $O/c15_ab_step.log
