#!/bin/bash
# Round 5, call 11: bench.py --gpus 2 rehearsal on the one GPU (host-staged gloo): autotune children, exchange diagnostics, JSON fields;
# and the forced one-rank exchange path over RCCL (both carriers)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
NST_DIST_BACKEND=gloo NST_BENCH_HANG_DUMP_S=400 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --roofline-steps 1 --autotune-budget 200 > $O/c11_two_rank_rehearsal.log 2>&1
echo "two-rank rehearsal rc=$?"; grep -E '^\{' $O/c11_two_rank_rehearsal.log | tail -1 | python -c '
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ("ms_per_step","n_gpus","rccl_world_size","dist_backend","reducer_messages_per_step","exchange","hw_queues_note")})
print("autotune:", json.dumps(d["autotune"])[:1500])
print("env:", d["env"])'
tail -5 $O/c11_two_rank_rehearsal.log | cut -c1-300
for nat in 0 1; do
  NST_DIST_FORCE=1 NST_DIST_NATIVE=$nat timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>/dev/null | grep '^{' | tail -1 > $O/c11_forced_exchange_native$nat.json
  python -c "
import json; d=json.load(open('gpurun_out/r05/c11_forced_exchange_native$nat.json')); print('forced native=$nat', round(d['ms_per_step'],3), d['exchange'])"
done
