#!/bin/bash
# Round 4, call 18: conv2 data gradient on the 256 x 256 tile core (parity, stand-alone and in-step A/B); exchange placement A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "conv2" > $O/c18_conv2_tests.log 2>&1
echo "conv2 tests rc=$? $(tail -n 1 $O/c18_conv2_tests.log)"; grep -E "^FAILED|^ERROR|^E  " $O/c18_conv2_tests.log | head
for v in 0 1; do
  NST_CONV2_DGRAD_G256=$v timeout 300 python scripts/conv_bench.py --iters 20 --out $O/c18_conv_bench_g$v.json 2>&1 | grep -i "dgrad\|conv2" | head -8
done
step() { timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))'; }
for r in 1 2 3; do for v in 0 1; do
  echo "NST_CONV2_DGRAD_G256=$v  $(NST_CONV2_DGRAD_G256=$v step) ms/step"
done; done | tee $O/c18_ab_step.log
# forced one-rank exchange over RCCL: where the grouped weight gradients (and with them the big buckets) are launched
for r in 1 2; do
  for cfg in "end 32 16" "end 256 16" "end 256 64" "encoder 32 16" "encoder 256 16" "encoder 256 64"; do
    set -- $cfg
    ms=$(NST_DIST_FORCE=1 NST_WGRAD_GROUP_AT=$1 NST_DIST_BUCKET_MB=$2 NCCL_MAX_NCHANNELS=$3 step)
    echo "forced exchange: group at $1, bucket $2 MB, channels $3: $ms ms/step"
  done
done | tee $O/c18_ab_exchange.log
