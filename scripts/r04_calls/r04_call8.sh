#!/bin/bash
# Round 4, call 8: whole GPU tier; soak of the forced exchange path over RCCL in graph mode (10 runs x 200 steps); step A/B of
# the group's launch point and the padded logits rows
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1800 python -m pytest tests -m gpu -q --tb=short --durations=10 > $O/c8_gpu_tests.log 2>&1
echo "gpu tests rc=$? $(tail -n 1 $O/c8_gpu_tests.log)"; grep -E "^FAILED|^ERROR|^E  |SKIP" $O/c8_gpu_tests.log | head -20
: > $O/c8_graph_rccl_soak.log
for i in $(seq 1 10); do
  NST_DIST_FORCE=1 timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --roofline-steps 0 > $O/c8_soak_$i.json 2> $O/c8_soak_$i.err
  rc=$?
  echo "run $i rc=$rc $(grep '^{' $O/c8_soak_$i.json | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step graph" if d.get("graph_replay", d.get("graph")) else "ms/step", "messages", d.get("reducer_messages_per_step"), "rccl_world", d.get("rccl_world_size"))' 2>/dev/null)" | tee -a $O/c8_graph_rccl_soak.log
  [ $rc -ne 0 ] && tail -5 $O/c8_soak_$i.err | tee -a $O/c8_graph_rccl_soak.log
done
for r in 1; do
  for cfg in "NST_WGRAD_GROUP_AT=end" "NST_WGRAD_GROUP_AT=encoder"; do
    ms=$(env $cfg timeout 300 python bench.py --no-cpu-baseline --roofline-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')
    echo "$cfg  $ms ms/step"
  done
done | tee $O/c8_ab_step.log
