#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_widening.py tests/test_gpu_graph.py tests/test_gpu_multi.py -m gpu -q --tb=short -x > $O/r03_tests_call13.log 2>&1
echo "model tests rc=$? $(tail -n 1 $O/r03_tests_call13.log)"; grep -E "^FAILED|^ERROR|^E  " $O/r03_tests_call13.log | head
bash scripts/ab_env.sh 3 NST_DEC_KV_GROUP 0 1 -- --steps 20 --warmup 5 2>&1 | tee $O/r03_ab_dec_kv_group.log
