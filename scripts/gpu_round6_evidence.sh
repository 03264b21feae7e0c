#!/bin/bash
# Round-6 evidence run (one gpurun call): rocprofv3 stats + timeline of bench.py (graph replay and eager), PMC passes for the
# feed-forward kernels (MFMA busy), the dense GEMM family (HBM bytes) and the whole step, micro-benchmarks, the bench lines.
#   gpurun --timeout 2700 -- 'bash scripts/gpu_round6_evidence.sh [tag]'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06}
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --tb=short > $O/${T}_gpu_tests_final.log 2>&1
echo "gpu tests rc=$? $(tail -n 1 $O/${T}_gpu_tests_final.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2 | tee $O/${T}_smoke.log
scripts/gpu_profile2.sh ${T}final_graph 8 > $O/${T}final_graph_profile.log 2>&1; tail -1 $O/${T}final_graph_profile.log
scripts/gpu_profile2.sh ${T}final 8 --eager > $O/${T}final_profile.log 2>&1; tail -1 $O/${T}final_profile.log
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE" \
  scripts/pmc_kernel.sh $O/${T}_pmc_ffn_fused.json ffn_pair scripts/ffn_bench.py --rows 28800 --iters 5 > $O/${T}_pmc_ffn.log 2>&1
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  scripts/pmc_kernel.sh $O/${T}_pmc_gemm.json dense_gemm_kernel_v3 bench.py --eager --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/${T}_pmc_gemm.log 2>&1
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
  scripts/pmc_kernel.sh $O/${T}_pmc_group256.json gemm256_group scripts/wgrad_group_bench.py > $O/${T}_pmc_group256.log 2>&1
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
  scripts/pmc_kernel.sh $O/${T}_pmc_attn.json attn_ scripts/attn_bench.py > $O/${T}_pmc_attn.log 2>&1
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  scripts/pmc_kernel.sh $O/${T}_pmc_conv2.json conv2_ scripts/conv_bench.py --iters 3 > $O/${T}_pmc_conv2.log 2>&1
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" \
  scripts/pmc_kernel.sh $O/${T}_pmc_ln.json ln_ bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 > $O/${T}_pmc_ln.log 2>&1
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
  scripts/pmc_kernel.sh $O/${T}_pmc_rowgemm.json rowgemm_kernel bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 > $O/${T}_pmc_rowgemm.log 2>&1
timeout 300 python scripts/ffn_cost_model.py ${T} > /dev/null 2>&1
[ -f neurst_amd/lib/libneurst_hip_ablation.so ] && NST_LIBRARY=$PWD/neurst_amd/lib/libneurst_hip_ablation.so timeout 600 python scripts/ffn_ablation.py ${T} > $O/${T}_ffn_ablation.log 2>&1
timeout 300 python scripts/rowgemm_bench.py ${T}_warm > /dev/null 2>&1; cp $O/${T}_warm_rowgemm_bench.json $O/${T}_rowgemm_bench_warm.json
timeout 300 python scripts/rowgemm_bench.py ${T}_cold --cold > /dev/null 2>&1; cp $O/${T}_cold_rowgemm_bench.json $O/${T}_rowgemm_bench_cold.json
timeout 120 python scripts/hbm_probe.py --out $O/${T}_hbm_probe.json > $O/${T}_hbm_probe.log 2>&1; tail -1 $O/${T}_hbm_probe.log | cut -c1-400
timeout 300 python scripts/ffn_bench.py --rows 28800,9600 --out $O/${T}_ffn_bench.json > $O/${T}_ffn_bench.log 2>&1
timeout 300 python scripts/conv_bench.py --out $O/${T}_conv_bench.json > $O/${T}_conv_bench.log 2>&1
timeout 300 python scripts/gemm_iso.py > $O/${T}_gemm_iso.json 2>/dev/null
timeout 300 python scripts/attn_bench.py > $O/${T}_attn_bench.json 2>/dev/null
timeout 300 python scripts/wgrad_group_bench.py > $O/${T}_wgrad_group_bench.json 2>/dev/null
timeout 400 python scripts/kernel_bench.py --only blas,gemm > $O/${T}_kernel_bench_vs_hipblaslt.txt 2>/dev/null
timeout 400 python bench.py > $O/${T}_bench_final.json 2> $O/${T}_bench_final.err
tail -1 $O/${T}_bench_final.json | cut -c1-400
STEP_MS=$(tail -1 $O/${T}_bench_final.json | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],2))')
bash scripts/pmc_whole_step.sh $T $STEP_MS > $O/${T}_pmc_whole_step_summary.log 2>&1; tail -14 $O/${T}_pmc_whole_step_summary.log
timeout 300 python bench.py --eager --no-cpu-baseline > $O/${T}_bench_eager.json 2>/dev/null
tail -1 $O/${T}_bench_eager.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['ms_per_step'], d['host_issue_ms_per_step'])"
NST_DIST_BACKEND=gloo NST_BENCH_HANG_DUMP_S=240 timeout 300 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --roofline-steps 1 > $O/${T}_two_rank_rehearsal_graph.log 2>&1
echo "two-rank rehearsal (graph replay, the default) rc=$?"; grep -E '^\{' $O/${T}_two_rank_rehearsal_graph.log | cut -c1-300
