#!/bin/bash
# Round-3 evidence run (one gpurun call): rocprofv3 stats + timeline of bench.py (eager and graph replay), PMC passes for the
# feed-forward kernels (MFMA busy) and for the dense GEMM family (HBM bytes), micro-benchmarks, the bench lines.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_round3_evidence.sh [tag]'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r03}
O=gpurun_out
mkdir -p $O
scripts/gpu_profile2.sh ${T}final 8 --eager > $O/${T}final_profile.log 2>&1; tail -2 $O/${T}final_profile.log
scripts/gpu_profile2.sh ${T}final_graph 8 > $O/${T}final_graph_profile.log 2>&1; tail -2 $O/${T}final_graph_profile.log
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE" \
  scripts/pmc_kernel.sh $O/${T}_pmc_ffn_fused.json ffn_pair scripts/ffn_bench.py --rows 28800 --iters 5 > $O/${T}_pmc_ffn.log 2>&1
grep -E "ffn_pair|MFMA|_us_pass0|GRBM" $O/${T}_pmc_ffn.log | head -30
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  scripts/pmc_kernel.sh $O/${T}_pmc_gemm.json dense_gemm_kernel_v3 bench.py --eager --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/${T}_pmc_gemm.log 2>&1
grep -E "dense_gemm|FETCH|WRITE|MFMA" $O/${T}_pmc_gemm.log | head -24
timeout 300 python scripts/ffn_bench.py --rows 28800,9600 --out $O/${T}_ffn_bench.json > $O/${T}_ffn_bench.log 2>&1
timeout 300 python scripts/conv_bench.py --out $O/${T}_conv_bench.json > $O/${T}_conv_bench.log 2>&1
timeout 300 python scripts/gemm_iso.py > $O/${T}_gemm_iso.json 2>/dev/null
timeout 400 python bench.py > $O/${T}_bench_final.json 2> $O/${T}_bench_final.err
tail -1 $O/${T}_bench_final.json | cut -c1-400
timeout 300 python bench.py --eager --no-cpu-baseline > $O/${T}_bench_eager.json 2>/dev/null
tail -1 $O/${T}_bench_eager.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', d['ms_per_step'], d['host_issue_ms_per_step'])"
