#!/bin/bash
# Round-2 evidence run (one gpurun call): rocprofv3 stats + timeline of bench.py (eager), PMC passes for the fused FFN kernels
# (MFMA busy) and for the dense GEMM family (HBM bytes), the FFN micro-benchmark, the bench line itself.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
scripts/gpu_profile2.sh r02final 8 > gpurun_out/r02final_profile.log 2>&1
tail -3 gpurun_out/r02final_profile.log
# FFN kernels: MFMA busy cycles / wave cycles, LDS conflicts, HBM bytes  (micro-benchmark at the benchmark shape)
PMC_GROUPS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE" \
  scripts/pmc_kernel.sh gpurun_out/r02_pmc_ffn_fused.json ffn_pair_kernel scripts/ffn_bench.py --rows 28800 --iters 5 > gpurun_out/r02_pmc_ffn.log 2>&1
tail -40 gpurun_out/r02_pmc_ffn.log | grep -E "ffn_pair|MFMA|WAVE_CYCLES|FETCH|WRITE|_us_pass0|GRBM"
# dense GEMM family inside the step: HBM bytes per step
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  scripts/pmc_kernel.sh gpurun_out/r02_pmc_gemm.json dense_gemm_kernel_v3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/r02_pmc_gemm.log 2>&1
grep -E "dense_gemm|FETCH|WRITE|MFMA" gpurun_out/r02_pmc_gemm.log | head -20
timeout 300 python scripts/ffn_bench.py --rows 28800,9600 --ablate --out gpurun_out/r02_ffn_bench.json > gpurun_out/r02_ffn_bench.log 2>&1
timeout 300 python scripts/conv_bench.py --out gpurun_out/r02_conv_bench.json > gpurun_out/r02_conv_bench.log 2>&1
# graph-mode kernel trace (no host gaps): per-launch timeline of one step
scripts/gpu_profile2.sh r02final_graph 8 --graph > gpurun_out/r02final_graph_profile.log 2>&1
timeout 300 python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err
tail -1 gpurun_out/r02_bench_final.json | cut -c1-400
timeout 300 python bench.py --graph --no-cpu-baseline > gpurun_out/r02_bench_graph.json 2>/dev/null
tail -1 gpurun_out/r02_bench_graph.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', d['ms_per_step'], d['host_issue_ms_per_step'])"
