#!/bin/bash
# copies what scripts/gpu_round6_evidence.sh / gpu_round6_secondary.sh left under gpurun_out/ into profiles/ (tracked)
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles; T=${1:-r06}
cp $G/${T}final_kernel_stats.csv $P/${T}_rocprofv3_kernel_stats_final.csv
cp $G/${T}final_timeline.json $P/${T}_timeline_eager_final.json
cp $G/${T}final_graph_kernel_stats.csv $P/${T}_rocprofv3_kernel_stats_graph_final.csv
cp $G/${T}final_graph_timeline.json $P/${T}_timeline_graph_final.json
cp $G/${T}final_graph_timeline_trace.csv $P/${T}_step_trace_graph_final.csv
cp $G/${T}_ffn_bench.json $P/${T}_ffn_bench.json
cp $G/${T}_conv_bench.json $P/${T}_conv_bench.json
cp $G/${T}_gemm_iso.json $P/${T}_gemm_iso.json
cp $G/${T}_pmc_ffn_fused.json $P/${T}_pmc_ffn_fused_raw.json
cp $G/${T}_pmc_gemm.json $P/${T}_pmc_gemm_raw.json
tail -1 $G/${T}_bench_final.json > $P/${T}_bench_bf16_final.json
tail -1 $G/${T}_bench_eager.json > $P/${T}_bench_bf16_eager_final.json
for f in bench_fp32 bench_bf16_ragged bench_bf16_speech_transformer_m bench_text_transformer_base_bf16 bench_text_transformer_big_bf16 bench_text_transformer_base_bf16_s128 bench_text_transformer_big_bf16_s128 bench_decode_bf16 bench_forced_exchange bench_forced_exchange_native; do
  [ -s $G/${T}_$f.json ] && grep '^{' $G/${T}_$f.json | tail -1 > $P/${T}_$f.json
done
[ -s $G/model_report.json ] && cp $G/model_report.json $P/${T}_model_parity_report.json
[ -s $G/kernel_report_tr.json ] && cp $G/kernel_report_tr.json $P/${T}_kernel_parity_report.json
[ -s $G/${T}_pytest_final.log ] && cp $G/${T}_pytest_final.log $P/${T}_gpu_pytest_final.log
python scripts/summarise_pmc.py --tag $T ${2:+--commit $2} > /dev/null
python scripts/kernel_resources.py $T > /dev/null 2>&1
ls -la $P | grep ${T}_ | wc -l
for f in pmc_group256 pmc_attn attn_bench wgrad_group_bench pmc_whole_step pmc_conv2 pmc_ln hbm_probe pmc_rowgemm rowgemm_bench_warm rowgemm_bench_cold ffn_cost_model ffn_ablation; do
  [ -s $G/${T}_$f.json ] && cp $G/${T}_$f.json $P/${T}_$f.json
done
[ -s $G/${T}_two_rank_rehearsal_graph.log ] && cp $G/${T}_two_rank_rehearsal_graph.log $P/${T}_two_rank_rehearsal_graph_final.log
[ -s $G/${T}_kernel_bench_vs_hipblaslt.txt ] && cp $G/${T}_kernel_bench_vs_hipblaslt.txt $P/${T}_kernel_bench_vs_hipblaslt.txt
[ -s $G/${T}_graph_rccl_soak.log ] && cp $G/${T}_graph_rccl_soak.log $P/${T}_graph_rccl_soak.log
[ -s $G/${T}_gpu_tests_final.log ] && cp $G/${T}_gpu_tests_final.log $P/${T}_gpu_pytest_final.log
