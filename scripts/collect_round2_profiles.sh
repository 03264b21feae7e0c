#!/bin/bash
# copies what scripts/gpu_round2_evidence.sh left under gpurun_out/ into profiles/ (tracked) and rebuilds the PMC summaries
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cp $G/r02final_kernel_stats.csv $P/r02_rocprofv3_kernel_stats_final.csv
cp $G/r02final_timeline.json $P/r02_timeline_eager_final.json
cp $G/r02final_graph_kernel_stats.csv $P/r02_rocprofv3_kernel_stats_graph_final.csv
cp $G/r02final_graph_timeline.json $P/r02_timeline_graph_final.json
cp $G/r02final_graph_timeline_trace.csv $P/r02_step_trace_graph_final.csv
cp $G/r02_ffn_bench.json $P/r02_ffn_bench_ablation.json
cp $G/r02_conv_bench.json $P/r02_conv_bench.json
cp $G/r02_pmc_ffn_fused.json $P/r02_pmc_ffn_fused_raw.json
cp $G/r02_pmc_gemm.json $P/r02_pmc_gemm_raw.json
tail -1 $G/r02_bench_final.json > $P/r02_bench_bf16_final.json
tail -1 $G/r02_bench_graph.json > $P/r02_bench_bf16_graph_final.json
[ -s $G/r02_bench_forced_exchange.json ] && cp $G/r02_bench_forced_exchange.json $P/r02_bench_bf16_forced_exchange_path.json
cp $G/model_report.json $P/r02_model_parity_report.json
cp $G/r02_pytest_final.log $P/r02_gpu_pytest_final.log
python scripts/summarise_pmc.py > /dev/null
python scripts/kernel_resources.py r02 > /dev/null 2>&1
python - <<'PY'
import json
d = json.load(open('profiles/r02_bench_bf16_final.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'host_issue_ms_per_step', 'model_tflops_per_s', 'model_mfma_frac')})
r = d['roofline']; print({k: r[k] for k in ('ms_per_step', 'achieved', 'frac', 'traffic', 'launches_per_step')})
for k, v in d['roofline_families'].items(): print(k, round(v['ms_per_step'], 3), round(v['frac'], 3))
g = json.load(open('profiles/r02_bench_bf16_graph_final.json')); print('graph', g['ms_per_step'], g['host_issue_ms_per_step'])
f = json.load(open('profiles/r02_bench_bf16_forced_exchange_path.json')); print('forced', f['ms_per_step'], f['reducer_messages_per_step'])
p = json.load(open('profiles/r02_pmc_gemm.json')); print({k: p[k] for k in ('hbm_bytes_per_step', 'hbm_bytes_per_launch', 'mfma_utilisation_in_step')})
print(d['cpu_baseline'])
PY
