#!/bin/bash
# Secondary measurements with the same bench.py contract: BASELINE config #2 (fp32, 1 GPU), ragged lengths,
# speech_transformer_m, the text models at ~32 768 tokens per step (256 pairs x 64 + 64), beam-search decoding, forced exchange.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06}
O=gpurun_out
mkdir -p $O
timeout 400 python bench.py --dtype fp32 --steps 6 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $O/${T}_bench_fp32.json 2>/dev/null; tail -1 $O/${T}_bench_fp32.json | cut -c1-200
timeout 300 python bench.py --ragged --no-cpu-baseline --roofline-steps 0 > $O/${T}_bench_bf16_ragged.json 2>/dev/null; tail -1 $O/${T}_bench_bf16_ragged.json | cut -c1-200
timeout 300 python bench.py --model speech_transformer_m --no-cpu-baseline --roofline-steps 0 > $O/${T}_bench_bf16_speech_transformer_m.json 2>/dev/null; tail -1 $O/${T}_bench_bf16_speech_transformer_m.json | cut -c1-200
timeout 300 python scripts/bench_text.py --model transformer_base --batch 256 > $O/${T}_bench_text_transformer_base_bf16.json 2>/dev/null; tail -1 $O/${T}_bench_text_transformer_base_bf16.json | cut -c1-300
timeout 300 python scripts/bench_text.py --model transformer_big --batch 256 > $O/${T}_bench_text_transformer_big_bf16.json 2>/dev/null; tail -1 $O/${T}_bench_text_transformer_big_bf16.json | cut -c1-300
# the text models also at the sentence length SURVEY 8(d) names (256 pairs x 128 + 128 = 65 536 tokens per step; BASELINE config #4 "stresses long-seq attention tiles")
timeout 300 python scripts/bench_text.py --model transformer_base --batch 256 --src-len 128 --trg-len 128 > $O/${T}_bench_text_transformer_base_bf16_s128.json 2>/dev/null; tail -1 $O/${T}_bench_text_transformer_base_bf16_s128.json | cut -c1-300
timeout 300 python scripts/bench_text.py --model transformer_big --batch 256 --src-len 128 --trg-len 128 > $O/${T}_bench_text_transformer_big_bf16_s128.json 2>/dev/null; tail -1 $O/${T}_bench_text_transformer_big_bf16_s128.json | cut -c1-300
timeout 300 python scripts/bench_decode.py --graphs > $O/${T}_bench_decode_bf16.json 2>/dev/null; tail -1 $O/${T}_bench_decode_bf16.json | cut -c1-300
# the exchange path over RCCL with ONE forced rank (both carriers): the line carries the per-step exchange timings (HIP events)
NST_DIST_FORCE=1 NST_DIST_NATIVE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>/dev/null | grep '^{' | tail -1 > $O/${T}_bench_forced_exchange.json; cut -c1-200 $O/${T}_bench_forced_exchange.json
NST_DIST_FORCE=1 NST_DIST_NATIVE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>/dev/null | grep '^{' | tail -1 > $O/${T}_bench_forced_exchange_native.json; cut -c1-200 $O/${T}_bench_forced_exchange_native.json
