#!/bin/bash
# Round 5, call 12: 256 x 256 forward / input-gradient GEMM: parity, text-model benches
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" --tb=short 2>&1 | grep -v "amdgpu.ids" > $O/c12_pytest.log
tail -15 $O/c12_pytest.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_zz_gpu_widening.py -q -m gpu -k "text or seq2seq or golden or forward_backward or transformer" --tb=short 2>&1 | tail -4
for m in transformer_base transformer_big; do
  timeout 300 python scripts/bench_text.py --model $m --batch 256 2>/dev/null | tail -1 > $O/c12_text_$m.json
  python -c "
import json; d=json.load(open('gpurun_out/r05/c12_text_$m.json')); print('$m', round(d['ms_per_step'],2), 'ms', 'mfma', round(d['model_mfma_frac'],3), {k:(round(v['ms_per_step'],2), round(v['frac'],3)) for k,v in d['roofline_families'].items()})"
done
