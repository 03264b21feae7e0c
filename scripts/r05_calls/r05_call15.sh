#!/bin/bash
# Round 5, call 15: feed-forward keep bits produced ahead: parity (bit-identical), stand-alone timing, step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_ffn.py -q -m gpu --tb=short -x 2>&1 | grep -v "amdgpu.ids" > $O/c15_pytest.log
tail -12 $O/c15_pytest.log | cut -c1-300
timeout 300 python - <<'PY' 2>&1 | tail -6
import torch, sys, json
sys.path.insert(0, ".")
from neurst_amd import kernels as K
DEV="cuda:0"
M,F=28800,2048
x=(torch.randn(M,256,device=DEV)*0.5).bfloat16(); res=(torch.randn(M,256,device=DEV)*0.5).bfloat16()
w1t=(torch.randn(F,256,device=DEV)/16).bfloat16(); w2t=(torch.randn(256,F,device=DEV)/45).bfloat16()
b1=torch.zeros(F,device=DEV); b2=torch.zeros(256,device=DEV)
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return round(s.elapsed_time(e)/iters*1e3,2)
kw=dict(residual=None, hidden_p=0.1, hidden_seed=5, hidden_site=3, out_p=0.1, out_seed=5, out_site=4, save_gate_bits=True)
bits=K.ffn_keep_bits(M,256,F,0.1,5,[3],0.1,DEV)
print("fwd philox      ", timeit(lambda: K.ffn_fwd(x,w1t,b1,w2t,b2,**kw)), "us")
print("fwd keep bits   ", timeit(lambda: K.ffn_fwd(x,w1t,b1,w2t,b2,keep_bits=bits[0],**kw)), "us")
print("fwd no dropout  ", timeit(lambda: K.ffn_fwd(x,w1t,b1,w2t,b2,residual=None,save_gate_bits=True)), "us")
print("fill 12 layers  ", timeit(lambda: K.ffn_keep_bits(M,256,F,0.1,5,list(range(12)),0.1,DEV)), "us")
print("fill 1 layer    ", timeit(lambda: K.ffn_keep_bits(M,256,F,0.1,5,[3],0.1,DEV)), "us")
PY
scripts/gpu_profile2.sh r05c15 8 > $O/c15_profile.log 2>&1
grep -E "ffn_|TOTAL" gpurun_out/r05c15_kernel_stats.csv | cut -c1-120
tail -1 gpurun_out/r05c15_prof_bench.json | python -c 'import sys,json; print("step ms", round(json.loads(sys.stdin.read())["ms_per_step"],3))'
