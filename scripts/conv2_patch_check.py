"""conv2 forward (LDS-resident patch kernel) vs torch conv2d on a few shapes + timing at B=160.  NST_CONV2_PATCH=0 times the implicit-GEMM path."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from neurst_amd import kernels as K
torch.manual_seed(0)
dev = "cuda:0"
def ref(x, w2, b2, relu):
    # x [B,T1,F1,C] ; w2 [3,3,C,C] (kh,kw,ci,co)
    xi = x.float().permute(0, 3, 1, 2)
    w = w2.float().view(3, 3, x.shape[-1], -1).permute(3, 2, 0, 1)
    y = torch.nn.functional.conv2d(xi, w, b2, stride=2, padding=1)
    if relu: y = y.relu()
    return y.permute(0, 2, 3, 1).contiguous()
for (B, T1, F1) in [] if os.environ.get("TIMING_ONLY") else [(2, 8, 6), (3, 50, 40), (1, 450, 40), (5, 34, 7), (2, 128, 41), (4, 2, 2), (16, 450, 40)]:
    C = 256
    x = (torch.randn(B, T1, F1, C, device=dev)).to(torch.bfloat16)
    w2 = (torch.randn(9 * C, C, device=dev) * 0.02).to(torch.bfloat16)
    b2 = torch.randn(C, device=dev)
    for relu in (False, True):
        y = K.conv2_fwd(x, w2, b2, relu=relu)
        r = ref(x, w2, b2, relu)
        err = (y.float() - r).abs().max().item()
        print(B, T1, F1, relu, "max err", err, "ref max", r.abs().max().item(), flush=True)
        assert err < 0.03 * max(1.0, r.abs().max().item()), "MISMATCH"
B, T1, F1, C = 160, 450, 40, 256
x = torch.randn(B, T1, F1, C, device=dev).to(torch.bfloat16)
w2 = (torch.randn(9 * C, C, device=dev) * 0.02).to(torch.bfloat16)
b2 = torch.randn(C, device=dev)
for _ in range(3): K.conv2_fwd(x, w2, b2, relu=True)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): K.conv2_fwd(x, w2, b2, relu=True)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
fl = 2.0 * B * 225 * 20 * 256 * 2304
print("conv2 fwd patch=%s: %.1f us  %.1f TF/s" % (os.environ.get("NST_CONV2_PATCH", "1"), us, fl / us / 1e6))
dy = torch.randn(B, 225, 20, C, device=dev).to(torch.bfloat16)
for _ in range(3): K.conv2_dgrad(dy, w2, T1, F1)
torch.cuda.synchronize()
e0.record()
for _ in range(20): K.conv2_dgrad(dy, w2, T1, F1)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("conv2 dgrad patch=%s: %.1f us  %.1f TF/s" % (os.environ.get("NST_CONV2_DGRAD_PATCH", "1"), us, fl / us / 1e6))
