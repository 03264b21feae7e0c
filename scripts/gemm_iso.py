import sys, torch
sys.path.insert(0, '/root/repo')
from neurst_amd import kernels as K
dev = torch.device('cuda:0')
def timed(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (M, N, Kd, tb) in [(28800, 768, 256, 0), (28800, 256, 256, 0), (28800, 256, 768, 1), (28800, 256, 256, 1), (9600, 2048, 256, 0), (9600, 256, 2048, 0), (9600, 768, 256, 0), (9600, 256, 256, 0), (28800, 512, 256, 0)]:
    A = torch.randn(M, Kd, device=dev).bfloat16()
    B = (torch.randn(N, Kd, device=dev) if tb else torch.randn(Kd, N, device=dev)).bfloat16()
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us = timed(lambda: K.gemm(A, B, M, N, Kd, trans_b=bool(tb), out=out, bias=bias))
    print(f"M{M} N{N} K{Kd} tb{tb}: {us:7.1f} us  {2.0*M*N*Kd/us/1e6:7.1f} TF/s")
