#!/usr/bin/env python3
"""The decoder's K = 256 projections (9600 rows) timed stand-alone, 50 launches each, on the kernel nst_gemm picks:
NST_GEMM_V2D=1 (default) one tile per workgroup with the whole reduction in flight, NST_GEMM_V2D=0 the persistent stream kernel.
Prints one JSON line {name: us}."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / iters * 1e3, 2)


def main():
    d = 256
    rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)   # noqa: E731
    out = {"NST_GEMM_V2D": os.environ.get("NST_GEMM_V2D", "1")}
    for M in (9600, 4800, 28800):
        x = rnd(M, d)
        for name, N in (("qkv", 768), ("out", 256)):
            w, wt, b = rnd(d, N), rnd(N, d), torch.zeros(N, device=DEV)
            res, gate = rnd(M, N), rnd(M, N)
            out[f"{name}.fwd_bias[{M}x{N}]"] = timeit(lambda: K.gemm(x, w, M, N, d, bias=b))
            out[f"{name}.fwd_bias_drop[{M}x{N}]"] = timeit(lambda: K.gemm(x, w, M, N, d, bias=b, dropout_p=0.1, seed=1, stream_id=2))
            out[f"{name}.dgrad_plain[{M}x{N}]"] = timeit(lambda: K.gemm(x, wt, M, N, d, trans_b=True))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
