#!/usr/bin/env python3
"""Products of the benchmark step the role-split stream kernel (dense_gemm_kernel_v3s, NST_GEMM_V3S=<largest K it takes>) can
take, timed stand-alone (HIP events, 50 launches).  Prints one JSON line {name: us}."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    d, Me, Md = 256, 128 * 225, 128 * 75
    rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)   # noqa: E731
    out = {"NST_GEMM_V3S": os.environ.get("NST_GEMM_V3S", "0")}
    for tag, M in (("enc", Me), ("dec", Md)):
        x = rnd(M, d)
        for name, N in (("qkv", 768), ("out", 256), ("kv_group", 3072), ("ffn1", 2048)):
            w, wt, b = rnd(d, N), rnd(N, d), torch.zeros(N, device=DEV)
            out[f"{tag}.{name}.fwd_bias[{M}x{N}]"] = timeit(lambda: K.gemm(x, w, M, N, d, bias=b))
            if name == "out":
                out[f"{tag}.{name}.fwd_bias_drop"] = timeit(lambda: K.gemm(x, w, M, N, d, bias=b, dropout_p=0.1, seed=1, stream_id=2))
                o, delta = rnd(M, N), torch.empty(M // 75 if tag == "dec" else M // 225, 4, 75 if tag == "dec" else 225, device=DEV)
                out[f"{tag}.{name}.dgrad_rowdot"] = timeit(lambda: K.gemm(x, wt, M, N, d, trans_b=True, rowdot=(o, delta, delta.shape[2])))
            out[f"{tag}.{name}.dgrad_plain[{M}x{N}]"] = timeit(lambda: K.gemm(x, wt, M, N, d, trans_b=True))
        # long reductions: ffn2 forward (K = 2048), qkv / ffn1 input gradients (K = 768 / 2048)
        for name, Kd, N in (("ffn2.fwd_bias_drop", 2048, 256), ("qkv.dgrad", 768, 256), ("ffn1.dgrad", 2048, 256)):
            xx, w, wt, b = rnd(M, Kd), rnd(Kd, N), rnd(N, Kd), torch.zeros(N, device=DEV)
            if "fwd" in name:
                out[f"{tag}.{name}[{M}x{N}x{Kd}]"] = timeit(lambda: K.gemm(xx, w, M, N, Kd, bias=b, dropout_p=0.1, seed=1, stream_id=2))
            else:
                out[f"{tag}.{name}[{M}x{N}x{Kd}]"] = timeit(lambda: K.gemm(xx, wt, M, N, Kd, trans_b=True))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
