#!/usr/bin/env python3
"""K = 256 products of the benchmark step, timed stand-alone (HIP events, 50 launches each) on the kernel nst_gemm picks:
NST_GEMM_RP=1 (default) the row-panel kernel (csrc/nst_gemm_rowpanel.h), NST_GEMM_RP=0 the 128 x 128 stream kernel.
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
    d, Me, Md = 256, 128 * 225, 128 * 75
    rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)   # noqa: E731
    out = {"NST_GEMM_RP": os.environ.get("NST_GEMM_RP", "1")}
    for tag, M in (("enc", Me), ("dec", Md)):
        x = rnd(M, d)
        for name, N in (("qkv", 768), ("out", 256), ("kv_group", 3072), ("ffn1", 2048)):
            w, wt, b = rnd(d, N), rnd(N, d), torch.zeros(N, device=DEV)
            res, gate = rnd(M, N), rnd(M, N)
            out[f"{tag}.{name}.fwd_bias[{M}x{N}]"] = timeit(lambda: K.gemm(x, w, M, N, d, bias=b))
            if name == "out":
                out[f"{tag}.{name}.fwd_bias_drop"] = timeit(lambda: K.gemm(x, w, M, N, d, bias=b, dropout_p=0.1, seed=1, stream_id=2))
                out[f"{tag}.{name}.fwd_bias_drop_res"] = timeit(lambda: K.gemm(x, w, M, N, d, bias=b, dropout_p=0.1, seed=1, stream_id=2, residual=res))
                o, delta = rnd(M, N), torch.empty(M // 75 if tag == "dec" else M // 225, 4, 75 if tag == "dec" else 225, device=DEV)
                out[f"{tag}.{name}.dgrad_rowdot"] = timeit(lambda: K.gemm(x, wt, M, N, d, trans_b=True, rowdot=(o, delta, delta.shape[2])))
            if name == "ffn1":
                out[f"{tag}.{name}.fwd_bias_relu_drop"] = timeit(lambda: K.gemm(x, w, M, N, d, bias=b, relu=True, dropout_p=0.1, seed=1, stream_id=2))
                out[f"{tag}.ffn2.dgrad_gate[{M}x{N}]"] = timeit(lambda: K.gemm(x, wt, M, N, d, trans_b=True, gate_src=gate, gate_scale=1.1))
            out[f"{tag}.{name}.dgrad_plain[{M}x{N}]"] = timeit(lambda: K.gemm(x, wt, M, N, d, trans_b=True))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
