#!/usr/bin/env python3
"""Times the fused feed-forward pair against the two-GEMM composition at the benchmark shapes (HIP events, one stream).
    python scripts/ffn_bench.py [--rows 28800,9600] [--ffn 2048] [--iters 30] [--out gpurun_out/ffn_bench.json]
TFLOP/s = 4*M*256*F / time (both products); MFMA fraction against the 2.5 PFLOP/s dense bf16 peak."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="28800,9600")
    ap.add_argument("--ffn", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from neurst_amd import kernels as K
    dev, d, F = "cuda:0", 256, args.ffn
    res = {}
    for M in [int(r) for r in args.rows.split(",")]:
        g = torch.Generator().manual_seed(0)
        bf = torch.bfloat16
        x = torch.randn(M, d, generator=g).to(bf).to(dev)
        r = torch.randn(M, d, generator=g).to(bf).to(dev)
        dy = torch.randn(M, d, generator=g).to(bf).to(dev)
        w1 = (torch.randn(d, F, generator=g) * d ** -0.5).to(bf).to(dev)
        w2 = (torch.randn(F, d, generator=g) * F ** -0.5).to(bf).to(dev)
        w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
        b1, b2 = torch.zeros(F, device=dev), torch.zeros(d, device=dev)
        flop = 4.0 * M * d * F
        row = {}
        for p in (0.0, 0.1):
            def fused_fwd():
                return K.ffn_fwd(x, w1t, b1, w2t, b2, residual=r, hidden_p=p, hidden_seed=1, hidden_site=1, out_p=p, out_seed=1, out_site=2)

            def fused_fwd_bits():
                return K.ffn_fwd(x, w1t, b1, w2t, b2, residual=r, hidden_p=p, hidden_seed=1, hidden_site=1, out_p=p, out_seed=1,
                                 out_site=2, save_gate_bits=True)

            def two_fwd():
                h = K.gemm(x, w1, M, F, d, bias=b1, relu=True, dropout_p=p, seed=1, stream_id=1)
                return K.gemm(h, w2, M, d, F, bias=b2, dropout_p=p, seed=1, stream_id=2, residual=r), h
            t_f, t_2 = timeit(fused_fwd, args.iters), timeit(two_fwd, args.iters)
            t_fb = timeit(fused_fwd_bits, args.iters)
            row[f"fwd_p{p}_gate_bits"] = {"fused_us": t_fb}
            row[f"fwd_p{p}"] = {"fused_us": t_f, "two_gemm_us": t_2, "fused_tflops": flop / t_f / 1e6, "fused_mfma_frac": flop / t_f / 1e6 / 2500.0}
        _, h, bits = K.ffn_fwd(x, w1t, b1, w2t, b2, residual=r, hidden_p=0.1, hidden_seed=1, hidden_site=1, save_gate_bits=True)

        def fused_bwd():
            return K.ffn_bwd(dy, h, w2, w1, hidden_p=0.1)

        def two_bwd():
            dh = K.gemm(dy, w2, M, F, d, trans_b=True, gate_src=h, gate_scale=K.dropout_inv_keep(0.1))
            return K.gemm(dh, w1, M, d, F, trans_b=True), dh
        t_f, t_2 = timeit(fused_bwd, args.iters), timeit(two_bwd, args.iters)
        if bits is not None:
            row["bwd_gate_bits"] = {"fused_us": timeit(lambda: K.ffn_bwd(dy, h, w2, w1, hidden_p=0.1, gate_bits=bits), args.iters)}
        row["bwd"] = {"fused_us": t_f, "two_gemm_us": t_2, "fused_tflops": flop / t_f / 1e6, "fused_mfma_frac": flop / t_f / 1e6 / 2500.0}
        res[f"M{M}_F{F}"] = row
        print(M, json.dumps(row))
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
