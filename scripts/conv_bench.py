#!/usr/bin/env python
"""Stand-alone timings of the conv front-end kernels at the benchmark shape (B=128, T1=450, F1=40, C=256, bf16).

    python scripts/conv_bench.py [--iters 20] [--out gpurun_out/conv_bench.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, T1, F1, C = a.batch, 450, 40, 256
    T2, F2 = T1 // 2, F1 // 2
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(B, T1, F1, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dy = (torch.randn(B, T2, F2, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w2 = (torch.randn(3, 3, C, C, generator=g) / 48.0).to(torch.bfloat16).to(dev)
    b2 = torch.zeros(C, device=dev)
    dw2, db2 = torch.zeros(3, 3, C, C, device=dev), torch.zeros(C, device=dev)
    flops = 2.0 * B * T2 * F2 * C * 9 * C
    res = {}
    for name, fn in (("conv2_fwd", lambda: K.conv2_fwd(x, w2, b2, relu=True)),
                     ("conv2_dgrad", lambda: K.conv2_dgrad(dy, w2, T1, F1)),
                     ("conv2_wgrad", lambda: K.conv2_wgrad(x, dy, dw2, db2=db2))):
        us = timed(fn, a.iters)
        res[name] = {"us": us, "tflops": flops / us / 1e6, "mfma_frac": flops / us / 1e6 / 2500.0}
        print(name, res[name])
    # layer 1 (C_in = 1): forward conv + LayerNorm + ReLU and its backward (HBM / VALU kernels: bytes per second)
    T, F = 2 * T1, 2 * F1
    src = torch.randn(B, T, F, generator=g).to(dev)
    w1 = (torch.randn(3, 3, 1, C, generator=g) * 0.3).to(dev)
    b1, g1, be1 = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    a1, mean1, rstd1 = K.conv1_ln_relu_fwd(src, w1, b1, g1, be1, True, 1e-6, torch.bfloat16)
    dout = (torch.randn(B, T1, F1, C, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    dw1, db1, dg1, dbe1 = torch.zeros_like(w1), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    nbytes = B * T1 * F1 * C * 2
    for name, fn in (("conv1_fwd", lambda: K.conv1_ln_relu_fwd(src, w1, b1, g1, be1, True, 1e-6, torch.bfloat16)),
                     ("conv1_bwd", lambda: K.conv1_ln_relu_bwd(src, w1, b1, g1, be1, mean1, rstd1, dout, dw1, db1, dg1, dbe1, True,
                                                               1e-6, accumulate=True))):
        us = timed(fn, a.iters)
        res[name] = {"us": us, "activation_tb_per_s": nbytes / us / 1e6}
        print(name, res[name])
    # the LayerNorm + ReLU behind conv2 (576 000 rows x 256 channels): one long HBM stream each way
    y2 = (torch.randn(B, T2, F2, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    yn, mean2, rstd2 = K.layernorm_fwd(y2, gam, bet, 1e-6, relu=True)
    dg2, db2n = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    for name, fn, nb in (("conv2_ln_relu_fwd", lambda: K.layernorm_fwd(y2, gam, bet, 1e-6, relu=True), 2 * y2.numel() * 2),
                         ("conv2_ln_relu_bwd", lambda: K.layernorm_bwd(dy, y2, gam, mean2, rstd2, dg2, db2n, y=yn), 4 * y2.numel() * 2),
                         # the form the model runs: the gate recomputed from x and the statistics, the saved activation not read
                         ("conv2_ln_relu_bwd_regate", lambda: K.layernorm_bwd(dy, y2, gam, mean2, rstd2, dg2, db2n, regate_beta=bet),
                          3 * y2.numel() * 2)):
        us = timed(fn, a.iters)
        res[name] = {"us": us, "tb_per_s": nb / us / 1e6}
        print(name, res[name])
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
