#!/usr/bin/env python3
"""Runs a few launches of one hot kernel at the benchmark shape, for rocprofv3 --pmc passes
(HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes; MFMA busy).   usage: pmc_target.py conv2_fwd|ffn2_fwd|ffn1_fwd"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

DEV = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "conv2_fwd"
dt = torch.bfloat16
if which == "conv2_fwd":
    B, C = 128, 256
    x = (torch.randn(B, 450, 40, C, device=DEV) * 0.5).to(dt)
    w2 = (torch.randn(3, 3, C, C, device=DEV) * 0.02).to(dt)
    b2 = torch.zeros(C, device=DEV)
    fn = lambda: K.conv2_fwd(x, w2, b2)  # noqa: E731
elif which == "conv1_bwd":
    B, C, T, F = 128, 256, 900, 80
    src = torch.randn(B, T, F, device=DEV)
    w1 = torch.randn(3, 3, 1, C, device=DEV) * 0.3
    b1, g1, be1 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    a1, mean1, rstd1 = K.conv1_ln_relu_fwd(src, w1, b1, g1, be1, True, 1e-6, dt)
    dout = (torch.randn_like(a1.float()) * 0.1).to(dt)
    dw1, db1, dg1, dbe1 = torch.zeros_like(w1), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    fn = lambda: K.conv1_ln_relu_bwd(src, w1, b1, g1, be1, mean1, rstd1, dout, dw1, db1, dg1, dbe1, True, 1e-6, accumulate=True)  # noqa: E731
elif which.endswith("_wgrad"):   # ffn1_wgrad / ffn2_wgrad / qkv_wgrad: dW = x^T . dz over 28 800 rows, split-K slabs
    from neurst_amd.layers.common_layers import _wgrad_split
    M = 28800
    kin, nout = {"ffn1_wgrad": (256, 2048), "ffn2_wgrad": (2048, 256), "qkv_wgrad": (256, 768)}[which]
    x, dz = torch.randn(M, kin, device=DEV).to(dt), torch.randn(M, nout, device=DEV).to(dt)
    dw, db = torch.zeros(kin, nout, device=DEV), torch.zeros(nout, device=DEV)
    sk = _wgrad_split(M, kin, nout, dt)
    fn = lambda: K.gemm(x, dz, kin, nout, M, trans_a=True, out=dw, accumulate=True, split_k=sk, colsum_out=db,  # noqa: E731
                        colsum_accumulate=True)
else:
    M, d, ffn = 28800, 256, 2048
    if which == "ffn2_fwd":
        a, w = (torch.randn(M, ffn, device=DEV) * 0.5).to(dt), (torch.randn(ffn, d, device=DEV) * 0.05).to(dt)
        fn = lambda: K.gemm(a, w, M, d, ffn)  # noqa: E731
    else:
        a, w = (torch.randn(M, d, device=DEV) * 0.5).to(dt), (torch.randn(d, ffn, device=DEV) * 0.05).to(dt)
        fn = lambda: K.gemm(a, w, M, ffn, d)  # noqa: E731
for _ in range(4):
    fn()
torch.cuda.synchronize()
