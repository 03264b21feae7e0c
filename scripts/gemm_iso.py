#!/usr/bin/env python3
"""Stand-alone timings of the d_model-wide GEMM shapes of speech_transformer_s at the benchmark batch (through the C ABI, HIP
events, interleaved rounds, medians) -> one JSON document on stdout.

  --ksweep   the same M x N at K = 256 .. 2048: slope = cost of one 64-deep K step, intercept = launch + prologue + epilogue
  default    the projection / feed-forward / gradient shapes with the epilogues the training step uses
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def timed_rounds(fns, rounds=7, it=20):
    """fns: {name: callable}; interleaved rounds -> {name: median microseconds}"""
    for f in fns.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    res = {n: [] for n in fns}
    for _ in range(rounds):
        for n, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(it):
                f()
            e.record()
            torch.cuda.synchronize()
            res[n].append(s.elapsed_time(e) / it * 1e3)
    return {n: statistics.median(v) for n, v in res.items()}


def operands(M, N, Kd, tb, ta=False, out_dtype=torch.bfloat16):
    A = (torch.randn(Kd, M, device=dev) if ta else torch.randn(M, Kd, device=dev)).bfloat16()
    B = (torch.randn(N, Kd, device=dev) if tb else torch.randn(Kd, N, device=dev)).bfloat16()
    return A, B, torch.empty(M, N, device=dev, dtype=out_dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ksweep", action="store_true")
    ap.add_argument("--rows", type=int, default=28800)
    args = ap.parse_args()
    M = args.rows
    out = {"rows": M, "NST_GEMM_GENERIC_EPI": os.environ.get("NST_GEMM_GENERIC_EPI", "0")}
    if args.ksweep:
        fns, meta = {}, {}
        for N in (768, 256):
            for Kd in (256, 512, 1024, 2048):
                A, B, C = operands(M, N, Kd, 0)
                bias = torch.zeros(N, device=dev)
                name = f"fwd_M{M}_N{N}_K{Kd}"
                fns[name] = (lambda A=A, B=B, C=C, N=N, Kd=Kd, bias=bias: K.gemm(A, B, M, N, Kd, out=C, bias=bias))
                meta[name] = (N, Kd)
        us = timed_rounds(fns)
        out["ksweep_us"] = us
        for N in (768, 256):
            t = {Kd: us[f"fwd_M{M}_N{N}_K{Kd}"] for Kd in (256, 512, 1024, 2048)}
            slope = (t[2048] - t[256]) / ((2048 - 256) / 64)
            out[f"N{N}"] = {"us_per_k_step": slope, "intercept_us": t[256] - 4 * slope,
                            "tflops_at_K2048": 2.0 * M * N * 2048 / t[2048] / 1e6}
        print(json.dumps(out))
        return
    d, F = 256, 2048
    res = torch.randn(M, d, device=dev).bfloat16()
    cases = {}
    A, B, C = operands(M, 3 * d, d, 0)
    b3 = torch.zeros(3 * d, device=dev)
    cases["qkv.fwd  [M,768,256] bias"] = (lambda A=A, B=B, C=C: K.gemm(A, B, M, 3 * d, d, out=C, bias=b3), 2.0 * M * 3 * d * d)
    A1, B1, C1 = operands(M, d, d, 0)
    b1 = torch.zeros(d, device=dev)
    cases["out.fwd  [M,256,256] bias+dropout+residual"] = (
        lambda: K.gemm(A1, B1, M, d, d, out=C1, bias=b1, dropout_p=0.1, seed=5, stream_id=3, residual=res), 2.0 * M * d * d)
    A2, B2, C2 = operands(M, d, 3 * d, 1)
    cases["qkv.dgrad [M,256,768] plain"] = (lambda: K.gemm(A2, B2, M, d, 3 * d, trans_b=True, out=C2), 2.0 * M * d * 3 * d)
    A3, B3, C3 = operands(M, d, d, 1)
    ctx = torch.randn(M, d, device=dev).bfloat16()
    delta = torch.empty(M // 225 if M % 225 == 0 else 1, 4, 225 if M % 225 == 0 else M, device=dev)
    if M % 225 == 0:
        cases["out.dgrad [M,256,256] rowdot"] = (
            lambda: K.gemm(A3, B3, M, d, d, trans_b=True, out=C3, rowdot=(ctx, delta, 225)), 2.0 * M * d * d)
    A4, B4, C4 = operands(M, F, d, 0)
    bF = torch.zeros(F, device=dev)
    cases["ffn1.fwd [M,2048,256] bias+relu+dropout"] = (
        lambda: K.gemm(A4, B4, M, F, d, out=C4, bias=bF, relu=True, dropout_p=0.1, seed=5, stream_id=4), 2.0 * M * F * d)
    A5, B5, C5 = operands(M, d, F, 0)
    cases["ffn2.fwd [M,256,2048] bias+dropout+residual"] = (
        lambda: K.gemm(A5, B5, M, d, F, out=C5, bias=b1, dropout_p=0.1, seed=5, stream_id=5, residual=res), 2.0 * M * d * F)
    # weight gradient: dW [256, 768] = x^T [256, M] . dz [M, 768], split-K slabs + fused bias gradient
    x = torch.randn(M, d, device=dev).bfloat16()
    dz = torch.randn(M, 3 * d, device=dev).bfloat16()
    dw = torch.zeros(d, 3 * d, device=dev)
    db = torch.zeros(3 * d, device=dev)
    from neurst_amd.layers.common_layers import _wgrad_split
    sk = _wgrad_split(M, d, 3 * d, torch.bfloat16)
    cases[f"qkv.wgrad [256,768,M] split {sk}"] = (
        lambda: K.gemm(x, dz, d, 3 * d, M, trans_a=True, out=dw, accumulate=True, split_k=sk, colsum_out=db, colsum_accumulate=True),
        2.0 * M * d * 3 * d)
    for tag, kin, nout in (("out", d, d), ("ffn1", d, F), ("ffn2", F, d)):
        xx = torch.randn(M, kin, device=dev).bfloat16()
        dd = torch.randn(M, nout, device=dev).bfloat16()
        ww = torch.zeros(kin, nout, device=dev)
        bb = torch.zeros(nout, device=dev)
        s2 = _wgrad_split(M, kin, nout, torch.bfloat16)
        cases[f"{tag}.wgrad [{kin},{nout},M] split {s2}"] = (
            lambda xx=xx, dd=dd, ww=ww, bb=bb, kin=kin, nout=nout, s2=s2: K.gemm(
                xx, dd, kin, nout, M, trans_a=True, out=ww, accumulate=True, split_k=s2, colsum_out=bb, colsum_accumulate=True),
            2.0 * M * kin * nout)
    us = timed_rounds({n: f for n, (f, _) in cases.items()})
    out["us"] = us
    out["tflops"] = {n: cases[n][1] / us[n] / 1e6 for n in us}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
