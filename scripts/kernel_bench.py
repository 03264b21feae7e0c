#!/usr/bin/env python3
"""Per-kernel timing at the headline workload's shapes (speech_transformer_s, B=128, T=900): every entry point of
libneurst_hip.so timed with HIP events on the launch stream, reported with its algorithmic FLOPs / bytes so each kernel
can be placed against its roofline (MFMA ~2500 TFLOP/s bf16, HBM ~8 TB/s).  Writes gpurun_out/kernel_bench.json.

    python scripts/kernel_bench.py [--dtype bf16|fp32] [--iters 20] [--only gemm,attn,...]
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    esz = 2 if a.dtype == "bf16" else 4
    only = set(x for x in a.only.split(",") if x)
    B, T2, L, d, H, ffn, V, C = 128, 225, 75, 256, 4, 2048, 8008, 256
    Me, Md = B * T2, B * L
    res = {}

    def rnd(*shape, dtype=dt):
        return (torch.randn(*shape, device=DEV) * 0.5).to(dtype)

    def rec(name, us, flops=None, bytes_=None):
        r = {"us": round(us, 2)}
        if flops:
            r["tflops"] = round(flops / us / 1e6, 1)
        if bytes_:
            r["gbps"] = round(bytes_ / us / 1e3, 1)
        res[name] = r
        print(f"{name:46s} {us:10.1f} us" + (f"  {r.get('tflops', 0):8.1f} TF/s" if flops else "") +
              (f"  {r.get('gbps', 0):8.1f} GB/s" if bytes_ else ""), flush=True)

    def want(g):
        return not only or g in only

    # ------------------------------------------------------------------ dense GEMMs
    if want("gemm"):
        shapes = [("qkv", Me, 3 * d, d), ("attn_out", Me, d, d), ("ffn1", Me, ffn, d), ("ffn2", Me, d, ffn),
                  ("front_dense", Me, d, 20 * C), ("dec_ffn1", Md, ffn, d), ("dec_ffn2", Md, d, ffn)]
        for name, M, N, Kd in shapes:
            x, w = rnd(M, Kd), rnd(Kd, N)
            dy = rnd(M, N)
            bias = torch.zeros(N, device=DEV)
            res_t = rnd(M, N)
            fl = 2.0 * M * N * Kd
            rec(f"gemm.{name}.fwd[{M}x{N}x{Kd}]", timeit(lambda: K.gemm(x, w, M, N, Kd, bias=bias), a.iters), fl)
            if name == "ffn1":
                rec(f"gemm.{name}.fwd+relu+dropout", timeit(lambda: K.gemm(x, w, M, N, Kd, bias=bias, relu=True, dropout_p=0.1, seed=1, stream_id=2), a.iters), fl)
            if name in ("ffn2", "attn_out"):
                rec(f"gemm.{name}.fwd+dropout+residual", timeit(lambda: K.gemm(x, w, M, N, Kd, bias=bias, dropout_p=0.1, seed=1, stream_id=2, residual=res_t), a.iters), fl)
            rec(f"gemm.{name}.dgrad", timeit(lambda: K.gemm(dy, w, M, Kd, N, trans_b=True), a.iters), fl)
            if name == "ffn2":
                gate = rnd(M, Kd)
                rec(f"gemm.{name}.dgrad+gate", timeit(lambda: K.gemm(dy, w, M, Kd, N, trans_b=True, gate_src=gate, gate_scale=1.1), a.iters), fl)
            dw = torch.zeros(Kd, N, device=DEV)
            from neurst_amd.layers.common_layers import _wgrad_split
            sp = _wgrad_split(M, Kd, N, dt)
            rec(f"gemm.{name}.wgrad(split{sp})", timeit(lambda: K.gemm(x, dy, Kd, N, M, trans_a=True, out=dw, accumulate=True, split_k=sp), a.iters), fl)
        x, E = rnd(Md, d), rnd(V, d)
        dl = rnd(Md, V)
        bias = torch.zeros(V, device=DEV)
        fl = 2.0 * Md * V * d
        rec(f"gemm.logits.fwd[{Md}x{V}x{d}]", timeit(lambda: K.gemm(x, E, Md, V, d, trans_b=True, bias=bias), a.iters), fl)
        rec("gemm.logits.dgrad", timeit(lambda: K.gemm(dl, E, Md, d, V), a.iters), fl)
        dE = torch.zeros(V, d, device=DEV)
        rec("gemm.logits.wgrad", timeit(lambda: K.gemm(dl, x, V, d, Md, trans_a=True, out=dE, accumulate=True, split_k=4), a.iters), fl)

    # ------------------------------------------------------------------ vendor-library reference (measurement only)
    if want("blas"):
        # torch.mm = hipBLASLt / rocBLAS on the same shapes: not part of the product path, only a yardstick for the
        # hand-written kernel (plain GEMM, no epilogue fusion).
        shapes = [("qkv", Me, 3 * d, d), ("attn_out", Me, d, d), ("ffn1", Me, ffn, d), ("ffn2", Me, d, ffn),
                  ("front_dense", Me, d, 20 * C), ("conv2_as_gemm", B * 225 * 20, C, 9 * C)]
        for name, M, N, Kd in shapes:
            x, w, dy = rnd(M, Kd), rnd(Kd, N), rnd(M, N)
            fl = 2.0 * M * N * Kd
            out = torch.empty(M, N, dtype=dt, device=DEV)
            rec(f"blas.{name}.fwd[{M}x{N}x{Kd}]", timeit(lambda: torch.mm(x, w, out=out), a.iters), fl)
            dxo = torch.empty(M, Kd, dtype=dt, device=DEV)
            rec(f"blas.{name}.dgrad", timeit(lambda: torch.mm(dy, w.t(), out=dxo), a.iters), fl)
            dwo = torch.empty(Kd, N, dtype=dt, device=DEV)
            rec(f"blas.{name}.wgrad", timeit(lambda: torch.mm(x.t(), dy, out=dwo), a.iters), fl)

    # ------------------------------------------------------------------ conv front end
    if want("conv"):
        src = torch.randn(B, 900, 80, device=DEV)
        w1, b1 = torch.randn(3, 3, 1, C, device=DEV) * 0.3, torch.zeros(C, device=DEV)
        g1, be1 = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        a1, mean1, rstd1 = K.conv1_ln_relu_fwd(src, w1, b1, g1, be1, True, 1e-6, dt)
        nb = a1.numel() * esz
        rec("conv1.fwd(+LN+ReLU)", timeit(lambda: K.conv1_ln_relu_fwd(src, w1, b1, g1, be1, True, 1e-6, dt), a.iters), 2.0 * a1.numel() * 9, nb)
        da1 = rnd(*a1.shape)
        dw1, db1, dg1, dbe1 = (torch.zeros_like(t) for t in (w1, b1, g1, be1))
        rec("conv1.bwd", timeit(lambda: K.conv1_ln_relu_bwd(src, w1, b1, g1, be1, mean1, rstd1, da1, dw1, db1, dg1, dbe1, True, 1e-6, accumulate=True), a.iters), None, nb)
        w2, b2 = (torch.randn(3, 3, C, C, device=DEV) * 0.02).to(dt), torch.zeros(C, device=DEV)
        fl = 2.0 * B * 225 * 20 * 9 * C * C
        y2 = K.conv2_fwd(a1, w2, b2)
        rec("conv2.fwd", timeit(lambda: K.conv2_fwd(a1, w2, b2), a.iters), fl)
        dy2 = rnd(*y2.shape)
        rec("conv2.dgrad", timeit(lambda: K.conv2_dgrad(dy2, w2, 450, 40), a.iters), fl)
        dw2 = torch.zeros(3, 3, C, C, device=DEV)
        rec("conv2.wgrad", timeit(lambda: K.conv2_wgrad(a1, dy2, dw2, accumulate=True), a.iters), fl)
        g2, be2 = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        a2, m2, r2 = K.layernorm_fwd(y2, g2, be2, 1e-6, relu=True)
        rec("conv2.ln_relu.fwd", timeit(lambda: K.layernorm_fwd(y2, g2, be2, 1e-6, relu=True), a.iters), None, 2 * y2.numel() * esz)
        dg2, dbe2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        rec("conv2.ln_relu.bwd", timeit(lambda: K.layernorm_bwd(dy2, y2, g2, m2, r2, dg2, dbe2, accumulate=True, y=a2), a.iters), None, 4 * y2.numel() * esz)
        del a1, da1, y2, dy2, a2

    # ------------------------------------------------------------------ LayerNorm / elementwise
    if want("ln"):
        x = rnd(Me, d)
        g, be = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
        y, mean, rstd = K.layernorm_fwd(x, g, be, 1e-6)
        rec("ln.fwd[28800x256]", timeit(lambda: K.layernorm_fwd(x, g, be, 1e-6), a.iters), None, 2 * x.numel() * esz)
        dy, dres = rnd(Me, d), rnd(Me, d)
        dg, dbe = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        rec("ln.bwd(+dres)", timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, dbe, accumulate=True, dres=dres), a.iters), None, 4 * x.numel() * esz)
        rec("dropout.bwd[28800x256]", timeit(lambda: K.scale_dropout_bwd(dy, 1.0, 0.1, 1, 2), a.iters), None, 2 * x.numel() * esz)
        out = torch.zeros(d, device=DEV)
        rec("colsum[28800x256]", timeit(lambda: K.colsum(dy, out, accumulate=True), a.iters), None, x.numel() * esz)
        h = rnd(Me, ffn)
        out2 = torch.zeros(ffn, device=DEV)
        rec("colsum[28800x2048]", timeit(lambda: K.colsum(h, out2, accumulate=True), a.iters), None, h.numel() * esz)

    # ------------------------------------------------------------------ attention
    if want("attn"):
        for name, Tq, Tk, causal, packed in [("enc_self", 225, 225, False, True), ("dec_self", 75, 75, True, True),
                                             ("cross", 75, 225, False, False)]:
            if packed:
                qkv = rnd(B, Tq, 3 * d)
                q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
                dqkv = torch.zeros_like(qkv)
                dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
            else:
                q = rnd(B, Tq, d)
                kv = rnd(B, Tk, 2 * d)
                k, v = kv[..., :d], kv[..., d:]
                dq = torch.zeros_like(q)
                dkv = torch.zeros_like(kv)
                dk, dv = dkv[..., :d], dkv[..., d:]
            bias = torch.zeros(B, Tk, device=DEV)
            fl = 4.0 * B * H * Tq * Tk * 64 * (0.5 if causal else 1.0)
            for p in ((0.1,) if a.tag == "_pmc" else (0.0, 0.1)):
                kw = dict(key_bias=None if causal else bias, causal=causal, dropout_p=p, seed=1, stream_id=3)
                out, lse, dmask = K.attention_fwd(q, k, v, H, 64, **kw)
                rec(f"attn.{name}.fwd(p={p})", timeit(lambda: K.attention_fwd(q, k, v, H, 64, **kw), a.iters), fl)
                dout = rnd(B, Tq, d)
                rec(f"attn.{name}.bwd(p={p})", timeit(lambda: K.attention_bwd(q, k, v, out, dout, lse, dq, dk, dv, H, 64, drop_mask=dmask, **kw), a.iters), 2.5 * fl)

    # ------------------------------------------------------------------ criterion / embedding / optimizer
    if want("misc"):
        logits = rnd(Md, V)
        labels = torch.randint(0, V, (Md,), device=DEV)
        wts = torch.ones(Md, device=DEV)
        xent, lse = K.ls_xent_fwd(logits, labels, wts, 0.1)
        rec("xent.fwd[9600x8008]", timeit(lambda: K.ls_xent_fwd(logits, labels, wts, 0.1), a.iters), None, logits.numel() * esz)
        rec("xent.bwd", timeit(lambda: K.ls_xent_bwd(logits, labels, wts, lse, 0.1, 1e-4), a.iters), None, 2 * logits.numel() * esz)
        n = 29217096
        p, m, v, g = (torch.randn(n, device=DEV) * 0.01 for _ in range(4))
        v.abs_()
        sh = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        rec("adam[29.2M]", timeit(lambda: K.adam_update(p, m, v, g, sh, 1e-4, 0.9, 0.98, 1e-9), a.iters), None, n * 30)
        table = rnd(V, d)
        ids = torch.randint(0, V, (B, L), device=DEV)
        pos = torch.zeros(L, d, device=DEV)
        rec("embedding.fwd", timeit(lambda: K.embedding_fwd(table, ids, pos, L, 16.0), a.iters), None, 2 * Md * d * esz)
        dtab = torch.zeros(V, d, device=DEV)
        dout = rnd(B, L, d)
        rec("embedding.bwd", timeit(lambda: K.embedding_bwd(dout, ids, dtab, 16.0), a.iters), None, Md * d * (esz + 8))
        gbuf = torch.zeros(n, device=DEV)
        rec("grad_memset[29.2M f32]", timeit(lambda: gbuf.zero_(), a.iters), None, n * 4)

    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"kernel_bench_{a.dtype}{a.tag}.json"), "w") as fp:
        json.dump(res, fp, indent=1)


if __name__ == "__main__":
    main()
