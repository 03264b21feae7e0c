#!/usr/bin/env python3
"""Weight-gradient GEMMs of speech_transformer_s at the benchmark batch on the 256 x 256 phase-staggered kernel
(nst_gemm256.h): a parity check against torch fp64 on reduced row counts, then stand-alone timings (HIP events, interleaved
rounds, medians; the split-K second stage is inside the timed region) for a sweep of split factors -> one JSON document.

  NST_GEMM256=0 python scripts/gemm256_bench.py   the same calls on the 128 x 128 stream kernel (the round-3 path)
"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def timed_rounds(fns, rounds=5, it=10):
    for f in fns.values():
        for _ in range(2):
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
    return {n: round(statistics.median(v), 2) for n, v in res.items()}


def parity():
    out = {}
    g = torch.Generator().manual_seed(5)
    for rows, kin, n, split in [(1800, 256, 2048, 8), (1800, 2048, 256, 8), (900, 256, 768, 4), (1000, 520, 264, 3),
                                (4096, 256, 256, 16), (130, 256, 256, 1), (64, 512, 256, 1), (2500, 1000, 520, 5)]:
        X = torch.randn(rows, kin, generator=g).bfloat16()
        dZ = torch.randn(rows, n, generator=g).bfloat16()
        ref_w, ref_b = X.double().t() @ dZ.double(), dZ.double().sum(0)
        dw = torch.full((kin, n), 3.0, device=dev)
        db = torch.full((n,), 5.0, device=dev)
        K.gemm(X.to(dev), dZ.to(dev), kin, n, rows, trans_a=True, out=dw, split_k=split, colsum_out=db)
        e1 = float((dw.cpu().double() - ref_w).abs().max() / ref_w.abs().max())
        e2 = float((db.cpu().double() - ref_b).abs().max() / ref_b.abs().max())
        K.gemm(X.to(dev), dZ.to(dev), kin, n, rows, trans_a=True, out=dw, split_k=split, accumulate=True, colsum_out=db,
               colsum_accumulate=True)
        e3 = float((dw.cpu().double() - 2 * ref_w).abs().max() / ref_w.abs().max())
        dw2 = torch.empty(kin, n, device=dev)
        K.gemm(X.to(dev), dZ.to(dev), kin, n, rows, trans_a=True, out=dw2, split_k=split)
        e4 = float((dw2.cpu().double() - ref_w).abs().max() / ref_w.abs().max())
        out[f"{rows}x{kin}x{n}/split{split}"] = {"dw": e1, "db": e2, "dw_acc": e3, "dw_nocs": e4,
                                                 "ok": bool(max(e1, e2, e3, e4) < 1e-5)}
    return out


def main():
    doc = {"NST_GEMM256": os.environ.get("NST_GEMM256", "1"), "parity": parity()}
    print(json.dumps({"parity": doc["parity"]}), flush=True)
    if not all(v["ok"] for v in doc["parity"].values()) and os.environ.get("NST_GEMM256", "1") in ("0", "1"):
        print(json.dumps(doc))
        sys.exit(1)
    shapes = [   # name, rows, k_in, n_out, splits
        ("enc.ffn1", 28800, 256, 2048, (8, 16, 24, 32)),
        ("enc.ffn2", 28800, 2048, 256, (8, 16, 24, 32)),
        ("enc.qkv", 28800, 256, 768, (16, 32, 48, 64)),
        ("enc.out", 28800, 256, 256, (32, 64, 128)),
        ("front", 28800, 5120, 256, (4, 8, 12)),
        ("logits", 9600, 8008, 256, (2, 4, 8)),
        ("dec.ffn1", 9600, 256, 2048, (8, 16, 32)),
        ("dec.ffn2", 9600, 2048, 256, (8, 16, 32)),
        ("dec.qkv", 9600, 256, 768, (16, 32)),
        ("dec.out", 9600, 256, 256, (32, 64)),
    ]
    only = os.environ.get("G256_ONLY")
    if only:
        shapes = [sh for sh in shapes if sh[0] in only.split(",")]
    fns, flops = {}, {}
    for name, rows, kin, n, splits in shapes:
        X = torch.randn(rows, kin, device=dev).bfloat16()
        dZ = torch.randn(rows, n, device=dev).bfloat16()
        dw = torch.zeros(kin, n, device=dev)
        db = torch.zeros(n, device=dev)
        for s in splits:
            key = f"{name}[{kin}x{n}]R{rows}/split{s}"
            fns[key] = (lambda X=X, dZ=dZ, dw=dw, db=db, kin=kin, n=n, rows=rows, s=s:
                        K.gemm(X, dZ, kin, n, rows, trans_a=True, out=dw, split_k=s, colsum_out=db))
            flops[key] = 2.0 * rows * kin * n
    us = timed_rounds(fns)
    doc["us"] = us
    doc["tflops"] = {k: round(flops[k] / us[k] / 1e6, 1) for k in us}
    best = {}
    for k, v in us.items():
        nm = k.split("/")[0]
        if nm not in best or v < best[nm][1]:
            best[nm] = (k.split("/")[1], v)
    doc["best"] = best
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
