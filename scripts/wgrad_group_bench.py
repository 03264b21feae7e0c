#!/usr/bin/env python3
"""Stand-alone timing of nst_gemm_wgrad_group on the weight gradients of speech_transformer_s at the benchmark batch
(B = 128 x T' = 225 encoder rows = 28 800, 128 x 75 decoder rows = 9 600): the encoder stack's 48 products, the decoder
stack's 42, the front dense layer, and everything in one launch -> one JSON document (HIP events, medians).

  NST_GEMM256=11|12|14 python scripts/wgrad_group_bench.py    timing ablations of the kernel (no MFMAs / no DMA / no fragment reads)
"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def product(rows, kin, n, ldz=None, col0=0, zbuf=None):
    x = torch.randn(rows, kin, device=dev).bfloat16()
    if zbuf is None:
        dz = torch.randn(rows, n, device=dev).bfloat16()
    else:
        dz = zbuf[:, col0:col0 + n]
    return (x, dz, torch.zeros(kin, n, device=dev), True, torch.zeros(n, device=dev), True)


def timed(fn, rounds=5, it=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / it * 1e3)
    return round(statistics.median(res), 1)


def main():
    d, F, Re, Rd = 256, 2048, 28800, 9600
    enc = []
    for _ in range(12):
        enc += [product(Re, d, F), product(Re, F, d), product(Re, d, 3 * d), product(Re, d, d)]
    dec = []
    dkv_all = torch.randn(Re, 6 * 2 * d, device=dev).bfloat16()
    mem = torch.randn(Re, d, device=dev).bfloat16()
    for i in range(6):
        dec += [product(Rd, d, F), product(Rd, F, d), product(Rd, d, 3 * d), product(Rd, d, d), product(Rd, d, d), product(Rd, d, d)]
    kv = []
    for i in range(6):
        kv.append((mem, dkv_all[:, i * 2 * d:(i + 1) * 2 * d], torch.zeros(d, 2 * d, device=dev), True, torch.zeros(2 * d, device=dev), True))
    front = [product(Re, 5120, d)]
    logits = [product(Rd, 8008, d)]
    table = torch.empty(1024 * 72, dtype=torch.uint8, device=dev)
    groups = {"encoder(48)": enc, "decoder(36)": dec, "decoder+kv(42)": dec + kv, "front(1)": front, "encoder+kv+front(55)": enc + kv + front,
              "all(97)": enc + kv + front + dec, "all+logits(98)": enc + kv + front + dec + logits, "enc.layer(4)": enc[:4],
              "enc.2layers(8)": enc[:8], "enc.4layers(16)": enc[:16], "enc.6layers(24)": enc[:24]}
    doc = {"NST_GEMM256": os.environ.get("NST_GEMM256", "0"), "us": {}, "tflops": {}, "units": {}}
    for name, items in groups.items():
        us = timed(lambda items=items: K.gemm_wgrad_group(items, table))
        fl = sum(2.0 * x.shape[0] * x.shape[1] * dz.shape[1] for x, dz, *_ in items)
        doc["us"][name] = us
        doc["tflops"][name] = round(fl / us / 1e6, 1)
        doc["units"][name] = sum(((x.shape[1] + 255) // 256) * ((dz.shape[1] + 255) // 256) for x, dz, *_ in items)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
