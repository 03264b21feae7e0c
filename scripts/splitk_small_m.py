#!/usr/bin/env python3
"""Would split-K pay for the decoder's K-deep, small-M GEMMs?  Times (through a HIP graph, 20 calls) the unsplit bf16 GEMM against
the f32 split-K form (slabs + reduce, existing path) for M = 9600: ffn2 forward [M,256,2048], logits dgrad [M,256,8008]."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

dev = "cuda:0"


def timed(fn, rounds=7, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(it):
                fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / it * 1e3)
    return statistics.median(out)


res = {}
M = 9600
for name, N, Kd, tb in (("ffn2_fwd", 256, 2048, False), ("logits_dgrad", 256, 8008, False), ("ffn1_dgrad", 256, 2048, True)):
    A = torch.randn(M, Kd, device=dev).bfloat16()
    B = (torch.randn(N, Kd, device=dev) if tb else torch.randn(Kd, N, device=dev)).bfloat16()
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    of = torch.empty(M, N, device=dev, dtype=torch.float32)
    res[name + ".bf16_unsplit_us"] = timed(lambda: K.gemm(A, B, M, N, Kd, trans_b=tb, out=ob))
    for sp in (2, 3, 4, 6):
        res[name + f".f32_split{sp}_us"] = timed(lambda: K.gemm(A, B, M, N, Kd, trans_b=tb, out=of, split_k=sp))
print(json.dumps(res, indent=1))
