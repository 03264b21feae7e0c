#!/usr/bin/env python3
"""Long reductions over few tiles, stand-alone (HIP events, 30 launches each): NST_GEMM_V2L=<max tiles> (0 = the stream kernel).
Prints one JSON line {name: us}."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=30, warmup=5):
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
    rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)   # noqa: E731
    out = {"NST_GEMM_V2L": os.environ.get("NST_GEMM_V2L", "256")}
    Me, Md, d = 28800, 9600, 256
    # decoder ffn2 forward: h [Md, 2048] . W2 [2048, 256] + bias, dropout (fp32 residual stream: no residual in the epilogue)
    h, w2, b = rnd(Md, 2048), rnd(2048, d), torch.zeros(d, device=DEV)
    out["dec.ffn2.fwd_bias_drop[9600x256x2048]"] = timeit(lambda: K.gemm(h, w2, Md, d, 2048, bias=b, dropout_p=0.1, seed=1, stream_id=2))
    # decoder ffn1 input gradient: dh [Md, 2048] . W1^T (W1 [256, 2048] read as [N = 256, K = 2048])
    w1 = rnd(d, 2048)
    out["dec.ffn1.dgrad[9600x256x2048]"] = timeit(lambda: K.gemm(h, w1, Md, d, 2048, trans_b=True))
    # logits input gradient: dlogits [Md, 8064] . E [8064, 256]
    dl, emb = rnd(Md, 8064), rnd(8064, d)
    out["logits.dgrad[9600x256x8064]"] = timeit(lambda: K.gemm(dl, emb, Md, d, 8064))
    # front dense forward [Me, 5120] . W [5120, 256] + bias, posenc
    a2, wf = rnd(Me, 5120), rnd(5120, d)
    pos = torch.randn(225, d, device=DEV)
    out["front.fwd_bias_posenc[28800x256x5120]"] = timeit(lambda: K.gemm(a2, wf, Me, d, 5120, bias=b, posenc=pos, posenc_period=225, emb_scale=16.0))
    # d(memory): dkv_all [Me, 3072] . Wkv^T ([256, 3072] read as [N = 256, K = 3072])
    dkv, wkv = rnd(Me, 3072), rnd(d, 3072)
    out["dmemory.dgrad[28800x256x3072]"] = timeit(lambda: K.gemm(dkv, wkv, Me, d, 3072, trans_b=True))
    # encoder-size ffn2 (should stay on the stream kernel or the pair): 450 tiles
    he = rnd(Me, 2048)
    out["enc.ffn2.fwd_bias_drop[28800x256x2048]"] = timeit(lambda: K.gemm(he, w2, Me, d, 2048, bias=b, dropout_p=0.1, seed=1, stream_id=2))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
