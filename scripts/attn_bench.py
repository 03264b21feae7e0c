#!/usr/bin/env python3
"""Stand-alone timings of the attention kernels at the benchmark shapes (bf16, B = 128, H = 4, dh = 64): encoder self
attention (T = 225), decoder self attention (T = 75, causal), cross attention (Tq = 75, Tk = 225).  HIP events, medians."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K  # noqa: E402

dev = "cuda:0"


def timed(fn, rounds=7, it=20):
    """`it` calls captured in one HIP graph (a 25 us kernel is otherwise timed at the ~30 us the Python call takes)."""
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


def main():
    B, H, dh, d = 128, 4, 64, 256
    res = {"NST_ATTN_FUSED_BWD": os.environ.get("NST_ATTN_FUSED_BWD", "1"), "NST_ATTN_FUSED_FWD": os.environ.get("NST_ATTN_FUSED_FWD", "1")}
    P = float(os.environ.get("ATTN_BENCH_P", "0.1"))
    for name, Tq, Tk, causal, p in (("enc_self", 225, 225, False, P), ("dec_self", 75, 75, True, P), ("dec_cross", 75, 225, False, P)):
        g = torch.Generator().manual_seed(1)
        if Tq == Tk:
            qkv = (torch.randn(B, Tq, 3 * d, generator=g) * 0.5).bfloat16().to(dev)
            q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            dqkv = torch.empty_like(qkv)
            dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
        else:
            q = (torch.randn(B, Tq, d, generator=g) * 0.5).bfloat16().to(dev)
            kv = (torch.randn(B, Tk, 2 * d, generator=g) * 0.5).bfloat16().to(dev)
            k, v = kv[..., :d], kv[..., d:]
            dq = torch.empty_like(q)
            dkv = torch.empty_like(kv)
            dk, dv = dkv[..., :d], dkv[..., d:]
        bias = torch.zeros(B, Tk, device=dev)
        bias[:, Tk - 10:] = K.FLOAT_MIN
        out, lse, mask = K.attention_fwd(q, k, v, H, dh, key_bias=bias, causal=causal, dropout_p=p, seed=3, stream_id=1)
        dout = (torch.randn(B, Tq, d, generator=g) * 0.5).bfloat16().to(dev)
        delta = (dout.float() * out.float()).view(B, Tq, H, dh).sum(-1).permute(0, 2, 1).contiguous()
        res[name + ".fwd_us"] = timed(lambda: K.attention_fwd(q, k, v, H, dh, key_bias=bias, causal=causal, dropout_p=p, seed=3, stream_id=1))
        res[name + ".bwd_us"] = timed(lambda: K.attention_bwd(q, k, v, out, dout, lse, dq, dk, dv, H, dh, key_bias=bias, causal=causal,
                                                            dropout_p=p, seed=3, stream_id=1, drop_mask=mask, delta=delta))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
