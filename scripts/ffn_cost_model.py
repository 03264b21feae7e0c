#!/usr/bin/env python3
"""Fixed and per-chunk cost of the eight-wave feed-forward pair (nst_ffn_fwd / nst_ffn_bwd, 28 800 rows, one 128-row workgroup per
CU): the launch is timed at filter sizes 256 .. 2048 (4 .. 32 chunks of 64 hidden units) with and without the dropouts, cold
caches (a 768 MB fill in front of every launch, subtracted), and fitted as  t = a + b * chunks.  b against the MFMA time of a
chunk (128 x 64 x 256 x 2 products = 8.39 MFLOP per workgroup) is the utilisation INSIDE the loop; a is what a kernel with one
workgroup per CU cannot hide (operand fragments, the first chunk's product, the output tile).
-> JSON on stdout and gpurun_out/<tag>_ffn_cost_model.json

    python scripts/ffn_cost_model.py [tag]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K   # noqa: E402

DEV = "cuda:0"
_FLUSH = None


def _time(fn, inner=8, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1000.0 / inner)
    out.sort()
    return out[len(out) // 2]


def cold(fn):
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device=DEV)
    flush = lambda: _FLUSH.fill_(1)     # noqa: E731
    return _time(lambda: (flush(), fn())) - _time(flush)


def fit(xs, ys):
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    return my - b * mx, b


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    M, d = 28800, 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, d, generator=g).to(DEV).bfloat16()
    dy = torch.randn(M, d, generator=g).to(DEV).bfloat16()
    res = {"rows": M, "cases": {}}
    for name, p in (("dropout 0.1 / 0.1", 0.1), ("no dropout", 0.0)):
        pts_f, pts_b = [], []
        for F in (256, 512, 1024, 2048):
            w1 = (torch.randn(d, F, generator=g) * d ** -0.5).to(DEV).bfloat16()
            w2 = (torch.randn(F, d, generator=g) * F ** -0.5).to(DEV).bfloat16()
            w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
            b1, b2 = torch.zeros(F, device=DEV), torch.zeros(d, device=DEV)
            y, h, bits = K.ffn_fwd(x, w1t, b1, w2t, b2, hidden_p=p, hidden_seed=1, hidden_site=1, out_p=p, out_seed=1, out_site=2,
                                   save_gate_bits=True)
            tf = cold(lambda: K.ffn_fwd(x, w1t, b1, w2t, b2, hidden_p=p, hidden_seed=1, hidden_site=1, out_p=p, out_seed=1,
                                        out_site=2, save_gate_bits=True))
            tb = cold(lambda: K.ffn_bwd(dy, h, w2, w1, hidden_p=p, gate_bits=bits))
            pts_f.append((F // 64, tf))
            pts_b.append((F // 64, tb))
        out = {}
        for direction, pts in (("forward", pts_f), ("backward", pts_b)):
            a, b = fit([c for c, _ in pts], [t for _, t in pts])
            mfma_us = 8.39e6 / (256 * 1024 * 1.95e9 / 256 / 1e6) / 1.0    # one chunk of one workgroup on one CU at 1.95 GHz (1024 FLOP / clk / CU x 4 SIMDs)
            out[direction] = {"us_by_chunks": {str(c): round(t, 2) for c, t in pts}, "fixed_us": round(a, 2), "per_chunk_us": round(b, 3),
                              "mfma_us_per_chunk_at_1.95GHz": round(8.39e6 / (4096 * 1.95e9) * 1e6, 3),
                              "in_loop_mfma_utilisation": round((8.39e6 / (4096 * 1.95e9) * 1e6) / b, 3),
                              "fixed_share_at_32_chunks": round(a / (a + 32 * b), 3)}
        res["cases"][name] = out
    print(json.dumps(res, indent=1))
    root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", f"{tag}_ffn_cost_model.json"), "w") as fp:
        json.dump(res, fp, indent=1)


if __name__ == "__main__":
    main()
