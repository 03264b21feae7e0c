#!/usr/bin/env python3
"""Times the whole-row products (nst_gemm_add_layernorm_fwd / nst_gemm_layernorm_bwd / nst_gemm_rowdot256) against the pair of
launches each replaces, at the benchmark's encoder (28 800 rows) and decoder (9 600 rows) shapes.  HIP events on the launch
stream, interleaved A/B, median of `reps` rounds of `inner` launches.  -> JSON on stdout (and gpurun_out/<tag>_rowgemm_bench.json).

    python scripts/rowgemm_bench.py [tag]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K   # noqa: E402

DEV = "cuda:0"


def timeit(fn, inner=20, reps=7):
    """median microseconds per call: `inner` calls captured into one HIP graph (no host time between the launches), replayed."""
    for _ in range(3):
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


_FLUSH = None
_timeit_warm = timeit


def timeit_cold(fn, inner=10, reps=5):
    """As timeit, but a 768 MB fill runs in front of every call (evicts the 8 x 4 MB of L2 and the 256 MB memory-side cache): the
    operands then come from HBM as they do inside the training step.  Returns microseconds per call with the fill's own time
    (measured the same way) subtracted."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(768 << 20, dtype=torch.uint8, device=DEV)

    def flush():
        _FLUSH.fill_(1)
    t_flush = _timeit_warm(flush, inner, reps)
    return _timeit_warm(lambda: (flush(), fn()), inner, reps) - t_flush


def main():
    global timeit
    if "--cold" in sys.argv:
        sys.argv.remove("--cold")
        timeit = timeit_cold
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    g = torch.Generator().manual_seed(0)
    res = {}
    for rows, T in ((28800, 225), (9600, 75)):
        x = torch.randn(rows, 256, generator=g).to(DEV)
        gamma, beta = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
        bias = torch.zeros(256, device=DEV)
        dres = torch.randn(rows, 256, generator=g).to(DEV).bfloat16()
        mean, rstd = torch.zeros(rows, device=DEV), torch.ones(rows, device=DEV)
        dg, db = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
        src = torch.randn(rows, 256, generator=g).to(DEV).bfloat16()
        dst = torch.zeros(rows * 4, device=DEV)
        batch = K.SplitkBatch(DEV)
        for k in (256, 768, 2048):
            A = torch.randn(rows, k, generator=g).to(DEV).bfloat16()
            Wf = (torch.randn(k, 256, generator=g) * k ** -0.5).to(DEV).bfloat16()     # forward operand [k, 256]
            Wb = Wf.t().contiguous()                                                    # input-gradient operand [256, k]
            key = f"rows{rows}_k{k}"
            if k != 768:
                def fused_f():
                    K.gemm_add_layernorm_fwd(A, Wf, x, gamma, beta, 1e-6, bias=bias, dropout_p=0.1, seed=1, stream_id=2)

                def pair_f():
                    d = K.gemm(A, Wf, rows, 256, k, bias=bias, dropout_p=0.1, seed=1, stream_id=2)
                    K.add_layernorm_fwd(x, d, gamma, beta, 1e-6)
                res[key + "_fwd_fused_us"] = timeit(fused_f)
                res[key + "_fwd_pair_us"] = timeit(pair_f)

            def fused_b():
                K.gemm_layernorm_bwd(A, Wb, x, gamma, mean, rstd, dg, db, dres=dres, emit_dropout=(0.1, 1, 3), batch=batch)
                batch.ln_n, batch.ln_since_join = 0, 0

            def pair_b():
                gg = K.gemm(A, Wb, rows, 256, k, trans_b=True)
                K.layernorm_bwd(gg, x, gamma, mean, rstd, dg, db, dres=dres, emit_dropout=(0.1, 1, 3), batch=batch)
                batch.ln_n, batch.ln_since_join = 0, 0
            res[key + "_bwd_fused_us"] = timeit(fused_b)
            res[key + "_bwd_pair_us"] = timeit(pair_b)
            if k == 256:
                res[key + "_rowdot_rows_us"] = timeit(lambda: K.gemm_rowdot256(A, Wb, rowdot=(src, dst, T)))
                res[key + "_rowdot_stream_us"] = timeit(lambda: K.gemm(A, Wb, rows, 256, k, trans_b=True, rowdot=(src, dst, T)))
    out = {k: round(v, 2) for k, v in res.items()}
    out["NST_ROWGEMM_CFG"] = os.environ.get("NST_ROWGEMM_CFG", "")
    out["cold"] = timeit is timeit_cold
    print(json.dumps(out, indent=1))
    root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", f"{tag}_rowgemm_bench.json"), "w") as fp:
        json.dump(out, fp, indent=1)


if __name__ == "__main__":
    main()
