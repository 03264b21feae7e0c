"""Parity of the whole-row products (ABI 10: nst_gemm_add_layernorm_fwd, nst_gemm_layernorm_bwd, nst_gemm_rowdot256) against
(a) the float64 restatement of their contract (oracle/kernel_emulation.py: the product rounded to bf16, then the float64
LayerNorm forward / backward of neurst/layers/common_layers.py:73-85) and (b) the unfused pair of HIP launches they replace
(nst_gemm + nst_add_layernorm_fwd / nst_layernorm_bwd_mixed), on the same seeded inputs.  Tolerance: 1e-2 relative to the
largest reference entry for bf16 outputs (north star), 1e-3 for the f32 outputs.

All tests need a real MI355X: run with  pytest -m gpu.
"""
import math

import pytest
import torch

from oracle import kernel_emulation as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def K():
    from neurst_amd import kernels
    return kernels


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def rel(got, ref):
    got, ref = got.detach().float().cpu().double(), ref.detach().double().cpu()
    assert got.shape == ref.shape, f"shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6), \
        float((got - ref).norm()) / max(float(ref.norm()), 1e-12)


def check(name, got, ref, tol, tol_l2=None):
    e, l2 = rel(got, ref)
    assert math.isfinite(e) and e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"
    assert l2 <= (tol_l2 if tol_l2 is not None else tol), f"{name}: rel L2 err {l2:.3e}"


def _weights(k, trans_b, seed):
    w = rnd(k, 256, dtype=torch.bfloat16, seed=seed, scale=k ** -0.5)      # the dense kernel [in, out]
    return w.t().contiguous() if trans_b else w                             # trans_b: the [256, k] operand of an input gradient


# rows: below one tile, ragged tails of both tile heights, the decoder (9 600) and encoder (28 800) row counts of the benchmark
SHAPES = [(5, 64), (37, 256), (64 * 3 + 5, 256), (1000, 768), (4000, 768), (9600, 256), (9600 + 13, 256), (9600, 2048), (14400, 128),
          (28800, 256), (28800 + 17, 256)]


@pytest.mark.parametrize("trans_b", [False])      # (the orientation a forward product has: W [k, 256] as stored)
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("rows,k", SHAPES)
def test_gemm_add_layernorm_fwd(K, rows, k, p, trans_b):
    if rows > 20000 and (p > 0 or trans_b) and k > 256:
        pytest.skip("large case once")
    A = rnd(rows, k, dtype=torch.bfloat16, seed=1)
    W = _weights(k, trans_b, 2)
    bias = rnd(256, seed=3) * 0.1
    x = rnd(rows, 256, seed=4, scale=2.0)
    gamma, beta = rnd(256, seed=5) * 0.2 + 1.0, rnd(256, seed=6) * 0.1
    seed, site = 1234567, 9
    y_r, xs_r, mean_r, rstd_r = E.gemm_add_layernorm_fwd(A, W, x, gamma, beta, 1e-6, bias=bias, trans_b=trans_b, dropout_p=p,
                                                         seed=seed, stream_id=site)
    d = lambda t: t.to(DEV)
    y, xs, mean, rstd = K.gemm_add_layernorm_fwd(d(A), d(W), d(x), d(gamma), d(beta), 1e-6, bias=d(bias), trans_b=trans_b,
                                                 dropout_p=p, seed=seed, stream_id=site)
    tag = f"gemm_add_ln[{rows}x{k},p={p},tb={int(trans_b)}]"
    assert y.dtype == torch.bfloat16 and xs.dtype == torch.float32
    # (the contribution is rounded to bf16 before it joins the stream: an entry on a rounding boundary may land one bf16 step of
    # delta away from the float64 restatement -- 4e-3 of the largest sum at most, while the L2 error stays at the f32 level)
    check(tag + ".xs", xs, xs_r, 4e-3, 3e-4)
    check(tag + ".y", y, y_r, 1e-2)
    check(tag + ".mean", mean, mean_r, 1e-3)
    check(tag + ".rstd", rstd, rstd_r, 1e-3)
    # the unfused pair on the device: the same dropout mask element for element, the same sums up to the accumulation order
    delta = K.gemm(d(A), d(W), rows, 256, k, trans_b=trans_b, bias=d(bias), dropout_p=p, seed=seed, stream_id=site)
    y2, xs2, mean2, rstd2 = K.add_layernorm_fwd(d(x), delta, d(gamma), d(beta), 1e-6)
    if p > 0:
        dropped = (xs2 == d(x))
        assert torch.equal(dropped, xs == d(x)) or float((dropped != (xs == d(x))).float().mean()) < 1e-4, tag + ": dropout masks differ"
    check(tag + ".xs vs pair", xs, xs2.double(), 4e-3, 1e-4)
    check(tag + ".y vs pair", y, y2.double(), 1e-2, 2e-3)
    # without the stored sum
    y3, none, _, _ = K.gemm_add_layernorm_fwd(d(A), d(W), d(x), d(gamma), d(beta), 1e-6, bias=d(bias), trans_b=trans_b,
                                              dropout_p=p, seed=seed, stream_id=site, want_sum=False)
    assert none is None and torch.equal(y3, y)


@pytest.mark.parametrize("trans_b", [True])       # (an input-gradient product reads the kernel [256, k] as stored)
@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("rows,k", SHAPES)
def test_gemm_layernorm_bwd(K, rows, k, drop, trans_b):
    if rows > 20000 and (drop or not trans_b) and k > 256:
        pytest.skip("large case once")
    A = rnd(rows, k, dtype=torch.bfloat16, seed=11)
    W = _weights(k, trans_b, 12)
    x = rnd(rows, 256, seed=13, scale=2.0) + 0.3
    gamma = rnd(256, seed=14) * 0.2 + 1.0
    dres = rnd(rows, 256, dtype=torch.bfloat16, seed=15)
    mean_r, rstd_r = E._ln_stats(x.double(), 1e-6)
    mean_r, rstd_r = mean_r.float(), rstd_r.float()
    emit = (0.25, 424242, 7) if drop else None
    dg_r, db_r = torch.zeros(256), torch.zeros(256)
    out_r = E.gemm_layernorm_bwd(A, W, x, gamma, mean_r, rstd_r, dg_r, db_r, dres=dres, emit_dropout=emit, trans_b=trans_b)
    d = lambda t: t.to(DEV)
    dg, db = torch.full((256,), 7.0, device=DEV), torch.full((256,), 7.0, device=DEV)
    out = K.gemm_layernorm_bwd(d(A), d(W), d(x), d(gamma), d(mean_r), d(rstd_r), dg, db, dres=d(dres), emit_dropout=emit,
                               trans_b=trans_b)
    tag = f"gemm_ln_bwd[{rows}x{k},drop={int(drop)},tb={int(trans_b)}]"
    dx, dx_r = (out[0], out_r[0]) if drop else (out, out_r)
    check(tag + ".dx", dx, dx_r, 2e-2, 8e-3)
    check(tag + ".dgamma", dg, dg_r, 1e-2)
    check(tag + ".dbeta", db, db_r, 1e-2)
    # the unfused pair on the device
    g = K.gemm(d(A), d(W), rows, 256, k, trans_b=trans_b)
    dg2, db2 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    out2 = K.layernorm_bwd(g, d(x), d(gamma), d(mean_r), d(rstd_r), dg2, db2, dres=d(dres), emit_dropout=emit)
    dx2 = out2[0] if drop else out2
    check(tag + ".dx vs pair", dx, dx2.double(), 1e-2, 2e-3)
    check(tag + ".dgamma vs pair", dg, dg2.double(), 1e-3)
    check(tag + ".dbeta vs pair", db, db2.double(), 1e-3)
    if drop:
        dz, dz2 = out[1], out2[1]
        mask0 = K.scale_dropout_bwd(torch.ones_like(dx), 1.0, *emit) == 0
        assert float(((dz == 0) != mask0).float().mean()) < 1e-3, tag + ": mask of dz"
        check(tag + ".dz vs pair", dz, dz2.double(), 1e-2, 2e-3)
    # accumulation into the parameter gradients and the deferred finalize through the batch
    batch = K.SplitkBatch(DEV)
    dg3, db3 = torch.full((256,), 3.0, device=DEV), torch.full((256,), -2.0, device=DEV)
    out3 = K.gemm_layernorm_bwd(d(A), d(W), d(x), d(gamma), d(mean_r), d(rstd_r), dg3, db3, accumulate=True, dres=d(dres),
                                emit_dropout=emit, trans_b=trans_b, batch=batch)
    assert batch.ln_n == 1 and torch.equal(out3[0] if drop else out3, dx)
    batch.flush()
    torch.cuda.synchronize()
    assert torch.allclose(dg3, dg + 3.0, rtol=1e-5, atol=1e-4) and torch.allclose(db3, db - 2.0, rtol=1e-5, atol=1e-4)
    # no residual gradient
    dg4, db4 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    dx4 = K.gemm_layernorm_bwd(d(A), d(W), d(x), d(gamma), d(mean_r), d(rstd_r), dg4, db4, trans_b=trans_b)
    check(tag + ".dx without dres", dx4.double() + d(dres).double(), dx_r.double(), 3e-2, 1e-2)


@pytest.mark.parametrize("B,T,k", [(3, 75, 256), (128, 75, 256), (2, 225, 256), (128, 225, 256), (5, 16, 64), (7, 33, 512)])
def test_gemm_rowdot256(K, B, T, k):
    rows = B * T
    A = rnd(rows, k, dtype=torch.bfloat16, seed=21)
    W = _weights(k, True, 22)
    src = rnd(rows, 256, dtype=torch.bfloat16, seed=23)
    dst_r = torch.zeros(B, 4, T)
    c_r = E.gemm_rowdot256(A, W, rowdot=(src, dst_r, T))
    d = lambda t: t.to(DEV)
    dst = torch.full((B, 4, T), 5.0, device=DEV)
    c = K.gemm_rowdot256(d(A), d(W), rowdot=(d(src), dst, T))
    tag = f"rowdot256[{B}x{T},{k}]"
    check(tag + ".C", c, c_r, 1e-2, 4e-3)
    check(tag + ".delta", dst, dst_r, 1e-2)
    # the stream kernel's epilogue of the same contract
    dst2 = torch.zeros(B, 4, T, device=DEV)
    c2 = K.gemm(d(A), d(W), rows, 256, k, trans_b=True, rowdot=(d(src), dst2, T))
    check(tag + ".C vs stream kernel", c, c2.double(), 1e-2, 2e-3)
    check(tag + ".delta vs stream kernel", dst, dst2.double(), 5e-3)


def test_rowgemm_refuses_what_it_cannot_do(K):
    A = rnd(40, 96, dtype=torch.bfloat16).to(DEV)           # k not a multiple of 64
    assert not K.rowgemm_supported(A, 256, 96) and K.rowgemm_supported(rnd(40, 128, dtype=torch.bfloat16).to(DEV), 256, 128)
    assert not K.rowgemm_supported(rnd(40, 128).to(DEV), 256, 128) and not K.rowgemm_supported(A, 512, 128)
    W = rnd(96, 256, dtype=torch.bfloat16).to(DEV)
    x = rnd(40, 256).to(DEV)
    with pytest.raises(RuntimeError):
        K.gemm_add_layernorm_fwd(A, W, x, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), 1e-6)
    # orientations that no training step issues are not built
    A2, W2 = rnd(40, 128, dtype=torch.bfloat16).to(DEV), rnd(256, 128, dtype=torch.bfloat16).to(DEV)
    with pytest.raises(RuntimeError, match="not built"):
        K.gemm_add_layernorm_fwd(A2, W2, x, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), 1e-6, trans_b=True)
