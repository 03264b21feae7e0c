"""Parity of every HIP kernel (called through the C ABI of libneurst_hip.so) against the CPU oracle /
plain torch fp64 math on the same seeded inputs.  Tolerances are the north-star ones:
1e-3 (fp32 path) and 1e-2 (bf16 path), relative to the magnitude of the reference tensor.

All tests need a real MI355X: run with  pytest -m gpu.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import neurst_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}
TOL_L2 = {torch.float32: float(os.environ.get("NST_TEST_L2_F32", "2e-5")), torch.bfloat16: float(os.environ.get("NST_TEST_L2_BF16", "8e-3"))}
REPORT = {}


@pytest.fixture(scope="module")
def K():
    from neurst_amd import kernels
    return kernels


@pytest.fixture(scope="module", autouse=True)
def _dump_report():
    yield
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    tag = "tr"
    path = os.path.join(out, f"kernel_report_{tag}.json")
    merged = {}
    if os.path.exists(path):
        try:
            merged = json.load(open(path))
        except Exception:
            merged = {}
    merged.update(REPORT)
    with open(path, "w") as fp:
        json.dump(merged, fp, indent=1, sort_keys=True)


def close(name, got, ref, dtype, scale=1.0):
    got = got.detach().float().cpu().double()
    ref = ref.detach().double()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    denom = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max()) / denom
    REPORT[name] = err
    assert math.isfinite(err) and err <= TOL[dtype] * scale, f"{name}: rel err {err:.3e} > {TOL[dtype] * scale:.1e}"
    # the same comparison in the L2 norm (max-abs over max-abs alone is lenient for bf16 outputs: one large entry sets the scale)
    l2 = float((got - ref).norm()) / max(float(ref.norm()), 1e-12)
    REPORT[name + " |l2"] = l2
    bound = TOL_L2[dtype] * max(scale, 0.05)     # (scale < 1 marks the exact-arithmetic cases: 1e-6 in fp32)
    assert math.isfinite(l2) and l2 <= bound, f"{name}: rel L2 err {l2:.3e} > {bound:.1e}"


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(*shape, generator=g) * scale
    return t.to(dtype)


DTYPES = [torch.float32, torch.bfloat16]


# ------------------------------------------------------------------------------------------------ probes
def test_mfma_fragment_maps(K):
    c16, c16f, tr = K.probe_mfma(DEV)
    i = np.arange(16)
    A32 = ((i[:, None] * 37 + np.arange(32)[None, :] * 11) % 17 - 8) / 8.0
    B32 = ((np.arange(32)[:, None] * 13 + i[None, :] * 7) % 19 - 9) / 16.0
    np.testing.assert_allclose(c16.cpu().numpy().reshape(16, 16), A32 @ B32, atol=1e-5)
    A4 = ((i[:, None] * 37 + np.arange(4)[None, :] * 11) % 17 - 8) / 8.0
    B4 = ((np.arange(4)[:, None] * 13 + i[None, :] * 7) % 19 - 9) / 16.0
    np.testing.assert_allclose(c16f.cpu().numpy().reshape(16, 16), A4 @ B4, atol=1e-6)


def test_lds_transpose_read_map(K):
    _, _, tr = K.probe_mfma(DEV)
    got = tr.cpu().numpy().astype(np.int64).reshape(64, 8)
    lane = np.arange(64)[:, None]
    j = np.arange(8)[None, :]
    expect = ((lane >> 4) * 8 + j) * 16 + (lane & 15)
    REPORT["tr_read_mismatches"] = int((got != expect).sum())
    np.testing.assert_array_equal(got, expect)


# ------------------------------------------------------------------------------------------------ fp32 residual stream
@pytest.mark.parametrize("x_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d", [(37, 8), (300, 256), (100, 512), (33, 1024), (1000, 128), (131, 64), (20001, 256), (77, 768),
                                    (1, 256), (5, 16)])
def test_add_layernorm_float32_residual_stream(K, x_dtype, rows, d):
    """nst_add_layernorm_fwd / nst_layernorm_bwd_mixed against float64: the sum x + delta is formed and stored in f32 (exact for
    a bf16 delta on an f32 or bf16 x up to one f32 rounding), normalised in the same pass; the backward reads the f32 sum."""
    x = rnd(rows, d, dtype=x_dtype, seed=1, scale=3.0)
    delta = rnd(rows, d, dtype=torch.bfloat16, seed=6)
    gamma = rnd(d, seed=2) * 0.2 + 1.0
    beta = rnd(d, seed=3) * 0.1
    dy = rnd(rows, d, dtype=torch.bfloat16, seed=4)
    dres = rnd(rows, d, dtype=torch.bfloat16, seed=5)
    xs_ref = (x.double() + delta.double()).float()          # the device's f32 sum (one rounding, same on both sides)
    xr = xs_ref.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = O.layer_norm(xr, gr, br, 1e-6)
    yr.backward(dy.double())
    y, xs, mean, rstd = K.add_layernorm_fwd(x.to(DEV), delta.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6)
    tag = f"add_ln[{x_dtype},{rows}x{d}]"
    assert xs.dtype == torch.float32 and y.dtype == torch.bfloat16
    assert torch.equal(xs.cpu(), xs_ref), tag + ": the f32 sum is not the correctly rounded x + delta"
    close(tag + ".y", y, yr, torch.bfloat16)
    close(tag + ".mean", mean, xs_ref.double().mean(1), torch.float32)
    y2, none, _, _ = K.add_layernorm_fwd(x.to(DEV), delta.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, want_sum=False)
    assert none is None and torch.equal(y2, y)
    # same statistics as the plain f32 forward on the stored sum (another instantiation: the last bit of rstd may differ)
    y3, mean3, rstd3 = K.layernorm_fwd(xs, gamma.to(DEV), beta.to(DEV), 1e-6) if torch.device(DEV).type == "cuda" else (None, mean, rstd)
    if y3 is not None:
        assert torch.allclose(mean3, mean, rtol=1e-6, atol=1e-7) and torch.allclose(rstd3, rstd, rtol=1e-6, atol=0)
    dgamma, dbeta = torch.full((d,), 7.0, device=DEV), torch.full((d,), 7.0, device=DEV)
    dx = K.layernorm_bwd(dy.to(DEV), xs, gamma.to(DEV), mean, rstd, dgamma, dbeta, dres=dres.to(DEV))
    assert dx.dtype == torch.bfloat16
    close(tag + ".dx", dx, xr.grad + dres.double(), torch.bfloat16, scale=2.0)
    close(tag + ".dgamma", dgamma, gr.grad, torch.bfloat16, scale=2.0)
    close(tag + ".dbeta", dbeta, br.grad, torch.bfloat16, scale=2.0)
    dg2, db2 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    dx2, dz2 = K.layernorm_bwd(dy.to(DEV), xs, gamma.to(DEV), mean, rstd, dg2, db2, dres=dres.to(DEV), emit_dropout=(0.3, 11, 5))
    assert torch.equal(dx2, dx)
    dz_ref = K.scale_dropout_bwd(dx2, 1.0, 0.3, 11, 5)
    assert torch.equal(dz2 == 0, dz_ref == 0)
    if torch.device(DEV).type != "cuda":
        return
    # deferred parameter gradients through the batch: bit-identical to the immediate second stage
    batch = K.SplitkBatch(DEV)
    g_k, b_k = torch.full((d,), 3.0, device=DEV), torch.full((d,), 3.0, device=DEV)
    dx_k = K.layernorm_bwd(dy.to(DEV), xs, gamma.to(DEV), mean, rstd, g_k, b_k, dres=dres.to(DEV), batch=batch)
    assert batch.ln_n == (1 if d <= 512 else 0) and torch.equal(dx_k, dx)
    batch.flush()
    torch.cuda.synchronize()
    assert torch.equal(g_k, dgamma) and torch.equal(b_k, dbeta)


def test_add_layernorm_refuses_what_it_cannot_do(K):
    if torch.device(DEV).type != "cuda":
        pytest.skip("argument checks of the library")
    x, delta = rnd(8, 12).to(DEV), rnd(8, 12, dtype=torch.bfloat16).to(DEV)
    with pytest.raises(RuntimeError):
        K.add_layernorm_fwd(x, delta, torch.ones(12, device=DEV), torch.zeros(12, device=DEV), 1e-6)
    assert not K.add_layernorm_supported(12, torch.bfloat16) and not K.add_layernorm_supported(256, torch.float32)
    assert K.add_layernorm_supported(256, torch.bfloat16)


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,d", [(7, 4), (37, 8), (300, 256), (100, 512), (33, 1024), (5, 250), (1000, 128), (131, 64),
                                    (20001, 256), (77, 768)])
@pytest.mark.parametrize("relu", [False, True])
def test_layernorm(K, dtype, rows, d, relu):
    x = rnd(rows, d, dtype=dtype, seed=1)
    gamma = rnd(d, seed=2) * 0.2 + 1.0
    beta = rnd(d, seed=3) * 0.1
    dy = rnd(rows, d, dtype=dtype, seed=4)
    dres = rnd(rows, d, dtype=dtype, seed=5)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = O.layer_norm(xr, gr, br, 1e-6)
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy.double())
    y, mean, rstd = K.layernorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, relu=relu)
    tag = f"ln{'_relu' if relu else ''}[{dtype},{rows}x{d}]"
    close(tag + ".y", y, yr, dtype)
    dgamma = torch.full((d,), 7.0, device=DEV)
    dbeta = torch.full((d,), 7.0, device=DEV)
    if relu:
        dx = K.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dgamma, dbeta, y=y)
        close(tag + ".dx", dx, xr.grad, dtype, scale=3.0)
        if d % 8 == 0:   # the same with the gate recomputed from x and the statistics instead of read from y
            dg_r, db_r = torch.full((d,), 7.0, device=DEV), torch.full((d,), 7.0, device=DEV)
            dx_r = K.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dg_r, db_r, regate_beta=beta.to(DEV))
            close(tag + ".dx_regate", dx_r, xr.grad, dtype, scale=3.0)
            if dtype == torch.float32:   # fp32: y > 0 and the recomputed LN(x) > 0 agree except where LN(x) is within an ulp of 0
                assert float((dx_r != dx).float().mean()) < 1e-3
                close(tag + ".dgamma_regate", dg_r, gr.grad, dtype, scale=2.0)
    else:
        dx = K.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dgamma, dbeta, dres=dres.to(DEV))
        close(tag + ".dx", dx, xr.grad + dres.double(), dtype, scale=2.0)
    if not relu or dtype == torch.float32:  # relu gate on a bf16-rounded y can flip near zero
        close(tag + ".dgamma", dgamma, gr.grad, dtype, scale=2.0)
        close(tag + ".dbeta", dbeta, br.grad, dtype, scale=2.0)
    if not relu:  # second output: dropout backward of dx under the consumer's mask, bit-identical to the stand-alone kernel
        dg2, db2 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        dx2, dz2 = K.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dg2, db2, dres=dres.to(DEV),
                                   emit_dropout=(0.3, 11, 5))
        assert torch.equal(dx2, dx)
        dz_ref = K.scale_dropout_bwd(dx2, 1.0, 0.3, 11, 5)
        if dtype == torch.float32:
            assert torch.equal(dz2, dz_ref)
        else:  # the fused output is rounded once from the f32 gradient, the stand-alone kernel re-rounds the bf16 dx
            assert torch.equal(dz2 == 0, dz_ref == 0)
            close(tag + ".dz_emit", dz2, dz_ref.float().cpu().double(), dtype)
        kept = float((dz2 != 0).float().mean())
        assert rows * d < 2000 or abs(kept - 0.7) < 0.05
    # accumulate
    before = dgamma.clone()
    K.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dgamma, dbeta, accumulate=True,
                    y=y if relu else None)
    close(tag + ".dgamma_acc", dgamma, 2 * before.cpu().double(), torch.float32)
    # deferred parameter gradients (nst_layernorm_bwd_deferred + nst_ln_finalize_multi): the dx kernels of several
    # LayerNorms leave their partial sums in the batch's slots, ONE later launch finishes all of them -- bit-identical
    # to the immediate second stage; d > 512 has no slot and falls back to it
    if torch.device(DEV).type != "cuda":     # (dry run over the emulation: there is no deferred stage to test)
        return
    batch = K.SplitkBatch(DEV)
    want_g, want_b = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    kw = dict(y=y) if relu else dict(dres=dres.to(DEV))
    want_dx = K.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, want_g, want_b, **kw)
    outs = []
    for k in range(3):
        g_k, b_k = torch.full((d,), 3.0, device=DEV), torch.full((d,), 3.0, device=DEV)
        dx_k = K.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, g_k, b_k, accumulate=(k == 2), batch=batch, **kw)
        outs.append((dx_k, g_k, b_k))
    assert batch.ln_n == (3 if d <= 512 else 0)
    if d <= 512:
        assert float(outs[0][1][0]) == 3.0, "the deferred stage ran before the flush"
    batch.flush()
    assert batch.ln_n == 0
    for k, (dx_k, g_k, b_k) in enumerate(outs):
        assert torch.equal(dx_k, want_dx)
        assert torch.equal(g_k, want_g + 3.0 if k == 2 else want_g), f"{tag}: deferred dgamma {k}"
        assert torch.equal(b_k, want_b + 3.0 if k == 2 else want_b), f"{tag}: deferred dbeta {k}"
    if not relu:
        g_e, b_e = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        dx_e, dz_e = K.layernorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, g_e, b_e, dres=dres.to(DEV),
                                     emit_dropout=(0.3, 11, 5), batch=batch)
        batch.flush()
        assert torch.equal(dx_e, dx2) and torch.equal(dz_e, dz2) and torch.equal(g_e, dg2) and torch.equal(b_e, db2)


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(5, 7, 3), (16, 16, 32), (128, 128, 64), (130, 136, 72), (300, 256, 256), (257, 120, 513),
               (64, 520, 40), (1000, 256, 2048)]


@pytest.mark.parametrize("B,T,H,Kd", [(3, 75, 4, 256), (2, 225, 4, 256), (5, 16, 2, 64), (1, 130, 8, 512)])
def test_gemm_rowdot_leaves_the_attention_delta(K, B, T, H, Kd):
    """NstGemmDesc.rowdot_*: the input gradient of the attention output projection (dO = dZ . Wo^T) also leaves
    delta[b, h, t] = sum over the head's 64 columns of dO (as stored in bf16) * O -- what attn_delta_*_kernel computes in a
    separate pass.  nst_attention_bwd with that delta (out == NULL) equals the call that computes it itself."""
    if torch.device(DEV).type != "cuda":
        pytest.skip("needs the device: the fused row dots exist in the stream kernel only")
    M, N = B * T, H * 64
    dz = rnd(M, Kd, dtype=torch.bfloat16, seed=31).to(DEV)
    w = (rnd(N, Kd, seed=32) / math.sqrt(Kd)).to(torch.bfloat16).to(DEV)       # [in = N, out = Kd] read as the [N, K] operand
    o = rnd(M, N, dtype=torch.bfloat16, seed=33).to(DEV)
    assert K.rowdot_supported(dz, N)
    delta = torch.full((B, H, T), 7.0, device=DEV)
    do = K.gemm(dz, w, M, N, Kd, trans_b=True, rowdot=(o, delta, T))
    assert torch.equal(do, K.gemm(dz, w, M, N, Kd, trans_b=True))
    want = (do.float().view(B, T, H, 64) * o.float().view(B, T, H, 64)).sum(-1).permute(0, 2, 1)
    assert float((delta - want).abs().max()) <= 1e-4 * float(want.abs().max())
    # the attention backward fed with it
    q, k, v = (rnd(B, T, N, dtype=torch.bfloat16, seed=40 + i).to(DEV) for i in range(3))
    ctx, lse, _ = K.attention_fwd(q, k, v, H, 64)
    outs = []
    for given in (False, True):
        dq, dk, dv = (torch.empty_like(q) for _ in range(3))
        dl = None
        if given:
            dl = torch.empty_like(lse)
            K.gemm(dz, w, M, N, Kd, trans_b=True, rowdot=(ctx.view(M, N), dl, T))
        K.attention_bwd(q, k, v, ctx, do.view(B, T, N), lse, dq, dk, dv, H, 64, delta=dl)
        outs.append((dq, dk, dv))
    for a, b_ in zip(*outs):
        assert float((a.float() - b_.float()).abs().max()) <= 2e-2 * float(a.float().abs().max()) + 1e-6


@pytest.mark.parametrize("M,N,Kd", [(4096, 4096, 1024), (4000, 3600, 520), (16384, 1024, 4096), (3840, 3584, 512)])
def test_gemm_256_tile_forward_and_input_gradient(K, M, N, Kd):
    """bf16 forward / input-gradient products whose 256 x 256 tiles fill the chip (K >= 512: the text Transformers) run on the
    phase-staggered 256 x 256 main loop with the compile-time epilogues (csrc/nst_gemm.hip: dense_gemm256_kernel): every
    epilogue it is dispatched for against fp64 on the same bf16 inputs -- whole tiles, ragged tiles in both output dimensions
    (4000 = 15 tiles + 160 rows, 3600 = 14 tiles + 16 columns), a reduction that is not a multiple of the K step (520), a long
    reduction (4096), both operand layouts, strided views.  Dropout masks: the library's own (oracle/philox.py over the element
    index row * N + col)."""
    from oracle import philox
    if torch.device(DEV).type != "cuda":
        pytest.skip("needs the device: kernel selection of the library")
    x = rnd(M, Kd, dtype=torch.bfloat16, seed=1)
    w = (rnd(Kd, N, seed=2) / math.sqrt(Kd)).to(torch.bfloat16)
    wt = (rnd(N, Kd, seed=3) / math.sqrt(Kd)).to(torch.bfloat16)
    bias = rnd(N, seed=4)
    res = rnd(M, N, dtype=torch.bfloat16, seed=5)
    gate = rnd(M, N, dtype=torch.bfloat16, seed=6)
    xd, wd, wtd, bd = x.to(DEV), w.to(DEV), wt.to(DEV), bias.to(DEV)
    base = (xd.double() @ wd.double()).cpu() + bias.double()
    tag = f"gemm256[{M}x{N}x{Kd}]"
    close(tag + ".bias", K.gemm(xd, wd, M, N, Kd, bias=bd), base, torch.bfloat16)
    close(tag + ".bias_relu", K.gemm(xd, wd, M, N, Kd, bias=bd, relu=True), torch.relu(base), torch.bfloat16)
    close(tag + ".bias_res", K.gemm(xd, wd, M, N, Kd, bias=bd, residual=res.to(DEV)), base + res.double(), torch.bfloat16)
    p, seed, site = 0.25, 1234, 7
    keep = torch.from_numpy(philox.keep_multiplier(seed, site, M * N, p)).reshape(M, N).double()
    got = K.gemm(xd, wd, M, N, Kd, bias=bd, dropout_p=p, seed=seed, stream_id=site)
    assert bool(((got.cpu() == 0) | (keep != 0)).all()) and abs(float((got == 0).float().mean()) - p) < 0.02
    close(tag + ".bias_drop", got, base * keep, torch.bfloat16)
    close(tag + ".bias_drop_res", K.gemm(xd, wd, M, N, Kd, bias=bd, dropout_p=p, seed=seed, stream_id=site, residual=res.to(DEV)),
          base * keep + res.double(), torch.bfloat16)
    close(tag + ".bias_relu_drop", K.gemm(xd, wd, M, N, Kd, bias=bd, relu=True, dropout_p=p, seed=seed, stream_id=site),
          torch.relu(base) * keep, torch.bfloat16)
    bt = (xd.double() @ wtd.double().t()).cpu()
    close(tag + ".dgrad", K.gemm(xd, wtd, M, N, Kd, trans_b=True), bt, torch.bfloat16)
    close(tag + ".dgrad_res", K.gemm(xd, wtd, M, N, Kd, trans_b=True, residual=res.to(DEV)), bt + res.double(), torch.bfloat16)
    close(tag + ".dgrad_gate", K.gemm(xd, wtd, M, N, Kd, trans_b=True, gate_src=gate.to(DEV), gate_scale=1.25),
          bt * (gate.double() > 0) * 1.25, torch.bfloat16)
    if N % 64 == 0 and M % 64 == 0:      # row dots per 64-column head next to the product (attention output projection)
        T_ = 64
        o = rnd(M, N, dtype=torch.bfloat16, seed=8).to(DEV)
        delta = torch.full((M // T_, N // 64, T_), 7.0, device=DEV)
        do = K.gemm(xd, wtd, M, N, Kd, trans_b=True, rowdot=(o, delta, T_))
        close(tag + ".dgrad_rowdot", do, bt, torch.bfloat16)
        want = (do.float().view(M // T_, T_, N // 64, 64) * o.float().view(M // T_, T_, N // 64, 64)).sum(-1).permute(0, 2, 1)
        assert float((delta - want).abs().max()) <= 1e-4 * float(want.abs().max())
    # strided views: x is a column block of a wider buffer, the output a column block of a wider one (q|k|v packing)
    widex = rnd(M, Kd + 64, dtype=torch.bfloat16, seed=9).to(DEV)
    widec = torch.zeros(M, N + 24, dtype=torch.bfloat16, device=DEV)
    K.gemm(widex[:, 32:32 + Kd], wd, M, N, Kd, bias=bd, out=widec[:, 16:16 + N])
    close(tag + ".strided", widec[:, 16:16 + N], (widex[:, 32:32 + Kd].double() @ wd.double()).cpu() + bias.double(), torch.bfloat16)
    assert float(widec[:, :16].abs().max()) == 0.0 and float(widec[:, 16 + N:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K_", GEMM_SHAPES)
def test_gemm_plain(K, dtype, ta, tb, M, N, K_):
    A = rnd(*((K_, M) if ta else (M, K_)), dtype=dtype, seed=11)
    B = rnd(*((N, K_) if tb else (K_, N)), dtype=dtype, seed=12)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    C_ = K.gemm(A.to(DEV), B.to(DEV), M, N, K_, trans_a=ta, trans_b=tb)
    close(f"gemm[{dtype},{ta}{tb},{M}x{N}x{K_}]", C_, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(K, dtype):
    M, N, K_ = 150, 264, 96
    A, B = rnd(M, K_, dtype=dtype, seed=1), rnd(K_, N, dtype=dtype, seed=2)
    bias = rnd(N, seed=3)
    res = rnd(M, N, dtype=dtype, seed=4)
    gate = rnd(M, N, dtype=dtype, seed=5)
    pos = rnd(50, N, seed=6)
    base = A.double() @ B.double()
    Ad, Bd = A.to(DEV), B.to(DEV)
    out = K.gemm(Ad, Bd, M, N, K_, alpha=0.5, bias=bias.to(DEV), relu=True)
    close(f"gemm_bias_relu[{dtype}]", out, torch.relu(0.5 * base + bias.double()), dtype)
    out = K.gemm(Ad, Bd, M, N, K_, bias=bias.to(DEV), residual=res.to(DEV))
    close(f"gemm_residual[{dtype}]", out, base + bias.double() + res.double(), dtype)
    out = K.gemm(Ad, Bd, M, N, K_, gate_src=gate.to(DEV), gate_scale=1.25)
    close(f"gemm_gate[{dtype}]", out, base * (gate.double() > 0) * 1.25, dtype)
    out = K.gemm(Ad, Bd, M, N, K_, bias=bias.to(DEV), posenc=pos.to(DEV), posenc_period=50, emb_scale=4.0)
    rows = torch.arange(M) % 50
    close(f"gemm_posenc[{dtype}]", out, (base + bias.double()) * 4.0 + pos.double()[rows], dtype)
    c0 = rnd(M, N, dtype=dtype, seed=7)
    out = c0.to(DEV).clone()
    K.gemm(Ad, Bd, M, N, K_, out=out, accumulate=True)
    close(f"gemm_accumulate[{dtype}]", out, base + c0.double(), dtype)
    # strided views: A / out are column slices of wider buffers
    wideA = rnd(M, K_ + 40, dtype=dtype, seed=8).to(DEV)
    wideC = torch.zeros(M, N + 24, dtype=dtype, device=DEV)
    K.gemm(wideA[:, 8:8 + K_], Bd, M, N, K_, out=wideC[:, 16:16 + N])
    close(f"gemm_strided[{dtype}]", wideC[:, 16:16 + N], wideA[:, 8:8 + K_].cpu().double() @ B.double(), dtype)
    assert float(wideC[:, :16].abs().max()) == 0.0 and float(wideC[:, 16 + N:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_splitk_wgrad(K, dtype):
    # dW[Kin,N] = X^T[Kin,rows] @ dY[rows,N], fp32 output, reduction over many rows
    rows, Kin, N = 3000, 136, 264
    X, dY = rnd(rows, Kin, dtype=dtype, seed=1), rnd(rows, N, dtype=dtype, seed=2)
    ref = X.double().t() @ dY.double()
    for split in (1, 7, 64):
        out = torch.full((Kin, N), 3.0, dtype=torch.float32, device=DEV)
        K.gemm(X.to(DEV), dY.to(DEV), Kin, N, rows, trans_a=True, out=out, split_k=split)
        close(f"gemm_wgrad[{dtype},split{split}]", out, ref, torch.float32 if dtype == torch.float32 else dtype)
    out = torch.full((Kin, N), 3.0, dtype=torch.float32, device=DEV)
    K.gemm(X.to(DEV), dY.to(DEV), Kin, N, rows, trans_a=True, out=out, split_k=16, accumulate=True)
    close(f"gemm_wgrad_acc[{dtype}]", out, ref + 3.0, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,Kin,N", [(3000, 136, 264), (28800 // 8, 256, 768), (517, 2048, 256), (300, 130, 132),
                                        (1000, 520, 264), (2500, 1000, 520), (130, 256, 256), (64, 512, 256)])
def test_gemm_wgrad_fused_colsum(K, dtype, rows, Kin, N):
    """dW = X^T.dZ with db = colsum(dZ) produced by the same kernel (MFMA ones-row), all split factors, accumulate;
    the fourth shape is not 8-element granular and takes the separate column-sum pass behind the same interface.  bf16 outputs
    of at least 256 x 256 run on the phase-staggered 256 x 256 kernel (nst_gemm256.h): the shapes from (28800 // 8, 256, 768)
    on cover whole tiles, ragged edge tiles in both output dimensions (520 = 2 tiles + 8, 264 = 1 tile + 8), reduction
    lengths that are not a multiple of the K step (517, 1000, 130) and a single K step (64); split 5 / 32 exercise its even
    K slicing (32 slices of 2500 rows: one or two K steps each)."""
    X, dZ = rnd(rows, Kin, dtype=dtype, seed=1), rnd(rows, N, dtype=dtype, seed=2)
    ref_w, ref_b = X.double().t() @ dZ.double(), dZ.double().sum(0)
    for split in (1, 5, 32):
        dw = torch.full((Kin, N), 3.0, dtype=torch.float32, device=DEV)
        db = torch.full((N,), 5.0, dtype=torch.float32, device=DEV)
        K.gemm(X.to(DEV), dZ.to(DEV), Kin, N, rows, trans_a=True, out=dw, split_k=split, colsum_out=db)
        tag = f"gemm_wgrad_colsum[{dtype},{rows}x{Kin}x{N},split{split}]"
        close(tag + ".dw", dw, ref_w, dtype)
        close(tag + ".db", db, ref_b, torch.float32)   # sums of the (already rounded) inputs: f32 accuracy either way
        K.gemm(X.to(DEV), dZ.to(DEV), Kin, N, rows, trans_a=True, out=dw, split_k=split, accumulate=True, colsum_out=db,
               colsum_accumulate=True)
        close(tag + ".dw_acc", dw, 2 * ref_w, dtype)
        close(tag + ".db_acc", db, 2 * ref_b, torch.float32)


@pytest.mark.parametrize("nprob", [1, 7, 60])
def test_gemm_wgrad_group(K, nprob):
    """nst_gemm_wgrad_group: n weight gradients dW_i (+)= X_i^T dZ_i with db_i (+)= colsum(dZ_i) from ONE launch of the
    phase-staggered 256 x 256 kernel, every output tile of every product a workgroup (no split-K).  Products of different
    shapes and reduction lengths (one K step ... dozens, lengths that are not a multiple of 64), ragged edge tiles, column-block
    views of wider buffers (the packed q|k|v / k|v gradients), with and without the column sums, overwrite and accumulate;
    60 products take the device-side product table (more than the 56 that travel as kernel arguments).  Against fp64 on the
    same bf16 inputs: the kernel accumulates in fp32 over the whole reduction."""
    shapes = [(1800, 256, 2048), (1800, 2048, 256), (900, 256, 768), (1000, 520, 264), (130, 256, 256), (64, 512, 256),
              (2500, 1000, 520), (300, 128, 128), (77 * 8, 136, 264), (4096, 256, 512)]
    items, refs = [], []
    for i in range(nprob):
        rows, kin, n = shapes[i % len(shapes)]
        wide_x = rnd(rows, kin + 16, dtype=torch.bfloat16, seed=100 + i).to(DEV)
        wide_z = rnd(rows, n + 24, dtype=torch.bfloat16, seed=200 + i).to(DEV)
        x, dz = wide_x[:, 8:8 + kin], wide_z[:, 16:16 + n]           # strided views, 16-byte aligned starts
        acc = bool(i % 2)
        dw = torch.full((kin, n + 8), 3.0, device=DEV)[:, :n]        # ldc > N
        has_b = i % 3 != 2
        db = torch.full((n,), 5.0, device=DEV) if has_b else None
        items.append((x, dz, dw, acc, db, acc))
        ref_w = x.cpu().double().t() @ dz.cpu().double() + (3.0 if acc else 0.0)
        ref_b = dz.cpu().double().sum(0) + (5.0 if acc else 0.0) if has_b else None
        refs.append((ref_w, ref_b))
    g = K.WgradGroup(torch.device(DEV))
    for it in items:
        assert g.accepts(it[0], it[1], it[2], it[4])
        g.add(*it)
    g.launch()
    assert len(g) == 0
    for i, ((x, dz, dw, acc, db, _), (ref_w, ref_b)) in enumerate(zip(items, refs)):
        tag = f"gemm_wgrad_group[{nprob}].{i}[{x.shape[0]}x{x.shape[1]}x{dz.shape[1]}]"
        close(tag + ".dw", dw, ref_w, torch.float32)     # fp32 accumulation of exact bf16 products: the fp32 bounds apply
        if db is not None:
            close(tag + ".db", db, ref_b, torch.float32)
    # neighbours of the ldc > N views stay untouched
    assert float(items[0][2]._base[:, -8:].min()) == 3.0 and float(items[0][2]._base[:, -8:].max()) == 3.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_dropout_epilogue(K, dtype):
    M, N, K_ = 256, 256, 64
    A, B = rnd(M, K_, dtype=dtype, seed=1), rnd(K_, N, dtype=dtype, seed=2)
    base = (A.double() @ B.double())
    out = K.gemm(A.to(DEV), B.to(DEV), M, N, K_, dropout_p=0.25, seed=123, stream_id=9).float().cpu().double()
    kept = out != 0
    frac = float(kept.double().mean())
    REPORT[f"gemm_dropout_keepfrac[{dtype}]"] = frac
    assert abs(frac - 0.75) < 0.02
    close(f"gemm_dropout_vals[{dtype}]", torch.where(kept, out, torch.zeros_like(out)),
          torch.where(kept, base / 0.75, torch.zeros_like(base)), dtype)
    # backward mask regeneration: scale_dropout_bwd with the same (seed, stream) gives the same mask
    ones = torch.ones(M, N, dtype=dtype, device=DEV)
    mask = K.scale_dropout_bwd(ones, 1.0, 0.25, 123, 9).float().cpu()
    assert torch.equal(mask != 0, kept)
    out2 = K.gemm(A.to(DEV), B.to(DEV), M, N, K_, dropout_p=0.25, seed=123, stream_id=10).float().cpu()
    assert not torch.equal(out2 != 0, kept)


def test_colsum(K):
    for dtype in DTYPES:
        x = rnd(1000, 264, dtype=dtype, seed=3)
        out = torch.full((264,), 2.0, device=DEV)
        K.colsum(x.to(DEV), out)
        close(f"colsum[{dtype}]", out, x.double().sum(0), dtype)
        K.colsum(x.to(DEV), out, accumulate=True)
        close(f"colsum_acc[{dtype}]", out, 2 * x.double().sum(0), dtype)


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, bias, causal, H, dh, keep=None, p=0.0):
    B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
    q4, k4, v4 = q.reshape(B, Tq, H, dh), k.reshape(B, Tk, H, dh), v.reshape(B, Tk, H, dh)
    logits = torch.einsum("bthd,bfhd->bhft", k4, q4 * dh ** -0.5)
    if bias is not None:
        logits = logits + bias[:, None, None, :]
    if causal:
        logits = logits + O.lower_triangle_attention_bias(Tq, q.dtype)
    w = torch.softmax(logits, -1)
    if keep is not None:
        w = w * keep / (1.0 - p)
    return torch.einsum("bhft,bthd->bfhd", w, v4).reshape(B, Tq, H * dh)


ATTN_CASES = [  # B, H, Tq, Tk, dh, causal, key-padding
    (1, 2, 2, 2, 2, False, False), (2, 2, 5, 7, 4, False, True), (2, 2, 3, 3, 4, True, False),
    (2, 4, 225, 225, 64, False, True), (2, 4, 75, 75, 64, True, False), (2, 4, 75, 225, 64, False, True),
    (1, 3, 130, 70, 24, False, True), (1, 2, 64, 64, 64, True, False)]


@pytest.fixture(params=[1, 2, "fused_bwd"])
def attn_mi(request, monkeypatch):
    """Force the number of 16-row blocks per wave (64*MI rows per workgroup) in the three tiled attention kernels; "fused_bwd":
    the one-workgroup-per-head backward (attn_bwd_head8_kernel: bf16, sequences <= 256 -- the library's default for such
    shapes; other cases fall back to the tiled kernels by themselves)."""
    mi = 1 if request.param == "fused_bwd" else request.param
    for k in ("NST_ATTN_MI_FWD", "NST_ATTN_MI_DKDV", "NST_ATTN_MI_DQ"):
        monkeypatch.setenv(k, str(mi))
    monkeypatch.setenv("NST_ATTN_FUSED_BWD", "1" if request.param == "fused_bwd" else "0")
    return mi


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Tq,Tk,dh,causal,pad", ATTN_CASES)
def test_attention(K, dtype, attn_mi, B, H, Tq, Tk, dh, causal, pad):
    d = H * dh
    self_att = Tq == Tk
    bias = None
    if pad:
        lens = torch.tensor([Tk - (i * Tk) // (2 * B) for i in range(B)])
        bias = (O.length_to_padding(lens, Tk) * O.FLOAT_MIN).float()
    if self_att:  # packed q|k|v projection output, as produced by the qkv GEMM
        qkv = rnd(B, Tq, 3 * d, dtype=dtype, seed=5)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        qkv_d = qkv.to(DEV)
        qd, kd, vd = qkv_d[..., :d], qkv_d[..., d:2 * d], qkv_d[..., 2 * d:]
    else:
        q = rnd(B, Tq, d, dtype=dtype, seed=5)
        kv = rnd(B, Tk, 2 * d, dtype=dtype, seed=6)
        k, v = kv[..., :d], kv[..., d:]
        qd = q.to(DEV)
        kv_d = kv.to(DEV)
        kd, vd = kv_d[..., :d], kv_d[..., d:]
    dout = rnd(B, Tq, d, dtype=dtype, seed=7)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, None if bias is None else bias.double(), causal, H, dh)
    ref.backward(dout.double())
    out, lse, _ = K.attention_fwd(qd, kd, vd, H, dh, key_bias=None if bias is None else bias.to(DEV), causal=causal)
    tag = f"attn[{dtype},mi{attn_mi},B{B}H{H}q{Tq}k{Tk}d{dh}{'c' if causal else ''}{'p' if pad else ''}]"
    close(tag + ".out", out, ref, dtype)
    if self_att:
        dqkv = torch.zeros_like(qkv_d)
        dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
    else:
        dq = torch.zeros_like(qd)
        dkv = torch.zeros_like(kv_d)
        dk, dv = dkv[..., :d], dkv[..., d:]
    K.attention_bwd(qd, kd, vd, out, dout.to(DEV), lse, dq, dk, dv, H, dh,
                    key_bias=None if bias is None else bias.to(DEV), causal=causal)
    close(tag + ".dq", dq, qr.grad, dtype, scale=3.0)
    close(tag + ".dk", dk, kr.grad, dtype, scale=3.0)
    close(tag + ".dv", dv, vr.grad, dtype, scale=3.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,Tq,Tk,dh,off,pad,drop", [(2, 2, 5, 9, 4, 0, False, 0.0), (2, 2, 7, 20, 8, 2, True, 0.0),
                                                      (2, 4, 75, 225, 64, 4, True, 0.0), (1, 2, 130, 200, 64, 70, True, 0.0),
                                                      (2, 2, 40, 100, 32, 8, False, 0.0), (1, 3, 90, 60, 16, 1, True, 0.0),
                                                      (2, 2, 64, 160, 64, 2, True, 0.2)])
def test_attention_waitk_offset(K, dtype, attn_mi, B, H, Tq, Tk, dh, off, pad, drop):
    """causal_offset (NstAttnDesc): key j masked for query i when j > i + off -- the wait-k cross-attention bias
    band_part(ones[Tq, Tk], -1, off) of layer_utils.py:56-78 -- together with key padding, forward and backward."""
    d = H * dh
    bias = None
    if pad:
        lens = torch.tensor([Tk - (i * Tk) // (3 * B) for i in range(B)])
        bias = (O.length_to_padding(lens, Tk) * O.FLOAT_MIN).float()
    q = rnd(B, Tq, d, dtype=dtype, seed=5)
    kv = rnd(B, Tk, 2 * d, dtype=dtype, seed=6)
    k, v = kv[..., :d], kv[..., d:]
    dout = rnd(B, Tq, d, dtype=dtype, seed=7)
    qd, kvd = q.to(DEV), kv.to(DEV)
    kd, vd = kvd[..., :d], kvd[..., d:]
    bd = None if bias is None else bias.to(DEV)
    out, lse, mask = K.attention_fwd(qd, kd, vd, H, dh, key_bias=bd, causal=True, causal_offset=off, dropout_p=drop, seed=3, stream_id=1)
    band = O.waitk_attention_bias(Tk, off + 1, Tq, torch.float64)                       # [Tq, Tk]
    assert band[0, min(off, Tk - 1)] == 0 and (off + 1 >= Tk or band[0, off + 1] < 0)
    full = band[None, :, :] if bias is None else torch.minimum(bias.double()[:, None, :], band[None, :, :])
    keep = None
    if drop > 0:   # recover the kept set from the kernel's own output is not possible here: check dropout-free rows instead
        out0, _, _ = K.attention_fwd(qd, kd, vd, H, dh, key_bias=bd, causal=True, causal_offset=off)
        assert not torch.equal(out0, out) and torch.isfinite(out.float()).all()
        out, lse, mask, drop = out0, K.attention_fwd(qd, kd, vd, H, dh, key_bias=bd, causal=True, causal_offset=off)[1], None, 0.0
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    q4, k4, v4 = qr.reshape(B, Tq, H, dh), kr.reshape(B, Tk, H, dh), vr.reshape(B, Tk, H, dh)
    w = torch.softmax(torch.einsum("bthd,bfhd->bhft", k4, q4 * dh ** -0.5) + full[:, None], -1)
    ref = torch.einsum("bhft,bthd->bfhd", w, v4).reshape(B, Tq, d)
    ref.backward(dout.double())
    tag = f"attn_waitk[{dtype},mi{attn_mi},q{Tq}k{Tk}d{dh}o{off}{'p' if pad else ''}]"
    close(tag + ".out", out, ref, dtype)
    dq, dkv = torch.zeros_like(qd), torch.zeros_like(kvd)
    K.attention_bwd(qd, kd, vd, out, dout.to(DEV), lse, dq, dkv[..., :d], dkv[..., d:], H, dh, key_bias=bd, causal=True,
                    causal_offset=off)
    close(tag + ".dq", dq, qr.grad, dtype, scale=3.0)
    close(tag + ".dk", dkv[..., :d], kr.grad, dtype, scale=3.0)
    close(tag + ".dv", dkv[..., d:], vr.grad, dtype, scale=3.0)
    # keys no query may see get exactly zero gradient
    hidden = (band.max(0).values < 0)
    if hidden.any():
        assert float(dkv[:, hidden].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,T,dh,causal", [(2, 2, 32, 32, False), (2, 2, 64, 64, True), (1, 4, 64, 64, False)])
def test_attention_dropout(K, dtype, attn_mi, B, H, T, dh, causal):
    # V = identity (T == dh) exposes the dropped probability matrix as the output, which yields the mask itself;
    # the backward kernels must then reproduce the gradients of exactly that mask from the saved bits.
    p = 0.3
    d = H * dh
    q, k = rnd(B, T, d, dtype=dtype, seed=1), rnd(B, T, d, dtype=dtype, seed=2)
    if T == dh:
        eye = torch.eye(T).reshape(1, T, 1, dh).expand(B, T, H, dh).reshape(B, T, d).to(dtype).contiguous()
        out, _, _ = K.attention_fwd(q.to(DEV), k.to(DEV), eye.to(DEV), H, dh, causal=causal, dropout_p=p, seed=77,
                                    stream_id=3)
        P = out.float().cpu().reshape(B, T, H, T).permute(0, 2, 1, 3)  # [B,H,Tq,Tk]
        keep = (P != 0).double()
        visible = torch.tril(torch.ones(T, T)) if causal else torch.ones(T, T)
        frac = float((keep * visible).sum() / (visible.sum() * B * H))
        REPORT[f"attn_dropout_keepfrac[{dtype},T{T}{'c' if causal else ''}]"] = frac
        assert abs(frac - (1 - p)) < 0.03
        if causal:
            keep = keep + (1 - visible)  # masked positions carry probability 0 either way
    v = rnd(B, T, d, dtype=dtype, seed=3)
    dout = rnd(B, T, d, dtype=dtype, seed=4)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, None, causal, H, dh, keep=keep, p=p)
    ref.backward(dout.double())
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    out, lse, mask = K.attention_fwd(qd, kd, vd, H, dh, causal=causal, dropout_p=p, seed=77, stream_id=3)
    tag = f"attn_dropout[{dtype},mi{attn_mi},T{T}{'c' if causal else ''}]"
    close(tag + ".out", out, ref, dtype)
    dq, dk, dv = torch.zeros_like(qd), torch.zeros_like(kd), torch.zeros_like(vd)
    K.attention_bwd(qd, kd, vd, out, dout.to(DEV), lse, dq, dk, dv, H, dh, causal=causal, dropout_p=p, seed=77,
                    stream_id=3, drop_mask=mask)
    close(tag + ".dq", dq, qr.grad, dtype, scale=3.0)
    close(tag + ".dk", dk, kr.grad, dtype, scale=3.0)
    close(tag + ".dv", dv, vr.grad, dtype, scale=3.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_dropout_long(K, dtype, attn_mi):
    """Several key tiles and query blocks (Tq=150, Tk=200, cross attention with padding): the mask is recovered from
    the bits the forward kernel saved, decoded with the documented layout, and fed to the float64 reference."""
    B, H, Tq, Tk, dh, p = 2, 2, 150, 200, 64, 0.25
    d = H * dh
    q = rnd(B, Tq, d, dtype=dtype, seed=11)
    kv = rnd(B, Tk, 2 * d, dtype=dtype, seed=12)
    k, v = kv[..., :d], kv[..., d:]
    lens = torch.tensor([Tk, Tk - 37])
    bias = (O.length_to_padding(lens, Tk) * O.FLOAT_MIN).float()
    qd, kvd = q.to(DEV), kv.to(DEV)
    kd, vd = kvd[..., :d], kvd[..., d:]
    out, lse, mask = K.attention_fwd(qd, kd, vd, H, dh, key_bias=bias.to(DEV), dropout_p=p, seed=5, stream_id=9)
    nqb, nkt = (Tq + 15) // 16, (Tk + 63) // 64
    words = mask.cpu().view(torch.int16).to(torch.int32).bitwise_and(0xffff).reshape(B, H, nqb, nkt, 64)
    e = torch.arange(16)
    bits = (words[..., None] >> e) & 1                                  # [B,H,nqb,nkt,lane,e]
    bits = bits.reshape(B, H, nqb, nkt, 4, 16, 4, 4)                     # lane = g*16 + lc ; e = f*4 + r
    # -> [B,H,nqb,lc,nkt,f,g,r] : query = qb*16 + lc ; key = kt*64 + f*16 + g*4 + r
    keep = bits.permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(B, H, nqb * 16, nkt * 64)[:, :, :Tq, :Tk].double()
    frac = float(keep.mean())
    REPORT[f"attn_dropout_long_keepfrac[{dtype}]"] = frac
    assert abs(frac - (1 - p)) < 0.01
    dout = rnd(B, Tq, d, dtype=dtype, seed=13)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, bias.double(), False, H, dh, keep=keep, p=p)
    ref.backward(dout.double())
    tag = f"attn_dropout_long[{dtype},mi{attn_mi}]"
    close(tag + ".out", out, ref, dtype)
    dq, dkv = torch.zeros_like(qd), torch.zeros_like(kvd)
    K.attention_bwd(qd, kd, vd, out, dout.to(DEV), lse, dq, dkv[..., :d], dkv[..., d:], H, dh, key_bias=bias.to(DEV),
                    dropout_p=p, seed=5, stream_id=9, drop_mask=mask)
    close(tag + ".dq", dq, qr.grad, dtype, scale=3.0)
    close(tag + ".dk", dkv[..., :d], kr.grad, dtype, scale=3.0)
    close(tag + ".dv", dkv[..., d:], vr.grad, dtype, scale=3.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_over_cache_prefix(K, dtype):
    """k / v given as the filled prefix [:, :t] of a longer [B, Tmax, d] cache (NstAttnDesc.bsk / bsv): the decoding step's
    view must give what a dense copy of the prefix gives, bit for bit."""
    B, H, dh, Tmax = 3, 4, 16, 40
    d = H * dh
    kc, vc = rnd(B, Tmax, d, dtype=dtype, seed=1).to(DEV), rnd(B, Tmax, d, dtype=dtype, seed=2).to(DEV)
    for t in (1, 7, 16, 33):
        q = rnd(B, 1, d, dtype=dtype, seed=10 + t).to(DEV)
        got, _, _ = K.attention_fwd(q, kc[:, :t], vc[:, :t], H, dh)
        want, _, _ = K.attention_fwd(q, kc[:, :t].contiguous(), vc[:, :t].contiguous(), H, dh)
        assert torch.equal(got, want)
        qf, kf, vf = q.double().view(B, 1, H, dh), kc[:, :t].double().view(B, t, H, dh), vc[:, :t].double().view(B, t, H, dh)
        p = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", qf, kf) * dh ** -0.5, -1)
        close(f"attn_cache[{dtype},t{t}]", got, torch.einsum("bhqk,bkhd->bqhd", p, vf).reshape(B, 1, d).cpu(), dtype)


# ------------------------------------------------------------------------------------------------ conv front end
def _conv1_ref(src, w1, b1, gamma, beta, ln):
    x = torch.nn.functional.conv2d(src[:, None], w1.permute(3, 2, 0, 1), b1, stride=2, padding=1)  # [B,C,T1,F1]
    x = x.permute(0, 2, 3, 1)
    if ln:
        x = O.layer_norm(x, gamma, beta, 1e-6)
    return torch.relu(x)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,F,C,ln", [(1, 11, 80, 5, True), (2, 33, 16, 8, True), (2, 14, 12, 6, False),
                                        (2, 50, 80, 256, True), (1, 9, 20, 512, True), (3, 23, 16, 64, False),
                                        # C = 256 takes the second (eight-wave, packed-f32) backward: without LayerNorm, and
                                        # with more 32-pixel groups (2500) than waves (2048: a wave re-stages its one tile buffer)
                                        (2, 37, 24, 256, False), (20, 200, 80, 256, True),
                                        # an odd output width, 462 pixels: the bf16 forward's 16-pixel groups straddle rows, the last is ragged
                                        (2, 21, 42, 256, True)])
def test_conv1(K, dtype, B, T, F, C, ln):
    if B == 20 and dtype == torch.float32:
        pytest.skip("the 80 000-pixel case exists for the bf16 kernel's tile re-staging; in fp32 the sums of 80 000 random-sign "
                    "terms cancel to ~1e-4 relative whatever the kernel does")
    src = rnd(B, T, F, seed=1)
    w1 = rnd(3, 3, 1, C, seed=2) * 0.4
    b1 = rnd(C, seed=3) * 0.1
    gamma, beta = rnd(C, seed=4) * 0.2 + 1.0, rnd(C, seed=5) * 0.1
    T1, F1 = (T + 1) // 2, (F + 1) // 2
    dout = rnd(B, T1, F1, C, dtype=dtype, seed=6)
    pr = [t.double().requires_grad_(True) for t in (w1, b1, gamma, beta)]
    ref = _conv1_ref(src.double(), *pr, ln)
    ref.backward(dout.double())
    dv = lambda t: t.to(DEV)
    out, mean, rstd = K.conv1_ln_relu_fwd(dv(src), dv(w1), dv(b1), dv(gamma), dv(beta), ln, 1e-6, dtype)
    tag = f"conv1[{dtype},B{B}T{T}F{F}C{C}{'ln' if ln else ''}]"
    close(tag + ".out", out, ref, dtype)
    if ln:   # the saved statistics are fp32 whatever the output type (the bf16 C = 256 forward computes the taps as hi/lo bf16
        # products on the matrix cores: ~2^-16 relative per product)
        pre = torch.nn.functional.conv2d(src.double()[:, None], w1.double().permute(3, 2, 0, 1), b1.double(), stride=2,
                                         padding=1).permute(0, 2, 3, 1)
        m_ref, v_ref = pre.mean(-1), pre.var(-1, unbiased=False)
        assert (mean.cpu().double().reshape(m_ref.shape) - m_ref).abs().max() < 2e-5, tag
        assert ((rstd.cpu().double().reshape(m_ref.shape) * (v_ref + 1e-6).sqrt()) - 1).abs().max() < 1e-4, tag
    dw1, db1 = torch.full((3, 3, 1, C), 5.0, device=DEV), torch.full((C,), 5.0, device=DEV)
    dg, dbe = torch.full((C,), 5.0, device=DEV), torch.full((C,), 5.0, device=DEV)
    K.conv1_ln_relu_bwd(dv(src), dv(w1), dv(b1), dv(gamma), dv(beta), mean, rstd, dv(dout), dw1, db1, dg, dbe, ln, 1e-6)
    close(tag + ".dw1", dw1, pr[0].grad, dtype, scale=2.0)
    close(tag + ".db1", db1, pr[1].grad, dtype, scale=2.0)
    if ln:
        close(tag + ".dgamma", dg, pr[2].grad, dtype, scale=2.0)
        close(tag + ".dbeta", dbe, pr[3].grad, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T1,F1,C", [(1, 6, 40, 5), (2, 9, 7, 8), (2, 12, 8, 64), (2, 25, 40, 256), (1, 5, 6, 40)])
def test_conv2(K, dtype, B, T1, F1, C):
    x = rnd(B, T1, F1, C, dtype=dtype, seed=1)
    w2 = (rnd(3, 3, C, C, seed=2) * (1.0 / math.sqrt(9 * C))).to(dtype)
    b2 = rnd(C, seed=3) * 0.1
    T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
    dy = rnd(B, T2, F2, C, dtype=dtype, seed=4)
    xr, wr = x.double().requires_grad_(True), w2.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1), b2.double(), stride=2, padding=1)
    ref = ref.permute(0, 2, 3, 1)
    ref.backward(dy.double())
    tag = f"conv2[{dtype},B{B}T{T1}F{F1}C{C}]"
    y = K.conv2_fwd(x.to(DEV), w2.to(DEV), b2.to(DEV))
    close(tag + ".y", y, ref, dtype)
    dx = K.conv2_dgrad(dy.to(DEV), w2.to(DEV), T1, F1)
    close(tag + ".dx", dx, xr.grad, dtype, scale=2.0)
    dw2 = torch.full((3, 3, C, C), 2.0, device=DEV)
    db2 = torch.full((C,), 2.0, device=DEV)
    db_ref = dy.double().sum((0, 1, 2))
    K.conv2_wgrad(x.to(DEV), dy.to(DEV), dw2, db2=db2)
    close(tag + ".dw2", dw2, wr.grad, dtype, scale=2.0)
    close(tag + ".db2", db2, db_ref, torch.float32)
    K.conv2_wgrad(x.to(DEV), dy.to(DEV), dw2, db2=db2, accumulate=True)
    close(tag + ".dw2_acc", dw2, 2 * wr.grad, dtype, scale=2.0)
    close(tag + ".db2_acc", db2, 2 * db_ref, torch.float32)
    dw3 = torch.full((3, 3, C, C), 2.0, device=DEV)
    K.conv2_wgrad(x.to(DEV), dy.to(DEV), dw3)   # without the bias gradient
    close(tag + ".dw2_nobias", dw3, wr.grad, dtype, scale=2.0)


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("B,T1,F1", [(2, 8, 6), (3, 50, 40), (5, 34, 7), (2, 128, 41), (4, 2, 2), (1, 450, 40), (9, 450, 40),
                                     (7, 300, 41)])
def test_conv2_c256_kernels(K, B, T1, F1, relu):
    """bf16, C == 256: the conv2 kernels on the 256 x 256 tile core (nst_conv.hip conv2_fwd256 / conv2_dgrad256 /
    conv2_wgrad256 kernels) and their fallbacks for small or odd grids -- images that share a tile, ragged last tiles, odd /
    even widths, the top padding row of every image, > 256 tiles, reductions cut into 19 and 13 slices with a ragged last
    K step."""
    C, dtype = 256, torch.bfloat16
    x = rnd(B, T1, F1, C, dtype=dtype, seed=11)
    w2 = (rnd(3, 3, C, C, seed=12) * (1.0 / math.sqrt(9 * C))).to(dtype)
    b2 = rnd(C, seed=13) * 0.1
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w2.double().permute(3, 2, 0, 1), b2.double(), stride=2,
                                     padding=1).permute(0, 2, 3, 1)
    if relu:
        ref = ref.relu()
    y = K.conv2_fwd(x.to(DEV), w2.to(DEV), b2.to(DEV), relu=relu)
    close(f"conv2_c256[B{B}T{T1}F{F1}relu{int(relu)}].y", y, ref, dtype)
    y2 = K.conv2_fwd(x.to(DEV), w2.to(DEV), b2.to(DEV), relu=relu)
    assert torch.equal(y, y2), "conv2 forward is not deterministic"
    if not relu:
        # data gradient: even T1 and F1 (>= 256 output pixels) take conv2_dgrad256_kernel (all four parity classes from one
        # workgroup), the odd width falls back to the per-class implicit GEMMs -- both against autograd of the same convolution
        T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
        dy = rnd(B, T2, F2, C, dtype=dtype, seed=14)
        xr = x.double().requires_grad_(True)
        torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), w2.double().permute(3, 2, 0, 1), None, stride=2,
                                   padding=1).permute(0, 2, 3, 1).backward(dy.double())
        dx = K.conv2_dgrad(dy.to(DEV), w2.to(DEV), T1, F1)
        close(f"conv2_c256[B{B}T{T1}F{F1}].dx", dx, xr.grad, dtype, scale=2.0)
        assert torch.equal(dx, K.conv2_dgrad(dy.to(DEV), w2.to(DEV), T1, F1))
        # weight gradient: conv2_wgrad256_kernel when the reduction fills the chip in slices of >= 32 K steps (the two
        # largest grids), the generic split-K kernels otherwise (pixels advance 64 per K step: image wraps, the top padding
        # row, ragged last K step, widths smaller than a K step)
        wr = w2.double().requires_grad_(True)
        torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1), None, stride=2,
                                   padding=1).permute(0, 2, 3, 1).backward(dy.double())
        dw2, db2 = torch.full((3, 3, C, C), 2.0, device=DEV), torch.full((C,), 2.0, device=DEV)
        K.conv2_wgrad(x.to(DEV), dy.to(DEV), dw2, db2=db2)
        close(f"conv2_c256[B{B}T{T1}F{F1}].dw2", dw2, wr.grad, dtype, scale=2.0)
        close(f"conv2_c256[B{B}T{T1}F{F1}].db2", db2, dy.double().sum((0, 1, 2)), torch.float32)
        first = dw2.clone()
        K.conv2_wgrad(x.to(DEV), dy.to(DEV), dw2, db2=db2, accumulate=True)     # += : deterministic, so exactly twice
        assert torch.equal(dw2, 2 * first)
        close(f"conv2_c256[B{B}T{T1}F{F1}].db2_acc", db2, 2 * dy.double().sum((0, 1, 2)), torch.float32)
        dw3 = torch.full((3, 3, C, C), 7.0, device=DEV)
        K.conv2_wgrad(x.to(DEV), dy.to(DEV), dw3)                               # without the bias gradient
        assert torch.equal(dw3, first)


def test_conv2_kernels_at_the_benchmark_grid(K):
    """The conv2 kernels at the grid bench.py times (B = 128, T1 = 450, F1 = 40, C = 256: 576 000 output pixels, 4 500 patch
    workgroups, the 56-slice XCD-pinned wide weight gradient).  A float64 convolution of this size takes minutes on the
    host, so the reference here is the library's own exact-fp32 path (implicit GEMM on v_mfma_f32_16x16x4_f32), which the
    cases above pin on float64 autograd at every smaller grid; bf16 tolerance 1e-2 relative to the tensor's magnitude."""
    B, T1, F1, C = 128, 450, 40, 256
    bf = torch.bfloat16
    x = rnd(B, T1, F1, C, dtype=bf, seed=21).to(DEV)
    w2 = (rnd(3, 3, C, C, seed=22) * (1.0 / math.sqrt(9 * C))).to(bf).to(DEV)
    b2 = (rnd(C, seed=23) * 0.1).to(DEV)
    T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
    dy = rnd(B, T2, F2, C, dtype=bf, seed=24).to(DEV)
    y = K.conv2_fwd(x, w2, b2)
    y32 = K.conv2_fwd(x.float(), w2.float(), b2)
    close("conv2_bench_grid.y", y, y32.cpu(), bf)
    del y, y32
    dx = K.conv2_dgrad(dy, w2, T1, F1)
    dx32 = K.conv2_dgrad(dy.float(), w2.float(), T1, F1)
    close("conv2_bench_grid.dx", dx, dx32.cpu(), bf, scale=2.0)
    del dx, dx32
    dw, db = torch.zeros(3, 3, C, C, device=DEV), torch.zeros(C, device=DEV)
    dw32, db32 = torch.zeros(3, 3, C, C, device=DEV), torch.zeros(C, device=DEV)
    K.conv2_wgrad(x, dy, dw, db2=db)
    K.conv2_wgrad(x.float(), dy.float(), dw32, db2=db32)
    close("conv2_bench_grid.dw2", dw, dw32.cpu(), bf, scale=2.0)
    close("conv2_bench_grid.db2", db, db32.cpu(), torch.float32)
    close("conv2_bench_grid.db2_vs_sum", db, dy.float().sum((0, 1, 2)).cpu(), torch.float32)


# ------------------------------------------------------------------------------------------------ embedding / elementwise
@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding(K, dtype):
    V, d, B, L = 50, 24, 3, 7
    table = rnd(V, d, dtype=dtype, seed=1)
    ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(2))
    pos = O.sinusoid_signal(L, d)
    ref = O.position_embedding(O.word_embedding(ids, table.double()))
    out = K.embedding_fwd(table.to(DEV), ids.to(DEV), pos.to(DEV), L, d ** 0.5)
    close(f"embedding[{dtype}].fwd", out, ref, dtype)
    dout = rnd(B, L, d, dtype=dtype, seed=3)
    dtab = torch.full((V, d), 1.0, device=DEV)
    K.embedding_bwd(dout.to(DEV), ids.to(DEV), dtab, d ** 0.5)
    refg = torch.ones(V, d, dtype=torch.float64)
    refg.index_put_((ids.reshape(-1),), dout.double().reshape(-1, d) * d ** 0.5, accumulate=True)
    close(f"embedding[{dtype}].bwd", dtab, refg, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_scale_posenc_dropout(K, dtype):
    B, T, d = 3, 10, 24
    x = rnd(B, T, d, dtype=dtype, seed=1)
    pos = O.sinusoid_signal(T, d)
    y = K.scale_posenc_dropout_fwd(x.to(DEV), pos.to(DEV), T, d ** 0.5)
    close(f"scale_posenc[{dtype}]", y, O.position_embedding(x.double()), dtype)
    y = K.scale_posenc_dropout_fwd(x.to(DEV), None, 1, 1.0, dropout_p=0.5, seed=5, stream_id=1).float().cpu()
    kept = y != 0
    assert abs(float(kept.float().mean()) - 0.5) < 0.1
    g = K.scale_dropout_bwd(torch.ones_like(x).to(DEV), 1.0, 0.5, 5, 1).float().cpu()
    assert torch.equal(g != 0, kept)
    close(f"dropout_scale[{dtype}]", y[kept], x.double()[kept] * 2.0, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,V,ls", [(6, 5, 0.1), (64, 8008, 0.1), (10, 1000, 0.0), (33, 300, 0.3), (4, 16384, 0.1), (3, 40000, 0.1)])
def test_ls_xent(K, dtype, rows, V, ls):
    logits = (rnd(rows, V, seed=1) * 2.0).to(dtype)
    labels = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(2))
    B = 2 if rows % 2 == 0 else 1
    L = rows // B
    lens = torch.tensor([L - i for i in range(B)])
    lr = logits.double().reshape(B, L, V).requires_grad_(True)
    nll, _, ntok = O.label_smoothed_cross_entropy(lr, labels.reshape(B, L), lens, ls)
    loss = O.reduce_loss(nll, ntok)
    loss.backward()
    weights = (1.0 - O.length_to_padding(lens, L)).reshape(-1).float()
    xent, lse = K.ls_xent_fwd(logits.to(DEV), labels.to(DEV), weights.to(DEV), ls)
    tag = f"xent[{dtype},{rows}x{V},ls{ls}]"
    close(tag + ".nll", xent.reshape(B, L).sum(1), nll, torch.float32, scale=5.0)
    dl = K.ls_xent_bwd(logits.to(DEV), labels.to(DEV), weights.to(DEV), lse, ls, 1.0 / float(ntok.sum()))
    close(tag + ".dlogits", dl, lr.grad.reshape(rows, V), dtype)


def test_seq_mask_and_xent_reduce(K):
    """nst_seq_mask (padding / non-padding / attention-bias masks from lengths, with the conv-subsampled lengths of
    speech_transformer.py:179-189 folded in) and nst_xent_reduce (the criterion's reductions) against the torch expressions of
    the reference functions (model_utils.py:44-75, layer_utils.py:19-32, label_smoothed_cross_entropy.py:46-53)."""
    lens = torch.tensor([900, 1, 0, 37, 451, 899, 2, 450])
    for T, halv in ((900, 0), (225, 2), (5, 0), (57, 1)):
        ln = lens.clone()
        for _ in range(halv):
            ln = (ln + 1) // 2
        inside = torch.arange(T)[None, :] < ln[:, None]
        for tok, pad in ((1.0, 0.0), (0.0, 1.0), (0.0, float(K.FLOAT_MIN))):
            got = K.seq_mask(lens.to(DEV), T, tok, pad, halvings=halv).cpu()
            want = torch.where(inside, torch.tensor(tok), torch.tensor(pad))
            assert torch.equal(got, want), (T, halv, tok, pad)
    from neurst_amd.layers import layer_utils
    from neurst_amd.models.model_utils import input_length_to_padding
    padding = input_length_to_padding(lens.to(DEV), 225, halvings=2)
    assert torch.equal(layer_utils.input_padding_to_bias(padding).cpu(), padding.cpu() * float(K.FLOAT_MIN))
    for B, L in ((128, 75), (3, 1), (300, 7)):
        xent, w = rnd(B, L, seed=4).abs(), (rnd(B, L, seed=5) > 0).float()
        w[:, 0] = 1.0
        nll, tok, loss, inv = K.xent_reduce(xent.to(DEV), w.to(DEV))
        close(f"xent_reduce[{B}x{L}].nll", nll, xent.double().sum(1), torch.float32)
        assert torch.equal(tok.cpu(), w.sum(1))
        want = float(xent.double().sum() / w.double().sum())
        assert abs(float(loss) - want) <= 1e-5 * abs(want) and abs(float(inv) - 1.0 / float(w.sum())) <= 1e-6 / float(w.sum())


def test_adam_and_cast(K):
    n = 10007
    p, g = rnd(n, seed=1), rnd(n, seed=2)
    m, v = rnd(n, seed=3) * 0.1, rnd(n, seed=4).abs() * 0.01
    lr, t = 3e-4, 7
    pr, mr, vr = O.keras_adam_step(p.double(), g.double() * 0.5, m.double(), v.double(), t, lr)
    pd, md, vd, gd = p.to(DEV), m.to(DEV), v.to(DEV), g.to(DEV)
    shadow = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    lr_t = lr * math.sqrt(1 - 0.98 ** t) / (1 - 0.9 ** t)
    K.adam_update(pd, md, vd, gd, shadow, lr_t, 0.9, 0.98, 1e-9, grad_scale=0.5)
    close("adam.p", pd, pr, torch.float32, scale=0.01)
    close("adam.m", md, mr, torch.float32, scale=0.01)
    close("adam.v", vd, vr, torch.float32, scale=0.01)
    assert torch.equal(shadow.cpu(), pd.cpu().to(torch.bfloat16))
    out = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    K.cast_f32_to_bf16(gd, out)
    assert torch.equal(out.cpu(), g.to(torch.bfloat16))


def test_errors_are_reported_not_fatal(K):
    from neurst_amd._lib import NstError
    with pytest.raises(NstError):
        K.layernorm_fwd(torch.zeros(2, 5000, device=DEV), torch.ones(5000, device=DEV), torch.zeros(5000, device=DEV), 1e-6)
    with pytest.raises(RuntimeError):
        K.layernorm_fwd(torch.zeros(2, 8), torch.ones(8), torch.zeros(8), 1e-6)  # CPU tensors: no fallback


# ------------------------------------------------------------------------------------------------ gradient clipping
@pytest.mark.parametrize("mode", ["value", "norm"])
def test_grad_clip_matches_oracle(K, mode):
    """nst_grad_clip on a flat buffer of ragged tensors (some below, some above the threshold; one longer than a table
    entry) against oracle.clip_gradients (gradaccum_keras_model.py:228-233), with the 1/world pre-scale folded in."""
    from neurst_amd.runtime import ParamStore
    g = torch.Generator().manual_seed(4)
    shapes = {"a/kernel": (37, 11), "a/bias": (11,), "b/kernel": (130, 70), "c/gamma": (5,), "d/kernel": (3, 3, 1, 16), "e/big": (9000,)}
    store = ParamStore()
    for n, shp in shapes.items():
        store.add(n, shp, torch.zeros(shp))
    store.finalize(DEV, torch.float32)
    grads = {n: torch.randn(shp, generator=g) * (3.0 if "kernel" in n else 0.02) for n, shp in shapes.items()}
    for n, p in store.params.items():
        p.grad.copy_(grads[n].to(DEV))
    pad_before = store.grad.clone()
    pre = 0.25
    table, nentries, seg_first, nseg = store.clip_tables()
    assert nseg == len(shapes) and nentries == sum((int(np.prod(s)) + 4095) // 4096 for s in shapes.values())
    kw = {"clip_value": 0.5} if mode == "value" else {"clip_norm": 2.0}
    K.grad_clip(store.grad, table, nentries, seg_first, nseg, pre_scale=pre, **kw)
    want = O.clip_gradients({n: v.double() * pre for n, v in grads.items()}, **kw)
    for n, p in store.params.items():
        close(f"grad_clip[{mode}].{n}", p.grad, want[n], torch.float32)
    if mode == "norm":
        norms = {n: float(p.grad.norm()) for n, p in store.params.items()}
        assert all(v <= 2.0 * (1 + 1e-5) for v in norms.values()) and norms["a/bias"] < 0.1     # small tensors untouched
    # the alignment padding between tensors is never written
    mask = torch.ones_like(store.grad, dtype=torch.bool)
    for p in store.params.values():
        mask[p.offset:p.offset + p.numel] = False
    assert torch.equal(store.grad[mask], pad_before[mask])
    with pytest.raises(RuntimeError):
        K.grad_clip(store.grad, table, nentries, seg_first, nseg, clip_value=1.0, clip_norm=1.0)


# ------------------------------------------------------------------------------------------------ loss scale, degenerate masks
def test_loss_scale_update_and_scaled_adam(K):
    """nst_loss_scale_update + nst_adam_update_dev (RevisedDynamicLossScale, revised_dynamic_loss_scale.py:48-107)."""
    n = 100003
    g = rnd(n, seed=1).to(DEV) * 512.0                     # gradients carrying a scale of 512
    state = torch.tensor([512.0, 0.0, 1.0, 512.0], device=DEV)
    counter = torch.zeros(4, dtype=torch.int32, device=DEV)
    p = rnd(n, seed=2).to(DEV)
    p0 = p.clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    K.loss_scale_update(g, state, 2, 2.0, counter)
    assert state.tolist() == [512.0, 1.0, 1.0, 512.0] and int(counter[0]) == 0
    K.adam_update(p, m, v, g, None, 1e-3, 0.9, 0.98, 1e-9, 1.0, loss_scale_state=state)
    pr, mr, vr = O.keras_adam_step(p0.cpu().double(), (g / 512.0).cpu().double(), torch.zeros(n, dtype=torch.float64),
                                   torch.zeros(n, dtype=torch.float64), 1, 1e-3 * math.sqrt(1 - 0.98) / (1 - 0.9))
    # keras_adam_step applies the bias correction itself from lr: compare through the raw formula instead
    gi = (g / 512.0).cpu().double()
    want = p0.cpu().double() - 1e-3 * (0.1 * gi) / ((0.02 * gi * gi).sqrt() + 1e-9)
    close("loss_scale.adam", p, want, torch.float32)
    K.loss_scale_update(g, state, 2, 2.0, counter)
    assert state.tolist() == [1024.0, 0.0, 1.0, 512.0]     # second good step: the scale doubles, the gradients carried 512
    g[77] = float("nan")
    p1 = p.clone()
    K.loss_scale_update(g, state, 2, 2.0, counter)
    assert state.tolist() == [512.0, 0.0, 0.0, 1024.0] and int(counter[0]) == 0
    K.adam_update(p, m, v, g, None, 1e-3, 0.9, 0.98, 1e-9, 1.0, loss_scale_state=state)
    assert torch.equal(p, p1)                              # skipped
    state[0] = 1.5
    g[77] = float("inf")
    K.loss_scale_update(g, state, 2, 2.0, counter)
    assert state[0].item() == 1.0                          # floor


def test_attention_fully_padded_rows_match_the_fp32_reference_semantics(K):
    """All keys of a batch element padded (source length 0): the reference adds the FINITE bias -1e9 in fp32
    (neurst/utils/compat.py:24, multi_head_attention.py:147-160), the logits all round to -1e9 and the softmax is uniform.
    An fp64 oracle cannot show this (it keeps the logits apart), so the forward of that element is checked against the same
    math in float32 torch; the partially padded element next to it is checked forward AND backward.  (The backward of the
    fully padded element is a documented limitation: the saved log-sum-exp is one fp32 number and cannot hold log(Tk) next
    to -1e9, so the recomputed probabilities of such a row are not the uniform ones; it costs one more VALU operation per
    probability in three VALU-bound kernels to carry the row's bias offset separately, for a case -- an utterance without a
    single frame -- that the data pipeline filters out, neurst/tasks/speech2text.py:236-260.)"""
    B, H, T, dh = 2, 2, 40, 64
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, T, H * dh, generator=g) for _ in range(3))
    do = torch.randn(B, T, H * dh, generator=g)
    do[0] = 0.0                                           # no gradient enters through the degenerate element
    bias = torch.zeros(B, T)
    bias[0] = -1.0e9                                      # every key of batch element 0 is padding
    bias[1, 30:] = -1.0e9
    qd, kd, vd = (t.to(DEV) for t in (q, k, v))
    out, lse, _ = K.attention_fwd(qd, kd, vd, H, dh, key_bias=bias.to(DEV))
    dq, dk, dv = (torch.empty_like(qd) for _ in range(3))
    K.attention_bwd(qd, kd, vd, out, do.to(DEV), lse, dq, dk, dv, H, dh, key_bias=bias.to(DEV))
    q32, k32, v32 = (t.clone().requires_grad_(True) for t in (q, k, v))
    q4 = (q32 * dh ** -0.5).view(B, T, H, dh)
    logits = torch.einsum("bqhd,bkhd->bhqk", q4, k32.view(B, T, H, dh)) + bias[:, None, None, :]   # float32, like TF
    w = torch.softmax(logits, -1)
    ref = torch.einsum("bhqk,bkhd->bqhd", w, v32.view(B, T, H, dh)).reshape(B, T, H * dh)
    ref.backward(do)
    assert float((w[0] - 1.0 / T).abs().max()) < 1e-6     # the reference's fp32 softmax IS uniform there
    close("attn_fully_padded.out", out, ref.detach(), torch.float32)
    close("attn_partially_padded.dv", dv[1], v32.grad[1], torch.float32)
    close("attn_partially_padded.dq", dq[1], q32.grad[1], torch.float32, scale=5)
    close("attn_partially_padded.dk", dk[1], k32.grad[1], torch.float32, scale=5)
    assert float(dv[0].float().abs().max()) == 0.0 and float(dq[0].float().abs().max()) == 0.0
