"""Parity of the fused feed-forward pair (nst_ffn_fwd / nst_ffn_bwd, neurst_amd/csrc/nst_ffn.hip) through the C ABI against
float64 math of the reference's TransformerFFN + PrePostProcessingWrapper (neurst/layers/common_layers.py:145-160, 73-85) on
the same seeded inputs, bf16 tolerance 1e-2 relative to the magnitude of the reference tensor; dropout masks must be the
library's Philox masks BIT FOR BIT (oracle/philox.py restates the generator)."""
import math

import numpy as np
import pytest
import torch

from oracle import philox

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def rel(got, ref):
    got, ref = got.detach().float().cpu().double(), ref.double()
    assert got.shape == ref.shape
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)


def _operands(M, F, seed):
    d = 256
    x, res = rnd(M, d, seed=seed), rnd(M, d, seed=seed + 1)
    w1, w2 = rnd(d, F, seed=seed + 2, scale=d ** -0.5), rnd(F, d, seed=seed + 3, scale=F ** -0.5)
    g = torch.Generator().manual_seed(seed + 4)
    b1, b2 = torch.randn(F, generator=g) * 0.1, torch.randn(d, generator=g) * 0.1
    return x, res, w1, w2, b1, b2


# M: full 128-row tiles, ragged tails, fewer rows than one tile, one row, and sizes on both sides of the "fills the chip" mark
@pytest.mark.parametrize("M,F", [(256, 256), (192, 128), (200, 512), (1, 128), (77, 2048), (9600, 2048), (20480 + 40, 256)])
def test_ffn_forward_without_dropout(M, F):
    from neurst_amd import kernels as K
    x, res, w1, w2, b1, b2 = _operands(M, F, seed=M + F)
    y, h = K.ffn_fwd(x.to(DEV), w1.t().contiguous().to(DEV), b1.to(DEV), w2.t().contiguous().to(DEV), b2.to(DEV),
                     residual=res.to(DEV))
    h_ref = (x.double() @ w1.double() + b1.double()).clamp_min(0)
    assert rel(h, h_ref) <= 1e-2
    # the second product consumes the bf16-rounded hidden tile: compare against that
    y_ref = h.float().cpu().double() @ w2.double() + b2.double() + res.double()
    assert rel(y, y_ref) <= 1e-2
    # without a residual / biases
    y2, _ = K.ffn_fwd(x.to(DEV), w1.t().contiguous().to(DEV), None, w2.t().contiguous().to(DEV), None)
    h2 = (x.double() @ w1.double()).clamp_min(0).to(BF).double()
    assert rel(y2, h2 @ w2.double()) <= 1e-2


@pytest.mark.parametrize("M,F", [(256, 256), (200, 512), (9600, 2048)])
def test_ffn_forward_dropout_masks_are_the_library_masks(M, F):
    from neurst_amd import kernels as K
    x, res, w1, w2, b1, b2 = _operands(M, F, seed=7 * M + F)
    p1, p2, seed, s1, s2 = 0.25, 0.1, 987654321, 11, 12
    b1 = b1.abs() + 12.0       # every pre-activation positive: a zero in the hidden tile is a dropped unit
    y, h = K.ffn_fwd(x.to(DEV), w1.t().contiguous().to(DEV), b1.to(DEV), w2.t().contiguous().to(DEV), b2.to(DEV),
                     residual=res.to(DEV), hidden_p=p1, hidden_seed=seed, hidden_site=s1, out_p=p2, out_seed=seed, out_site=s2)
    hc = h.float().cpu().double()
    keep1 = torch.from_numpy(philox.keep_multiplier(seed, s1, M * F, p1)).reshape(M, F)
    pre = x.double() @ w1.double() + b1.double()
    assert float(pre.min()) > 0
    assert torch.equal(hc != 0, keep1 != 0), "hidden dropout mask differs from the Philox restatement"
    assert rel(h, pre * keep1) <= 1e-2
    keep2 = torch.from_numpy(philox.keep_multiplier(seed, s2, M * 256, p2)).reshape(M, 256)
    y_ref = (hc @ w2.double() + b2.double()) * keep2 + res.double()
    assert rel(y, y_ref) <= 1e-2
    # same masks as the two-GEMM path (nst_gemm epilogues) draws for the same (seed, site)
    h_g = K.gemm(x.to(DEV), w1.to(DEV), M, F, 256, bias=b1.to(DEV), relu=True, dropout_p=p1, seed=seed, stream_id=s1)
    assert torch.equal(h_g != 0, h != 0)
    y_g = K.gemm(h_g, w2.to(DEV), M, 256, F, bias=b2.to(DEV), dropout_p=p2, seed=seed, stream_id=s2, residual=res.to(DEV))
    assert rel(y, y_g.float().cpu()) <= 2e-2


@pytest.mark.parametrize("M,F", [(256, 256), (192, 128), (200, 512), (1, 128), (63, 384), (9600, 2048), (20480 + 40, 256)])
@pytest.mark.parametrize("with_residual", [False, True])
def test_ffn_backward(M, F, with_residual):
    from neurst_amd import kernels as K
    x, res, w1, w2, b1, b2 = _operands(M, F, seed=3 * M + F)
    dy = rnd(M, 256, seed=5)
    p = 0.25
    keep = torch.from_numpy(philox.keep_multiplier(5, 3, M * F, p)).reshape(M, F)
    h = ((x.double() @ w1.double() + b1.double()).clamp_min(0) * keep).to(BF)       # what the forward saved
    dx, dh = K.ffn_bwd(dy.to(DEV), h.to(DEV), w2.to(DEV), w1.to(DEV), hidden_p=p, residual=res.to(DEV) if with_residual else None)
    gate = torch.where(h > 0, K.dropout_inv_keep(p), 0.0).double()
    dh_ref = (dy.double() @ w2.double().t()) * gate
    assert rel(dh, dh_ref) <= 1e-2
    assert torch.equal(dh.float().cpu() != 0, (dh_ref.to(BF) != 0)) or rel(dh, dh_ref) <= 1e-2
    dx_ref = dh.float().cpu().double() @ w1.double().t() + (res.double() if with_residual else 0.0)
    assert rel(dx, dx_ref) <= 1e-2
    # agrees with the two-GEMM path
    dh_g = K.gemm(dy.to(DEV), w2.to(DEV), M, F, 256, trans_b=True, gate_src=h.to(DEV), gate_scale=K.dropout_inv_keep(p))
    assert rel(dh, dh_g.float().cpu()) <= 1e-2


def test_ffn_pair_at_the_benchmark_shape():
    """The shape and the kernel variants bench.py times: 28 800 rows x 2048 hidden units, 128-row workgroups on full tiles
    (ffn_pair8_kernel<fwd, dropout on both sites, full> and <bwd, full>).  Forward without dropout and the backward
    against float64; with dropout the masks must equal the stream-GEMM path's masks for the same (seed, site) -- that path
    is pinned on the Philox restatement at the smaller sizes above -- and the values its outputs."""
    from neurst_amd import kernels as K
    M, F = 28800, 2048
    x, res, w1, w2, b1, b2 = _operands(M, F, seed=99)
    xd, rd = x.to(DEV), res.to(DEV)
    w1d, w2d, b1d, b2d = w1.to(DEV), w2.to(DEV), b1.to(DEV), b2.to(DEV)
    w1t, w2t = w1.t().contiguous().to(DEV), w2.t().contiguous().to(DEV)
    y, h = K.ffn_fwd(xd, w1t, b1d, w2t, b2d, residual=rd)
    h_ref = (x.double() @ w1.double() + b1.double()).clamp_min(0)
    assert rel(h, h_ref) <= 1e-2
    y_ref = h.float().cpu().double() @ w2.double() + b2.double() + res.double()
    assert rel(y, y_ref) <= 1e-2
    del h_ref, y_ref
    p1, p2, seed, s1, s2 = 0.1, 0.1, 24681357, 21, 22
    yd, hd = K.ffn_fwd(xd, w1t, b1d, w2t, b2d, residual=rd, hidden_p=p1, hidden_seed=seed, hidden_site=s1, out_p=p2,
                       out_seed=seed, out_site=s2)
    h_g = K.gemm(xd, w1d, M, F, 256, bias=b1d, relu=True, dropout_p=p1, seed=seed, stream_id=s1)
    # a unit is zero where it was dropped or where its pre-activation is negative -- on both paths alike, up to the sign of
    # pre-activations within bf16 rounding of zero
    differ = int(((h_g != 0) != (hd != 0)).sum())
    assert differ <= 1e-5 * M * F, f"{differ} hidden units differ in their zero pattern"
    assert rel(hd, h_g.float().cpu()) <= 1e-2
    y_g = K.gemm(hd, w2d, M, 256, F, bias=b2d, dropout_p=p2, seed=seed, stream_id=s2, residual=rd)
    assert rel(yd, y_g.float().cpu()) <= 1e-2
    keep = abs(float((hd != 0).float().mean()) / max(float((h != 0).float().mean()), 1e-9) - (1.0 - p1))
    assert keep < 5e-3, "hidden keep rate"
    # backward on the saved (dropped) hidden tile
    dy = rnd(M, 256, seed=5).to(DEV)
    dx, dh = K.ffn_bwd(dy, hd, w2d, w1d, hidden_p=p1, residual=rd)
    gate = torch.where(hd.cpu() > 0, K.dropout_inv_keep(p1), 0.0).double()
    dh_ref = (dy.cpu().double() @ w2.double().t()) * gate
    assert rel(dh, dh_ref) <= 1e-2
    dx_ref = dh.float().cpu().double() @ w1.double().t() + res.double()
    assert rel(dx, dx_ref) <= 1e-2


@pytest.mark.parametrize("M,F,p1", [(28800, 2048, 0.1), (28800, 2048, 0.0), (20480 + 40, 256, 0.25), (25600, 128, 0.1)])
def test_ffn_gate_bits_give_the_backward_of_the_saved_activation(M, F, p1):
    """The eight-wave forward can hand the backward `hidden > 0` as one bit per element (NstFfnDesc.gate_bits) instead of
    the [rows, F] activation.  Both gates are the same predicate on the same bf16 values, so the two backwards must agree
    BIT FOR BIT (full and ragged workgroups, with and without hidden dropout), and the forward's outputs do not change."""
    from neurst_amd import kernels as K
    x, res, w1, w2, b1, b2 = _operands(M, F, seed=M + 3 * F)
    xd, rd = x.to(DEV), res.to(DEV)
    w1d, w2d, b1d, b2d = w1.to(DEV), w2.to(DEV), b1.to(DEV), b2.to(DEV)
    w1t, w2t = w1.t().contiguous().to(DEV), w2.t().contiguous().to(DEV)
    kw = dict(residual=rd, hidden_p=p1, hidden_seed=1234, hidden_site=5, out_p=0.1, out_seed=1234, out_site=6)
    y0, h0 = K.ffn_fwd(xd, w1t, b1d, w2t, b2d, **kw)
    y1, h1, bits = K.ffn_fwd(xd, w1t, b1d, w2t, b2d, save_gate_bits=True, **kw)
    assert bits is not None and bits.numel() == M * (F // 32) * 4, "this shape runs the eight-wave kernels"
    assert torch.equal(y0, y1) and torch.equal(h0, h1)
    zero_frac = float((h1 == 0).float().mean())
    assert 0.3 < zero_frac < 0.8                      # the gate is neither all ones nor all zeros
    dy = rnd(M, 256, seed=5).to(DEV)
    dx_h, dh_h = K.ffn_bwd(dy, h1, w2d, w1d, hidden_p=p1, residual=rd)
    dx_b, dh_b = K.ffn_bwd(dy, h1, w2d, w1d, hidden_p=p1, residual=rd, gate_bits=bits)
    assert torch.equal(dh_h, dh_b), "gate bits and the saved activation gate differently"
    assert torch.equal(dx_h, dx_b)
    assert torch.equal(dh_b != 0, (h1 > 0) & (dh_b != 0)) and int((dh_b != 0).sum()) > 0.9 * int((h1 > 0).sum())


def test_ffn_gate_bits_are_refused_where_no_kernel_writes_them():
    import ctypes as C
    from neurst_amd import kernels as K
    from neurst_amd._lib import lib
    M, F = 512, 256
    x, res, w1, w2, b1, b2 = _operands(M, F, seed=1)
    y, h, bits = K.ffn_fwd(x.to(DEV), w1.t().contiguous().to(DEV), b1.to(DEV), w2.t().contiguous().to(DEV), b2.to(DEV),
                           save_gate_bits=True)
    assert bits is None
    desc = K._ffn_desc(M, 256, F, 0.0, 0, 0)
    assert lib.nst_ffn_gate_bits_bytes(C.byref(desc)) == 0
    buf = torch.zeros(M * F // 8, dtype=torch.uint8, device=DEV)
    desc.gate_bits, desc.gate_bits_bytes = buf.data_ptr(), buf.numel()
    yy, hh = torch.empty_like(y), torch.empty_like(h)
    rc = lib.nst_ffn_fwd(C.byref(desc), x.to(DEV).data_ptr(), w1.t().contiguous().to(DEV).data_ptr(), None,
                         w2.t().contiguous().to(DEV).data_ptr(), None, None, hh.data_ptr(), yy.data_ptr(), None)
    assert rc != 0 and b"gate_bits" in lib.nst_last_error_string()


def test_transposed_weight_copies_follow_the_optimizer():
    """The forward reads transposed bf16 copies of the two FFN kernels: they must equal the bf16 shadow transposed after
    construction, after a state-dict load and after every optimizer step."""
    from neurst_amd.layers.common_layers import TransformerFFN
    from neurst_amd.optimizers.adam import Adam
    from neurst_amd.runtime import Runtime
    rt = Runtime(device=DEV, dtype="bfloat16", seed=1)
    ffn = TransformerFFN(rt, "ffn", 256, 384, 0.1, torch.Generator().manual_seed(0))
    assert ffn.fused
    rt.store.finalize(rt.device, rt.dtype)

    def check():
        assert torch.equal(ffn._w1t.t, ffn.dense1.kernel.compute.t()) and torch.equal(ffn._w2t.t, ffn.dense2.kernel.compute.t())
    check()
    opt = Adam(rt.store, learning_rate=1e-2)
    rt.store.grad.normal_()
    opt.apply_gradients()
    check()
    rt.store.load_state_dict({n: torch.randn(p.shape) for n, p in rt.store.params.items()})
    check()


def test_transpose_launch_vector_and_elementwise_paths():
    """nst_transpose_bf16 on a table of matrices: shapes with both sides a multiple of 8 take the 16-byte path (incl. tiles cut by
    the matrix edge), anything else the element-wise one; bit-exact."""
    import numpy as np
    from neurst_amd import kernels as K
    shapes = [(256, 2048), (2048, 256), (72, 200), (8, 8), (64, 64), (65, 130), (7, 24), (136, 9)]
    g = torch.Generator().manual_seed(3)
    srcs = [torch.randn(r, c, generator=g).to(torch.bfloat16).to(DEV) for r, c in shapes]
    dsts = [torch.full((c, r), 7.0, dtype=torch.bfloat16, device=DEV) for r, c in shapes]
    rows, tiles = [], 0
    for (r, c), a, b in zip(shapes, srcs, dsts):
        tiles_c = (c + 63) // 64
        rows.append((a.data_ptr(), b.data_ptr(), r, c, tiles_c, tiles))
        tiles += ((r + 63) // 64) * tiles_c
    arr = np.array(rows, dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("tiles_c", "<i4"),
                                         ("tile0", "<i4")]))
    table = torch.from_numpy(arr.view(np.uint8).copy()).to(DEV)
    K.transpose_bf16(table, len(shapes), tiles)
    torch.cuda.synchronize()
    for (r, c), a, b in zip(shapes, srcs, dsts):
        assert torch.equal(b, a.t()), (r, c)


def test_ffn_layer_fused_equals_two_gemm_path(monkeypatch):
    """TransformerFFN inside the pre-norm wrapper: fused launch vs the two-GEMM composition, forward and backward, same
    dropout masks (both dropouts on)."""
    from neurst_amd.layers.common_layers import PrePostProcessingWrapper, TransformerFFN
    from neurst_amd.runtime import Runtime
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setattr("neurst_amd.layers.common_layers._FFN_FUSED", fused == "1")
        monkeypatch.setattr("neurst_amd.layers.common_layers._FFN_FUSED_MIN_ROWS", 1)
        monkeypatch.setattr("neurst_amd.layers.common_layers._FFN_FUSED_BWD", True)
        rt = Runtime(device=DEV, dtype="bfloat16", seed=3)
        w = PrePostProcessingWrapper(rt, "w", TransformerFFN(rt, "w/ffn", 256, 512, 0.1, torch.Generator().manual_seed(0)), 256,
                                     0.1, 1e-6)
        assert w.layer.fused == (fused == "1")
        rt.store.finalize(rt.device, rt.dtype)
        x = rnd(300, 256, seed=1).to(DEV)
        y = w.forward(x, True)
        rt.store.begin_backward()
        dx = w.backward(rnd(300, 256, seed=2).to(DEV))
        rt.join_wgrad_stream()
        torch.cuda.synchronize()
        outs.append((y.float().cpu(), dx.float().cpu(), rt.store.grad.clone().cpu()))
    (y1, dx1, g1), (y0, dx0, g0) = outs
    assert rel(y1, y0) <= 1e-2 and rel(dx1, dx0) <= 2e-2
    assert float((g1 - g0).norm() / g0.norm()) <= 2e-2


# ------------------------------------------------------------------------------------------------ the pair with the wrapper's row stages
@pytest.mark.parametrize("M,F,p1,p2", [(28800, 2048, 0.1, 0.1), (28800, 2048, 0.0, 0.0), (20480 + 40, 256, 0.25, 0.1), (25600, 128, 0.1, 0.0),
                                       (9600, 2048, 0.1, 0.1), (9600, 2048, 0.0, 0.0), (4000 + 19, 512, 0.2, 0.1), (1024, 256, 0.1, 0.1)])
def test_ffn_add_layernorm_fwd_equals_the_pair_of_launches(M, F, p1, p2):
    """nst_ffn_add_layernorm_fwd against nst_ffn_fwd (no residual, the same masks, gate bits) + nst_add_layernorm_fwd on the
    float32 stream: the same hidden activation and gate bits bit for bit (one main loop), the same sum / LayerNorm output /
    statistics up to the f32 accumulation order of the row reductions."""
    from neurst_amd import kernels as K
    # >= 20 480 rows: one launch; 1 024 .. 20 479: the hidden dimension split over workgroups + a row launch (the decoder's 9 600)
    assert K.ffn_ln_supported(M, 256, F) == (1 if M >= 20480 else 2) and not K.ffn_ln_supported(512, 256, 2048)
    x, _, w1, w2, b1, b2 = _operands(M, F, seed=3 * M + F)
    g = torch.Generator().manual_seed(M + 5)
    xres = (torch.randn(M, 256, generator=g) * 2.0).to(DEV)
    gamma, beta = (1.0 + 0.2 * torch.randn(256, generator=g)).to(DEV), (0.1 * torch.randn(256, generator=g)).to(DEV)
    d = lambda t: t.to(DEV)
    w1t, w2t = d(w1.t().contiguous()), d(w2.t().contiguous())
    kw = dict(hidden_p=p1, hidden_seed=77, hidden_site=3, out_p=p2, out_seed=77, out_site=4)
    y, xs, mean, rstd, h, bits = K.ffn_add_layernorm_fwd(d(x), w1t, d(b1), w2t, d(b2), xres, gamma, beta, 1e-6, **kw)
    delta, h2, bits2 = K.ffn_fwd(d(x), w1t, d(b1), w2t, d(b2), save_gate_bits=True, **kw)
    y2, xs2, mean2, rstd2 = K.add_layernorm_fwd(xres, delta, gamma, beta, 1e-6)
    assert torch.equal(h, h2), "hidden activation differs from the plain pair kernel"
    if M >= 20480:
        assert torch.equal(bits, bits2), "gate bits differ from the plain pair kernel"
        assert torch.equal(xs, xs2), "the float32 stream differs (same product, same rounding of delta)"
    else:   # split form: the partial sums of the slices are added in another order (and nst_ffn_fwd has no bit path here)
        assert bits2 is None and float((xs - xs2).abs().max()) <= 4e-3 * float(xs2.abs().max())
        assert float((xs - xs2).norm() / xs2.norm()) <= 3e-4
    assert rel(y, y2.float().cpu()) <= 1e-2 and float((y.float() - y2.float()).norm() / y2.float().norm()) <= 2e-3
    if M >= 20480:
        assert torch.allclose(mean, mean2, rtol=1e-5, atol=1e-6) and torch.allclose(rstd, rstd2, rtol=1e-5, atol=0)
    else:   # (another summation order of the product: a delta entry may land one bf16 step away)
        assert torch.allclose(mean, mean2, rtol=1e-3, atol=2e-4) and torch.allclose(rstd, rstd2, rtol=2e-3, atol=0)
    y3, none, _, _, _, _ = K.ffn_add_layernorm_fwd(d(x), w1t, d(b1), w2t, d(b2), xres, gamma, beta, 1e-6, want_sum=False, **kw)
    assert none is None and torch.equal(y3, y)


@pytest.mark.parametrize("M,F,p1,drop", [(28800, 2048, 0.1, True), (28800, 2048, 0.0, False), (20480 + 40, 256, 0.25, True), (25600, 128, 0.1, False),
                                         (9600, 2048, 0.1, True), (9600, 2048, 0.0, False), (4000 + 19, 512, 0.2, True), (1024, 256, 0.1, False)])
def test_ffn_layernorm_bwd_equals_the_pair_of_launches(M, F, p1, drop):
    """nst_ffn_layernorm_bwd against nst_ffn_bwd + nst_layernorm_bwd_mixed: d(hidden) bit for bit, dx / dz / dgamma / dbeta up to
    the accumulation order; also through the deferred finalize of the batch."""
    from neurst_amd import kernels as K
    x, _, w1, w2, b1, b2 = _operands(M, F, seed=5 * M + F)
    g = torch.Generator().manual_seed(M + 9)
    d = lambda t: t.to(DEV)
    w1t, w2t = d(w1.t().contiguous()), d(w2.t().contiguous())
    # (the gate bits of the split form come from the LN forward entry only: nst_ffn_fwd has no bit path below 20 480 rows)
    xr0 = torch.zeros(M, 256, device=DEV)
    one, zero = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    _, _, _, _, h, bits = K.ffn_add_layernorm_fwd(d(x), w1t, d(b1), w2t, d(b2), xr0, one, zero, 1e-6, hidden_p=p1, hidden_seed=5,
                                                  hidden_site=2, want_sum=False)
    dy = d(rnd(M, 256, seed=M + 2))
    dres = d(rnd(M, 256, seed=M + 3))
    xln = (torch.randn(M, 256, generator=g) * 2.0 + 0.3).to(DEV)
    gamma = (1.0 + 0.2 * torch.randn(256, generator=g)).to(DEV)
    mean = xln.mean(1)
    rstd = ((xln - mean[:, None]).pow(2).mean(1) + 1e-6).rsqrt()
    emit = (0.2, 31337, 6) if drop else None
    dg, db = torch.full((256,), 7.0, device=DEV), torch.full((256,), 7.0, device=DEV)
    dx, dz, dh = K.ffn_layernorm_bwd(dy, h, d(w2), d(w1), xln, gamma, mean, rstd, dg, db, hidden_p=p1, gate_bits=bits, dres=dres,
                                     emit_dropout=emit)
    gmid, dh2 = K.ffn_bwd(dy, h, d(w2), d(w1), hidden_p=p1, gate_bits=bits if M >= 20480 else None)
    dg2, db2 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    out2 = K.layernorm_bwd(gmid, xln, gamma, mean, rstd, dg2, db2, dres=dres, emit_dropout=emit)
    dx2, dz2 = out2 if drop else (out2, None)
    assert torch.equal(dh, dh2), "d(hidden) differs from the plain pair kernel"
    assert rel(dx, dx2.float().cpu()) <= 1e-2 and float((dx.float() - dx2.float()).norm() / dx2.float().norm()) <= 2e-3
    assert rel(dg, dg2.cpu()) <= 1e-3 and rel(db, db2.cpu()) <= 1e-3
    if drop:
        assert rel(dz, dz2.float().cpu()) <= 1e-2 and float(((dz == 0) != (dz2 == 0)).float().mean()) < 1e-3
    else:
        assert dz is None
    batch = K.SplitkBatch(DEV)
    dg3, db3 = torch.full((256,), 3.0, device=DEV), torch.full((256,), -2.0, device=DEV)
    dx3, dz3, dh3 = K.ffn_layernorm_bwd(dy, h, d(w2), d(w1), xln, gamma, mean, rstd, dg3, db3, hidden_p=p1, gate_bits=bits, dres=dres,
                                        emit_dropout=emit, accumulate=True, batch=batch)
    assert batch.ln_n == 1 and torch.equal(dx3, dx) and torch.equal(dh3, dh)
    batch.flush()
    torch.cuda.synchronize()
    assert torch.allclose(dg3, dg + 3.0, rtol=1e-5, atol=1e-4) and torch.allclose(db3, db - 2.0, rtol=1e-5, atol=1e-4)
