"""TEST INFRASTRUCTURE -- CPU restatement of the CONTRACT of every tensor-level entry point of `neurst_amd.kernels`
(= the C ABI of include/neurst_hip.h), written in plain torch on float64.

Purpose: the host side of the path (hand-scheduled forward / backward of the layers, gradient bookkeeping in the flat
buffer, the data-parallel reducer hooks, the train step, decoding caches) is ordinary Python that can be checked
without a GPU -- but `neurst_amd.kernels` refuses CPU tensors by design (no fallback in the product).  The `-m "not
gpu"` tests therefore install THIS module over `neurst_amd.kernels` with `install(monkeypatch)` and run the same
layer code against `oracle/neurst_oracle.py`.  Nothing in `neurst_amd/` imports this file; `bench.py` does not either.

Each function mirrors the signature of its namesake in neurst_amd/kernels.py and the semantics documented in
include/neurst_hip.h (the reference lines it stands for are cited there).  Dropout: every mask outside attention is the
kernels' own (oracle/philox.py restates the generator: Philox4x32-7 keyed by (seed, site), 16-bit field of the element's
linear index), so forward and backward regenerate the same mask exactly like the device does.  The dropout on attention
probabilities uses the same generator over the linear index of [B, H, Tq, Tk] -- NOT the device's bit layout (the HIP
forward writes its keep bits to a buffer the backward reads; here the mask tensor itself plays that buffer).
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import philox

FLOAT_MIN = -1.0e9
F64 = torch.float64


def _keep(p, seed, site, shape):
    """float64 keep multipliers (0 or 1/keep) of a tensor of `shape` whose elements are numbered in row-major order; the
    library's seed offset (dropout_seed_offset_set / _add) is added to the seed like the kernels do."""
    n = int(np.prod(shape))
    return torch.from_numpy(philox.keep_multiplier(seed + int(_SEED_OFFSET[0]), site, n, p)).reshape(tuple(shape))


# ---------------------------------------------------------------------------------------------------- LayerNorm
def _ln_stats(x2, eps):
    mean = x2.mean(dim=1)
    var = ((x2 - mean[:, None]) ** 2).mean(dim=1)
    return mean, (var + eps).rsqrt()


def layernorm_fwd(x, gamma, beta, eps, relu=False):
    assert x.is_contiguous()
    d = x.shape[-1]
    x2 = x.reshape(-1, d).to(F64)
    mean, rstd = _ln_stats(x2, eps)
    y = (x2 - mean[:, None]) * rstd[:, None] * gamma.to(F64) + beta.to(F64)
    if relu:
        y = y.clamp_min(0)
    return y.to(x.dtype).reshape(x.shape), mean.float(), rstd.float()


def add_layernorm_supported(d, dtype):
    return dtype == torch.bfloat16 and d % 8 == 0 and d <= 1024


def add_layernorm_fwd(x, delta, gamma, beta, eps, want_sum=True):
    """nst_add_layernorm_fwd: the float32 residual stream.  x f32 or bf16, delta bf16 -> y = LN(x + delta) in bf16, the sum in
    f32 (exactly what the next call adds to and the backward normalises again), mean, rstd."""
    assert x.is_contiguous() and delta is not None and delta.is_contiguous() and delta.dtype == torch.bfloat16
    assert x.dtype in (torch.float32, torch.bfloat16) and delta.shape == x.shape
    d = x.shape[-1]
    assert add_layernorm_supported(d, delta.dtype)
    xs = (x.reshape(-1, d).to(F64) + delta.reshape(-1, d).to(F64)).float()      # the device adds in f32 and stores f32
    mean, rstd = _ln_stats(xs.to(F64), eps)
    y = (xs.to(F64) - mean[:, None]) * rstd[:, None] * gamma.to(F64) + beta.to(F64)
    return y.to(delta.dtype).reshape(x.shape), (xs.reshape(x.shape) if want_sum else None), mean.float(), rstd.float()


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, accumulate=False, dres=None, y=None, emit_dropout=None, batch=None,
                  regate_beta=None):
    # (regate_beta: nst_layernorm_relu_bwd_regate -- the ReLU gate recomputed from x, the saved statistics, gamma and beta)
    # (x.dtype != dy.dtype: the f32 sum saved by add_layernorm_fwd under bf16 gradients, nst_layernorm_bwd_mixed)
    assert dy.is_contiguous() and x.is_contiguous() and (dy.dtype == x.dtype or (x.dtype == torch.float32 and y is None))
    assert dres is None or (dres.is_contiguous() and dres.dtype == dy.dtype and y is None)
    d = x.shape[-1]
    g = dy.reshape(-1, d).to(F64)
    if y is not None:  # backward of relu(LN(x)): gate by the saved output
        g = g * (y.reshape(-1, d) > 0).to(F64)
    xh = (x.reshape(-1, d).to(F64) - mean.to(F64)[:, None]) * rstd.to(F64)[:, None]
    if regate_beta is not None:
        assert y is None and dres is None and emit_dropout is None
        g = g * ((xh * gamma.to(F64) + regate_beta.to(F64)) > 0).to(F64)
    dg, db = (g * xh).sum(0).float(), g.sum(0).float()
    if accumulate:
        dgamma.add_(dg)
        dbeta.add_(db)
    else:
        dgamma.copy_(dg)
        dbeta.copy_(db)
    gh = g * gamma.to(F64)
    dx = (gh - gh.mean(1, keepdim=True) - xh * (gh * xh).mean(1, keepdim=True)) * rstd.to(F64)[:, None]
    if dres is not None:
        dx = dx + dres.reshape(-1, d).to(F64)
    dx = dx.to(dy.dtype).reshape(x.shape)
    if emit_dropout is not None:
        p, seed, site = emit_dropout
        return dx, (dx.to(F64) * _keep(p, seed, site, dx.shape)).to(dx.dtype)
    return dx


# ---------------------------------------------------------------------------------------------------- GEMM
def gemm(A, B, M, N, K, trans_a=False, trans_b=False, out=None, out_dtype=None, alpha=1.0, bias=None, relu=False,
         dropout_p=0.0, seed=0, stream_id=0, residual=None, gate_src=None, gate_scale=1.0, posenc=None,
         posenc_period=0, emb_scale=1.0, accumulate=False, split_k=1, colsum_out=None, colsum_accumulate=False, batch=None,
         rowdot=None):
    assert rowdot is None, "the emulation tier never asks for the fused row dots (rowdot_supported is False there)"
    assert A.dim() == 2 and B.dim() == 2 and A.dtype == B.dtype and A.stride(1) == 1 and B.stride(1) == 1
    assert out is None or (out.dim() == 2 and out.stride(1) == 1)
    assert residual is None or residual.stride(1) == 1
    assert gate_src is None or (gate_src.stride(1) == 1 and gate_src.dtype == (out.dtype if out is not None else out_dtype or A.dtype))
    assert posenc is None or (posenc.dtype == torch.float32 and posenc.is_contiguous() and posenc.shape[1] == N)
    assert colsum_out is None or (colsum_out.dtype == torch.float32 and colsum_out.numel() == N and colsum_out.is_contiguous())
    a = (A.t() if trans_a else A).to(F64)
    b = (B.t() if trans_b else B).to(F64)
    assert a.shape == (M, K) and b.shape == (K, N), f"gemm operand shapes {tuple(a.shape)} x {tuple(b.shape)} vs {M,N,K}"
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or A.dtype)
    assert tuple(out.shape) == (M, N)
    if split_k > 1:
        assert out.dtype == torch.float32 and bias is None and not relu and residual is None and gate_src is None \
            and posenc is None, "split_k needs the plain f32 epilogue"
    v = alpha * (a @ b)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
        v = v + bias.to(F64)
    if relu:
        v = v.clamp_min(0)
    if dropout_p > 0:
        assert split_k <= 1
        v = v * _keep(dropout_p, seed, stream_id, (M, N))     # element index row * N + col
    if residual is not None:
        assert residual.dtype == out.dtype
        v = v + residual.to(F64)
    if gate_src is not None:
        v = v * torch.where(gate_src > 0, gate_scale, 0.0).to(F64)
    if posenc is not None:
        period = posenc_period or posenc.shape[0]
        rows = torch.arange(M) % period
        v = v * emb_scale + posenc.to(F64)[rows]
    if accumulate:
        out.add_(v.to(out.dtype))
    else:
        out.copy_(v.to(out.dtype))
    if colsum_out is not None:
        assert not trans_b, "colsum rides on B [K,N]"
        cs = b.sum(0).float()
        if colsum_accumulate:
            colsum_out.add_(cs)
        else:
            colsum_out.copy_(cs)
    return out


_SEED_OFFSET = [0]      # the scalar in use: a one-element list, rebound by dropout_seed_offset_bind
_SEED_OWN = _SEED_OFFSET


# ---------------------------------------------------------------------------------------------------- whole-row products
def rowgemm_supported(A, n, k):
    return A.dtype == torch.bfloat16 and A.dim() == 2 and A.stride(1) == 1 and A.shape[0] > 0 and n == 256 and k % 64 == 0 and k >= 64


def gemm_add_layernorm_fwd(A, W, x, gamma, beta, eps, bias=None, trans_b=False, dropout_p=0.0, seed=0, stream_id=0, want_sum=True):
    """nst_gemm_add_layernorm_fwd = nst_gemm (bias, dropout, bf16 output) followed by nst_add_layernorm_fwd."""
    rows, k = A.shape
    n = W.shape[0] if trans_b else W.shape[1]
    assert rowgemm_supported(A, n, k) and x.dtype == torch.float32 and x.is_contiguous() and x.numel() == rows * n
    delta = gemm(A, W, rows, n, k, trans_b=trans_b, bias=bias, dropout_p=dropout_p, seed=seed, stream_id=stream_id)
    return add_layernorm_fwd(x.reshape(rows, n), delta, gamma, beta, eps, want_sum=want_sum)


def gemm_layernorm_bwd(A, W, x, gamma, mean, rstd, dgamma, dbeta, accumulate=False, dres=None, emit_dropout=None, batch=None,
                       trans_b=True):
    """nst_gemm_layernorm_bwd = nst_gemm (bf16 output) followed by nst_layernorm_bwd_mixed."""
    rows, k = A.shape
    n = W.shape[0] if trans_b else W.shape[1]
    assert rowgemm_supported(A, n, k) and x.dtype == torch.float32 and x.is_contiguous() and x.numel() == rows * n
    g = gemm(A, W, rows, n, k, trans_b=trans_b)
    return layernorm_bwd(g, x.reshape(rows, n), gamma, mean, rstd, dgamma, dbeta, accumulate=accumulate,
                         dres=None if dres is None else dres.reshape(rows, n), emit_dropout=emit_dropout, batch=batch)


def gemm_rowdot256(A, W, rowdot, trans_b=True):
    rows, k = A.shape
    n = W.shape[0] if trans_b else W.shape[1]
    assert rowgemm_supported(A, n, k)
    out = gemm(A, W, rows, n, k, trans_b=trans_b)
    if rowdot is not None:
        src, dst, T = rowdot
        prod = (out.to(F64) * src.reshape(rows, n).to(F64)).reshape(rows // T, T, n // 64, 64).sum(-1)   # [B, T, H]
        dst.copy_(prod.permute(0, 2, 1).reshape(dst.shape).float())
    return out


def dropout_seed_offset_bind(scalar):
    """On the CPU tier the bound scalar is the runtime's 1-element int64 tensor (or None for the library's own)."""
    global _SEED_OFFSET
    _SEED_OFFSET = scalar if scalar is not None else _SEED_OWN


def dropout_seed_offset_set(value):
    _SEED_OFFSET[0] = int(value)


def dropout_seed_offset_add(delta=1):
    _SEED_OFFSET[0] += int(delta)


def splitk_reduce_multi(jobs, n):
    """The emulated gemm completes a split-K product at once: nothing is ever deferred on the CPU tier."""
    assert n == 0


def ffn_supported(d_model, filter_size, dtype):
    return dtype == torch.bfloat16 and d_model == 256 and filter_size >= 128 and filter_size % 128 == 0 and filter_size <= 8192


def ffn_fwd(x, w1t, b1, w2t, b2, residual=None, hidden_p=0.0, hidden_seed=0, hidden_site=0, out_p=0.0, out_seed=0,
            out_site=0, save_gate_bits=False):
    rows, d = x.shape
    f = w1t.shape[0]
    assert x.is_contiguous() and w1t.is_contiguous() and w2t.is_contiguous() and w1t.shape == (f, d) and w2t.shape == (d, f)
    assert x.dtype == torch.bfloat16 and w1t.dtype == x.dtype and w2t.dtype == x.dtype and d == 256 and f % 128 == 0
    assert residual is None or (residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype)
    h = (x.to(F64) @ w1t.to(F64).t() + (b1.to(F64) if b1 is not None else 0.0)).clamp_min(0)
    if hidden_p > 0:
        h = h * _keep(hidden_p, hidden_seed, hidden_site, (rows, f))
    h = h.to(x.dtype)                                   # the hidden tile feeds the second product as bf16
    y = h.to(F64) @ w2t.to(F64).t() + (b2.to(F64) if b2 is not None else 0.0)
    if out_p > 0:
        y = y * _keep(out_p, out_seed, out_site, (rows, d))
    if residual is not None:
        y = y + residual.to(F64)
    return (y.to(x.dtype), h, None) if save_gate_bits else (y.to(x.dtype), h)   # (no bit path: the activation is the gate)


def ffn_bwd(dy, hidden, w2, w1, hidden_p=0.0, residual=None, gate_bits=None):
    rows, d = dy.shape
    f = w2.shape[0]
    assert dy.is_contiguous() and hidden.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()
    assert w2.shape == (f, d) and w1.shape == (d, f) and hidden.shape == (rows, f) and dy.dtype == torch.bfloat16
    from neurst_amd.kernels import dropout_inv_keep
    gate = torch.where(hidden > 0, dropout_inv_keep(hidden_p) if hidden_p > 0 else 1.0, 0.0).to(F64)
    dh = ((dy.to(F64) @ w2.to(F64).t()) * gate).to(dy.dtype)
    dx = dh.to(F64) @ w1.to(F64).t()
    if residual is not None:
        dx = dx + residual.to(F64)
    return dx.to(dy.dtype), dh


def ffn_ln_supported(rows, d, f):
    return (1 if rows >= 128 * 160 else 2 if rows >= 1024 else 0) if (d == 256 and f % 128 == 0) else 0


def ffn_add_layernorm_fwd(x, w1t, b1, w2t, b2, x_res, gamma, beta, eps, hidden_p=0.0, hidden_seed=0, hidden_site=0, out_p=0.0,
                          out_seed=0, out_site=0, want_sum=True):
    """nst_ffn_add_layernorm_fwd = nst_ffn_fwd (no residual, gate bits) followed by nst_add_layernorm_fwd."""
    delta, hidden, bits = ffn_fwd(x, w1t, b1, w2t, b2, residual=None, hidden_p=hidden_p, hidden_seed=hidden_seed,
                                  hidden_site=hidden_site, out_p=out_p, out_seed=out_seed, out_site=out_site, save_gate_bits=True)
    y, xs, mean, rstd = add_layernorm_fwd(x_res.reshape(x.shape), delta, gamma, beta, eps, want_sum=want_sum)
    return y, xs, mean, rstd, hidden, torch.zeros(4, dtype=torch.uint8)     # (a stand-in for the device's opaque gate bits)


def ffn_layernorm_bwd(dy, hidden, w2, w1, x_ln, gamma, mean, rstd, dgamma, dbeta, hidden_p=0.0, gate_bits=None, accumulate=False,
                      dres=None, emit_dropout=None, batch=None):
    """nst_ffn_layernorm_bwd = nst_ffn_bwd followed by nst_layernorm_bwd_mixed."""
    assert gate_bits is not None
    g, dhidden = ffn_bwd(dy, hidden, w2, w1, hidden_p=hidden_p, residual=None, gate_bits=None)   # (the activation is the gate here)
    out = layernorm_bwd(g, x_ln.reshape(dy.shape), gamma, mean, rstd, dgamma, dbeta, accumulate=accumulate,
                        dres=None if dres is None else dres.reshape(dy.shape), emit_dropout=emit_dropout, batch=batch)
    dx, dz = out if emit_dropout is not None else (out, None)
    return dx, dz, dhidden


def transpose_bf16(table, njobs, total_tiles):
    """`table` on the CPU tier is the Python list of (src, dst) tensor pairs ParamStore keeps next to the device table."""
    for src, dst in table:
        dst.copy_(src.t())


def pack2d(table, njobs, total_blocks):
    """`table` on the CPU tier is a Python list of (src, dst) tensor pairs (dst a strided block of the packed buffer);
    ParamStore itself refreshes its packed copies without this entry point on CPU tensors."""
    for src, dst in table:
        dst.copy_(src)


def colsum(x, out, accumulate=False):
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32
    s = x.to(F64).sum(0).float()
    if accumulate:
        out.add_(s)
    else:
        out.copy_(s)
    return out


def grad_clip(grad, table, nentries, seg_first, nseg, pre_scale=1.0, clip_value=None, clip_norm=None):
    if bool(clip_value) == bool(clip_norm):   # the library reports NST_ERR_INVALID_ARG, the binding raises
        raise RuntimeError("grad_clip: exactly one of clip_value / clip_norm must be positive")
    rows = table.cpu().numpy().view(np.dtype([("off", "<i8"), ("n", "<i4"), ("seg", "<i4")]))
    first = seg_first.cpu().tolist()
    assert len(rows) == nentries and len(first) == nseg + 1
    grad.mul_(pre_scale)
    for s in range(nseg):
        ent = rows[first[s]:first[s + 1]]
        if clip_value:
            for e in ent:
                grad[int(e["off"]):int(e["off"]) + int(e["n"])].clamp_(-clip_value, clip_value)
        else:
            sq = sum(float((grad[int(e["off"]):int(e["off"]) + int(e["n"])].double() ** 2).sum()) for e in ent)
            f = clip_norm / max(sq ** 0.5, clip_norm)
            for e in ent:
                grad[int(e["off"]):int(e["off"]) + int(e["n"])].mul_(f)


# ---------------------------------------------------------------------------------------------------- attention
def _attn_probs(q, k, H, dh, key_bias, causal, causal_offset):
    """q [B,Tq,H*dh] / k [B,Tk,H*dh] float64 -> probabilities [B,H,Tq,Tk] and the log-sum-exp of the biased logits.
    Padding bias is the reference's finite FLOAT_MIN; causal / wait-k masked keys get exact zeros."""
    B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
    qh = q.reshape(B, Tq, H, dh).permute(0, 2, 1, 3) * (float(dh) ** -0.5)
    kh = k.reshape(B, Tk, H, dh).permute(0, 2, 1, 3)
    logits = qh @ kh.transpose(-1, -2)
    if key_bias is not None:
        logits = logits + key_bias.to(F64)[:, None, None, :]
    if causal:
        i, j = torch.arange(Tq)[:, None], torch.arange(Tk)[None, :]
        logits = logits.masked_fill((j > i + int(causal_offset))[None, None], float("-inf"))
    lse = torch.logsumexp(logits, dim=-1)
    return torch.exp(logits - lse[..., None]), lse


def _check_attention_views(q, k, v, out, H, dh, key_bias, causal_offset):
    """The stride / layout requirements the real binding enforces before it fills the descriptor."""
    from neurst_amd import kernels as K
    K._attn_desc(q, k, v, out, H, dh, False, 0.0, 0, 0, causal_offset)
    assert q.shape[-1] == H * dh and k.shape[-1] == H * dh and v.shape[-1] == H * dh and k.shape[:2] == v.shape[:2]
    assert causal_offset >= 0
    if key_bias is not None:
        assert key_bias.dtype == torch.float32 and key_bias.is_contiguous() and tuple(key_bias.shape) == (q.shape[0], k.shape[1])


def attention_fwd(q, k, v, H, dh, key_bias=None, causal=False, dropout_p=0.0, seed=0, stream_id=0, causal_offset=0):
    B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
    _check_attention_views(q, k, v, torch.empty(B, Tq, H * dh, dtype=q.dtype), H, dh, key_bias, causal_offset)
    P, lse = _attn_probs(q.to(F64), k.to(F64), H, dh, key_bias, causal, causal_offset)
    vh = v.to(F64).reshape(B, Tk, H, dh).permute(0, 2, 1, 3)
    mask = _keep(dropout_p, seed, stream_id, P.shape) if dropout_p > 0 else None
    out = ((P if mask is None else P * mask) @ vh).permute(0, 2, 1, 3).reshape(B, Tq, H * dh)
    return out.to(q.dtype).contiguous(), lse.float().contiguous(), mask


def rowdot_supported(x, n):
    return False


def attention_bwd(q, k, v, out, dout, lse, dq, dk, dv, H, dh, key_bias=None, causal=False, dropout_p=0.0, seed=0,
                  stream_id=0, drop_mask=None, causal_offset=0, delta=None):
    assert delta is None
    assert (dropout_p > 0) == (drop_mask is not None), "attention_bwd: dropout needs the mask written by attention_fwd"
    B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
    assert dout.is_contiguous() and out.is_contiguous()
    assert dq.stride(1) == q.stride(1) and dk.stride(1) == k.stride(1) and dv.stride(1) == v.stride(1)
    _check_attention_views(q, k, v, out, H, dh, key_bias, causal_offset)
    qd, kd, vd = q.to(F64), k.to(F64), v.to(F64)
    P, _ = _attn_probs(qd, kd, H, dh, key_bias, causal, causal_offset)
    scale = float(dh) ** -0.5
    do = dout.to(F64).reshape(B, Tq, H, dh).permute(0, 2, 1, 3)
    vh = vd.reshape(B, Tk, H, dh).permute(0, 2, 1, 3)
    qh = qd.reshape(B, Tq, H, dh).permute(0, 2, 1, 3)
    kh = kd.reshape(B, Tk, H, dh).permute(0, 2, 1, 3)
    Pd = P if drop_mask is None else P * drop_mask          # the probabilities the forward multiplied with V
    dV = Pd.transpose(-1, -2) @ do
    dP = do @ vh.transpose(-1, -2)
    if drop_mask is not None:
        dP = dP * drop_mask
    dS = P * (dP - (dP * P).sum(-1, keepdim=True))
    dQ = (dS @ kh) * scale
    dK = (dS.transpose(-1, -2) @ qh) * scale
    dq.copy_(dQ.permute(0, 2, 1, 3).reshape(B, Tq, H * dh).to(dq.dtype))
    dk.copy_(dK.permute(0, 2, 1, 3).reshape(B, Tk, H * dh).to(dk.dtype))
    dv.copy_(dV.permute(0, 2, 1, 3).reshape(B, Tk, H * dh).to(dv.dtype))


# ---------------------------------------------------------------------------------------------------- conv front end
def _conv_s2(x_nchw, w_hwio, bias):
    """pad 1 on both spatial sides, 3x3 stride-2 VALID (audio_modalities.py:96-101)."""
    w = w_hwio.to(F64).permute(3, 2, 0, 1)
    return F.conv2d(F.pad(x_nchw, (1, 1, 1, 1)), w, None if bias is None else bias.to(F64), stride=2)


def _conv1_pre(src, w1, b1):
    return _conv_s2(src.to(F64)[:, None], w1, b1).permute(0, 2, 3, 1)  # [B,T1,F1,C]


def conv1_ln_relu_fwd(src, w1, b1, gamma, beta, layer_norm, eps, out_dtype):
    assert src.is_contiguous() and src.dtype == torch.float32 and w1.is_contiguous()
    z = _conv1_pre(src, w1, b1)
    mean = rstd = None
    if layer_norm:
        C = z.shape[-1]
        m, r = _ln_stats(z.reshape(-1, C), eps)
        z = ((z.reshape(-1, C) - m[:, None]) * r[:, None] * gamma.to(F64) + beta.to(F64)).reshape(z.shape)
        mean, rstd = m.float(), r.float()
    return z.clamp_min(0).to(out_dtype).contiguous(), mean, rstd


def conv1_ln_relu_bwd(src, w1, b1, gamma, beta, mean, rstd, dout, dw1, db1, dgamma, dbeta, layer_norm, eps,
                      accumulate=False):
    assert dout.is_contiguous()
    w = w1.to(F64).clone().requires_grad_(True)
    b = b1.to(F64).clone().requires_grad_(True)
    leaves = [w, b]
    z = _conv_s2(src.to(F64)[:, None], w, b).permute(0, 2, 3, 1)
    if layer_norm:
        g = gamma.to(F64).clone().requires_grad_(True)
        be = beta.to(F64).clone().requires_grad_(True)
        leaves += [g, be]
        C = z.shape[-1]
        z2 = z.reshape(-1, C)
        m = z2.mean(1, keepdim=True)
        var = ((z2 - m) ** 2).mean(1, keepdim=True)
        z = ((z2 - m) * (var + eps).rsqrt() * g + be).reshape(z.shape)
    y = z.clamp_min(0)
    grads = torch.autograd.grad(y, leaves, dout.to(F64))
    outs = [dw1, db1] + ([dgamma, dbeta] if layer_norm else [])
    for o, gr in zip(outs, grads):
        gr = gr.float().reshape(o.shape)
        if accumulate:
            o.add_(gr)
        else:
            o.copy_(gr)


def conv2_fwd(x, w2, b2, relu=False):
    assert x.is_contiguous() and w2.is_contiguous() and w2.dtype == x.dtype
    y = _conv_s2(x.to(F64).permute(0, 3, 1, 2), w2, b2).permute(0, 2, 3, 1)
    if relu:
        y = y.clamp_min(0)
    return y.to(x.dtype).contiguous()


def conv2_dgrad(dy, w2, T1, F1):
    assert dy.is_contiguous() and w2.dtype == dy.dtype
    B, T2, F2, C = dy.shape
    x = torch.zeros(B, C, T1, F1, dtype=F64, requires_grad=True)
    y = _conv_s2(x, w2, None)
    (dx,) = torch.autograd.grad(y, x, dy.to(F64).permute(0, 3, 1, 2))
    return dx.permute(0, 2, 3, 1).to(dy.dtype).contiguous()


def conv2_wgrad(x, dy, dw2, db2=None, accumulate=False):
    assert x.is_contiguous() and dy.is_contiguous() and dw2.dtype == torch.float32
    w = torch.zeros(dw2.shape, dtype=F64, requires_grad=True)
    y = _conv_s2(x.to(F64).permute(0, 3, 1, 2), w, None)
    (g,) = torch.autograd.grad(y, w, dy.to(F64).permute(0, 3, 1, 2))
    bsum = dy.to(F64).sum((0, 1, 2)).float()
    if accumulate:
        dw2.add_(g.float())
        if db2 is not None:
            db2.add_(bsum)
    else:
        dw2.copy_(g.float())
        if db2 is not None:
            db2.copy_(bsum)


# ---------------------------------------------------------------------------------------------------- embedding / elementwise
def embedding_fwd(table, ids, posenc, L, emb_scale, dropout_p=0.0, seed=0, stream_id=0):
    assert ids.dtype == torch.int64 and (posenc is None or (posenc.is_contiguous() and posenc.shape[0] >= min(L, ids.numel())))
    d = table.shape[1]
    flat = ids.reshape(-1)
    out = table.to(F64)[flat] * emb_scale
    if posenc is not None:
        out = out + posenc.to(F64)[torch.arange(flat.numel()) % L]
    if dropout_p > 0:
        out = out * _keep(dropout_p, seed, stream_id, out.shape)
    return out.to(table.dtype).reshape(*ids.shape, d)


def embedding_bwd(dout, ids, dtable, emb_scale, dropout_p=0.0, seed=0, stream_id=0):
    assert dout.is_contiguous() and dtable.dtype == torch.float32
    d = dtable.shape[1]
    acc = torch.zeros(dtable.shape, dtype=F64)
    g = dout.reshape(-1, d).to(F64) * emb_scale
    if dropout_p > 0:
        g = g * _keep(dropout_p, seed, stream_id, g.shape)
    acc.index_add_(0, ids.reshape(-1), g)
    dtable.add_(acc.float())  # always accumulates (neurst_hip.h)


def scale_posenc_dropout_fwd(x, posenc, period, scale, dropout_p=0.0, seed=0, stream_id=0):
    assert x.is_contiguous()
    d = x.shape[-1]
    y = x.reshape(-1, d).to(F64) * scale
    if posenc is not None:
        y = y + posenc.to(F64)[torch.arange(y.shape[0]) % period]
    if dropout_p > 0:
        y = y * _keep(dropout_p, seed, stream_id, y.shape)
    return y.to(x.dtype).reshape(x.shape)


def scale_dropout_bwd(dy, scale, dropout_p=0.0, seed=0, stream_id=0):
    assert dy.is_contiguous()
    g = dy.to(F64) * scale
    if dropout_p > 0:
        g = g * _keep(dropout_p, seed, stream_id, g.shape)
    return g.to(dy.dtype)


# ---------------------------------------------------------------------------------------------------- criterion / optimizer
def _xent_consts(V, ls):
    conf = 1.0 - ls
    low = ls / (V - 1) if V > 1 else 0.0
    norm = -(conf * np.log(conf) + (V - 1) * low * np.log(low + 1e-20)) if ls != 0 else 0.0
    return conf, low, norm


def ls_xent_fwd(logits, labels, weights, label_smoothing):
    assert logits.dim() == 2 and logits.stride(1) == 1
    rows, V = logits.shape
    conf, low, norm = _xent_consts(V, label_smoothing)
    lg = logits.to(F64)
    lse = torch.logsumexp(lg, dim=1)
    logp = lg - lse[:, None]
    tgt = logp.gather(1, labels.reshape(-1, 1))[:, 0]
    xent = -(low * (logp.sum(1) - tgt) + conf * tgt) - norm
    return (xent * weights.to(F64)).float(), lse.float()


def ls_xent_bwd(logits, labels, weights, lse, label_smoothing, gscale, out=None, gscale_dev=None):
    rows, V = logits.shape
    conf, low, _ = _xent_consts(V, label_smoothing)
    p = torch.exp(logits.to(F64) - lse.to(F64)[:, None])
    soft = torch.full((rows, V), low, dtype=F64)
    soft.scatter_(1, labels.reshape(-1, 1), conf)
    s = gscale * (float(gscale_dev.reshape(-1)[0]) if gscale_dev is not None else 1.0)
    g = ((p - soft) * weights.to(F64)[:, None] * s).to(logits.dtype)
    if out is not None:
        out.copy_(g)
        return out
    return g


def loss_scale_update(grad, state, growth_steps, multiplier, counter):
    finite = bool(torch.isfinite(grad).all())
    state[3] = state[0]
    state[2] = 1.0 if finite else 0.0
    if finite:
        if float(state[1]) + 1.0 >= growth_steps:
            grown = state[0] * multiplier
            if bool(torch.isfinite(grown)):
                state[0] = grown
            state[1] = 0.0
        else:
            state[1] += 1.0
    else:
        state[0] = max(float(state[0]) / multiplier, 1.0)
        state[1] = 0.0


def adam_update(p, m, v, g, shadow, lr_t, beta1, beta2, eps, grad_scale=1.0, loss_scale_state=None):
    if loss_scale_state is not None:
        if float(loss_scale_state[2]) == 0.0:
            return
        grad_scale = grad_scale / float(loss_scale_state[3])
    gs = g * grad_scale
    m.mul_(beta1).add_(gs, alpha=1.0 - beta1)
    v.mul_(beta2).add_(gs * gs, alpha=1.0 - beta2)
    p.sub_(lr_t * m / (v.sqrt() + eps))
    if shadow is not None:
        shadow.copy_(p.to(torch.bfloat16))


def cast_f32_to_bf16(src, dst):
    dst.copy_(src.to(torch.bfloat16))


_NAMES = ["layernorm_fwd", "layernorm_bwd", "add_layernorm_fwd", "add_layernorm_supported", "gemm", "colsum", "grad_clip", "attention_fwd", "attention_bwd",
          "conv1_ln_relu_fwd", "conv1_ln_relu_bwd", "conv2_fwd", "conv2_dgrad", "conv2_wgrad", "embedding_fwd",
          "embedding_bwd", "scale_posenc_dropout_fwd", "scale_dropout_bwd", "ls_xent_fwd", "ls_xent_bwd", "adam_update",
          "cast_f32_to_bf16", "ffn_supported", "ffn_fwd", "ffn_bwd", "transpose_bf16", "pack2d",
          "dropout_seed_offset_bind", "dropout_seed_offset_set", "dropout_seed_offset_add", "loss_scale_update", "splitk_reduce_multi",
          "gemm_wgrad_group", "seq_mask", "xent_reduce", "rowgemm_supported", "gemm_add_layernorm_fwd", "gemm_layernorm_bwd",
          "gemm_rowdot256", "ffn_ln_supported", "ffn_add_layernorm_fwd", "ffn_layernorm_bwd"]


def seq_mask(lengths, max_len, on_token, on_padding, halvings=0, stride=2):
    ln = lengths.to(torch.int64)
    for _ in range(int(halvings)):
        ln = (ln + stride - 1) // stride
    inside = torch.arange(int(max_len))[None, :] < ln[:, None]
    return torch.where(inside, torch.tensor(float(on_token)), torch.tensor(float(on_padding))).float()


def xent_reduce(xent, weights):
    B, L = weights.shape
    nll, tok = xent.reshape(B, L).to(F64).sum(1), weights.to(F64).sum(1)
    return nll.float(), tok.float(), (nll.sum() / tok.sum()).float().reshape(1), (1.0 / tok.sum()).float().reshape(1)


def gemm_wgrad_group(items, table=None):
    """nst_gemm_wgrad_group: every product exactly as the plain weight-gradient gemm (no split)."""
    for x, dz, out, acc, cs, cs_acc in items:
        gemm(x, dz, x.shape[1], dz.shape[1], x.shape[0], trans_a=True, out=out, accumulate=acc, colsum_out=cs,
             colsum_accumulate=cs_acc)


def install(monkeypatch):
    """Replaces the tensor-level entry points of neurst_amd.kernels for the duration of one test (pytest monkeypatch:
    undone automatically).  Returns the list of names it replaced."""
    from neurst_amd import kernels as K
    for n in _NAMES:
        assert hasattr(K, n), f"neurst_amd.kernels has no entry point {n}"
        monkeypatch.setattr(K, n, globals()[n])
    return list(_NAMES)
