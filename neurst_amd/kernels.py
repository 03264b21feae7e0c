"""Thin tensor-level wrappers over the C ABI of libneurst_hip.so.

PyTorch is used only as the owner of device memory and streams: every function
passes raw device pointers and the current HIP stream to the library.  All
inputs must live on a ROCm device; nothing here falls back to torch math.
"""
import ctypes as C
import os

import torch

from neurst_amd._lib import (NST_BF16, NST_F32, NstAttnDesc, NstFfnDesc, NstGemmDesc, NstLnFinalizeJob, NstRowGemmDesc, NstSplitkJob,
                             check, lib)

FLOAT_MIN = -1.0e9  # neurst/utils/compat.py:24

_WS = {}


def dropout_inv_keep(p):
    """Keep-multiplier the kernels apply for rate p outside attention: the mask compares 16-bit Philox fields against
    round(p*65536), so the multiplier is 65536/(65536 - round(p*65536)) (nst_common.h: nst_dropout_params16)."""
    t = min(max(int(p * 65536.0 + 0.5), 0), 65535)
    return 65536.0 / (65536.0 - t)


def _workspace(nbytes, device):
    """Persistent scratch for two-stage reductions (split-K slabs, LayerNorm / bias-gradient partial sums); grown
    on demand.  One buffer per (device, stream): launches on one stream are ordered, so consecutive users never
    overlap, and concurrent streams never share a buffer."""
    key = (device, _stream())   # (callers run on `device`: the runtime pins one device per process)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _dt(t):
    if t.dtype == torch.float32:
        return NST_F32
    if t.dtype == torch.bfloat16:
        return NST_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("neurst_amd kernels need ROCm device tensors (no CPU fallback)")
    return t.data_ptr()


# raw handle of the current stream of the current device: two C calls instead of torch.cuda.current_stream()'s Python objects
# (an eager step issues ~500 launches; the Stream object cost ~5 us of host time each)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class _KernelProbe(object):
    """HIP-event timing of kernel entry points on the stream they are launched on (bench.py's roofline pass).
    Inactive unless start(names) was called; events are only read back in stop().  Every probed call reports its
    algorithmic work (FLOP) so that a family of launches can be priced as sum(work) / sum(duration)."""

    def __init__(self):
        self.names, self.events = (), []

    def start(self, names):
        self.names, self.events = ((names,) if isinstance(names, str) else tuple(names)), []

    def stop(self):
        """-> {name: [(milliseconds, work, bytes), ...]}"""
        torch.cuda.synchronize()
        out = {}
        for name, a, b, work, nbytes in self.events:
            out.setdefault(name, []).append((a.elapsed_time(b), work, nbytes))
        self.names, self.events = (), []
        return out

    def begin(self, name):
        if name not in self.names:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return (name, ev)

    def end(self, tok, work=0.0, nbytes=0.0):
        """work: algorithmic FLOP of the launch; nbytes: its algorithmic HBM bytes (operands read once + outputs written once)."""
        if tok is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record(torch.cuda.current_stream())
            self.events.append((tok[0], tok[1], e2, float(work), float(nbytes)))


PROBE = _KernelProbe()


# ------------------------------------------------------------------------------------------------ dropout seed offset
def dropout_seed_offset_bind(scalar):
    """scalar: 1-element int64 device tensor that the following launches use as the dropout seed offset (None: the library's)."""
    check(lib.nst_dropout_seed_offset_bind(_p(scalar)), "dropout_seed_offset_bind", launches=False)


def dropout_seed_offset_set(value):
    """The library's device scalar that every dropout kernel adds to its seed when it runs (0 by default)."""
    check(lib.nst_dropout_seed_offset_set(int(value), _stream()), "dropout_seed_offset_set")


def dropout_seed_offset_add(delta=1):
    check(lib.nst_dropout_seed_offset_add(int(delta), _stream()), "dropout_seed_offset_add")


# ------------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, eps, relu=False):
    assert x.is_contiguous()
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    fn = lib.nst_layernorm_relu_fwd if relu else lib.nst_layernorm_fwd
    check(fn(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, d, float(eps), _dt(x), _stream()),
          "layernorm_fwd")
    return y, mean, rstd


def add_layernorm_fwd(x, delta, gamma, beta, eps, want_sum=True):
    """The fp32 residual stream of the pre-norm bf16 path (nst_add_layernorm_fwd): x [.., d] f32 or bf16, delta bf16 or None.
    -> (y bf16 = LayerNorm(x + delta), x_new f32 = x + delta (None unless want_sum; with delta None it is the f32 copy of x),
        mean, rstd)."""
    assert x.is_contiguous() and (delta is None or (delta.is_contiguous() and delta.dtype == torch.bfloat16 and delta.shape == x.shape))
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    xs = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_sum else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.nst_add_layernorm_fwd(_p(x), _dt(x), _p(delta), _p(xs), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, d,
                                    float(eps), NST_BF16, _stream()), "add_layernorm_fwd")
    return y, xs, mean, rstd


def add_layernorm_supported(d, dtype):
    """Shapes the fp32 residual stream exists for (the wide LayerNorm kernels): bf16 models with d % 8 == 0, d <= 1024."""
    return dtype == torch.bfloat16 and d % 8 == 0 and d <= 1024


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, accumulate=False, dres=None, y=None, emit_dropout=None, batch=None,
                  regate_beta=None):
    """dx = LN'(dy) (+ dres).  With y given: backward of relu(LN(x)); with regate_beta (the LayerNorm's beta) instead of y the
    ReLU gate is recomputed from x and the saved statistics (nst_layernorm_relu_bwd_regate: the saved activation is not read).
    emit_dropout=(p, seed, site): also returns dz = dropout_backward(dx) under that mask -> (dx, dz).
    batch (SplitkBatch): the dgamma / dbeta reduction is left to the batch's next flush (one launch for all pending
    LayerNorms, on the stream that flushes -- the weight-gradient stream), see nst_layernorm_bwd_deferred."""
    assert dy.is_contiguous() and x.is_contiguous()
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(dy)
    slot = batch.ln_slot(d) if batch is not None else None
    if regate_beta is not None:
        assert y is None and dres is None and emit_dropout is None and x.dtype == dy.dtype
        if slot is not None:
            ws_ptr, ws_bytes, job = slot
        else:
            ws = _workspace(64 << 20, x.device)
            ws_ptr, ws_bytes, job = ws.data_ptr(), ws.numel(), None
        check(lib.nst_layernorm_relu_bwd_regate(_p(dy), _p(x), _p(gamma), _p(regate_beta), _p(mean), _p(rstd), _p(dx), _p(dgamma),
                                                _p(dbeta), rows, d, _dt(x), int(accumulate), ws_ptr, ws_bytes, job, _stream()),
              "layernorm_relu_bwd_regate")
        return dx
    if x.dtype != dy.dtype:       # saved input of the fp32 residual stream: f32 x, bf16 gradients
        assert x.dtype == torch.float32 and dy.dtype == torch.bfloat16 and y is None
        assert dres is None or (dres.is_contiguous() and dres.dtype == dy.dtype)
        p, seed, site = emit_dropout if emit_dropout is not None else (0.0, 0, 0)
        dz = torch.empty_like(dy) if emit_dropout is not None else None
        if slot is not None:
            ws_ptr, ws_bytes, job = slot
        else:
            ws = _workspace(64 << 20, x.device)
            ws_ptr, ws_bytes, job = ws.data_ptr(), ws.numel(), None
        check(lib.nst_layernorm_bwd_mixed(_p(dy), _p(x), NST_F32, _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dz), p, seed,
                                          site, _p(dgamma), _p(dbeta), rows, d, _dt(dy), int(accumulate), ws_ptr, ws_bytes, job,
                                          _stream()), "layernorm_bwd_mixed")
        return (dx, dz) if emit_dropout is not None else dx
    if slot is not None:
        assert dres is None or (dres.is_contiguous() and dres.dtype == x.dtype)
        assert not (y is not None and (dres is not None or emit_dropout is not None))
        p, seed, site = emit_dropout if emit_dropout is not None else (0.0, 0, 0)
        dz = torch.empty_like(x) if emit_dropout is not None else None
        ws_ptr, ws_bytes, job = slot
        check(lib.nst_layernorm_bwd_deferred(_p(dy), _p(x), _p(y), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dz), p,
                                             seed, site, _p(dgamma), _p(dbeta), rows, d, _dt(x), int(accumulate), ws_ptr,
                                             ws_bytes, job, _stream()), "layernorm_bwd_deferred")
        return (dx, dz) if emit_dropout is not None else dx
    ws = _workspace(64 << 20, x.device)
    if emit_dropout is not None:
        assert y is None
        p, seed, site = emit_dropout
        dz = torch.empty_like(x)
        if dres is not None:
            assert dres.is_contiguous() and dres.dtype == x.dtype
        check(lib.nst_layernorm_bwd_dropout(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dz), p, seed,
                                            site, _p(dgamma), _p(dbeta), rows, d, _dt(x), int(accumulate), ws.data_ptr(),
                                            ws.numel(), _stream()), "layernorm_bwd_dropout")
        return dx, dz
    if y is not None:
        assert dres is None
        check(lib.nst_layernorm_relu_bwd(_p(dy), _p(x), _p(y), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma),
                                         _p(dbeta), rows, d, _dt(x), int(accumulate), ws.data_ptr(), ws.numel(),
                                         _stream()), "layernorm_relu_bwd")
    else:
        if dres is not None:
            assert dres.is_contiguous() and dres.dtype == x.dtype
        check(lib.nst_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dgamma),
                                    _p(dbeta), rows, d, _dt(x), int(accumulate), ws.data_ptr(), ws.numel(), _stream()),
              "layernorm_bwd")
    return dx


# ------------------------------------------------------------------------------------------------ GEMM
class SplitkBatch(object):
    """Collects the second stages of up to 8 split-K weight gradients (their slabs live side by side in one scratch buffer)
    and reduces them with ONE launch (nst_splitk_reduce_multi) instead of one 7-20 us launch each."""

    LN_SLOTS, LN_MAX_D = 64, 512     # deferred LayerNorm finalize stages: ring of partial-sum slots (8 MB each)

    def __init__(self, device, nbytes=192 << 20):
        self.jobs = (NstSplitkJob * 8)()
        self.n, self.cursor = 0, 0
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.device = device
        self.ln_jobs = (NstLnFinalizeJob * 16)()
        self.ln_n, self.ln_cursor, self.ln_since_join, self.ln_ws = 0, 0, 0, None

    def ln_slot(self, d):
        """-> (device pointer, bytes, host job pointer) for one deferred LayerNorm finalize, or None (finalize right away).
        The slot is written by the dx kernel on the CURRENT stream and read by the next flush on the flushing stream; a slot
        comes round again LN_SLOTS calls later, so no more than LN_SLOTS are handed out between two joins of the streams
        (joined(), called by Runtime.join_wgrad_stream at the end of every backward)."""
        if d > self.LN_MAX_D or self.ln_n >= 16 or self.ln_since_join >= self.LN_SLOTS or not hasattr(lib, "nst_layernorm_bwd_deferred"):
            return None
        slot_bytes = 2048 * 2 * self.LN_MAX_D * 4   # what nst_layernorm_bwd* asks of a partial-sum workspace
        if self.ln_ws is None:
            self.ln_ws = torch.empty(self.LN_SLOTS * slot_bytes, dtype=torch.uint8, device=self.device)
        ptr = self.ln_ws.data_ptr() + self.ln_cursor * slot_bytes
        self.ln_cursor = (self.ln_cursor + 1) % self.LN_SLOTS
        self.ln_since_join += 1
        job = C.addressof(self.ln_jobs[self.ln_n])
        self.ln_n += 1
        return ptr, slot_bytes, job

    def joined(self):
        self.ln_since_join = 0

    def region(self, nbytes):
        """-> (device pointer, bytes) of a free 256-byte aligned region, or None when the batch must be flushed first."""
        start = (self.cursor + 255) // 256 * 256
        if self.n >= 8 or start + nbytes > self.ws.numel():
            return None
        self.cursor = start + nbytes
        return self.ws.data_ptr() + start, nbytes

    def flush(self):
        if self.n:
            splitk_reduce_multi(self.jobs, self.n)
        self.n, self.cursor = 0, 0
        if self.ln_n:
            check(lib.nst_ln_finalize_multi(C.addressof(self.ln_jobs), self.ln_n, _stream()), "ln_finalize_multi")
        self.ln_n = 0


class WgradGroup(object):
    """Weight-gradient products dW (+)= X^T dZ waiting for ONE nst_gemm_wgrad_group launch: every 256 x 256 output tile of every
    product becomes a workgroup of the same grid, so the gradients of a whole layer stack fill the chip without split-K slabs.
    The group keeps its operands alive until launch()."""

    MIN_OUTPUTS = 128 * 128    # smaller gradients waste most of a 256 x 256 tile: they keep the per-product path

    def __init__(self, device):
        self.device = device
        self.items = []
        self.table = None       # device copy of the product table for launches of more than 56 products

    @staticmethod
    def _al(t, esz):
        return t.data_ptr() % 16 == 0 and (t.stride(0) * esz) % 16 == 0 and t.stride(1) == 1

    def accepts(self, x, dz, out, colsum_out=None):
        """x [rows, M], dz [rows, N] bf16 views, out [M, N] f32 view: what the grouped kernel can take."""
        M, N = x.shape[1], dz.shape[1]
        return (x.dtype == torch.bfloat16 and dz.dtype == torch.bfloat16 and out.dtype == torch.float32
                and x.device.type == torch.device(self.device).type and M % 8 == 0 and N % 8 == 0 and M * N >= self.MIN_OUTPUTS and x.shape[0] == dz.shape[0] and x.shape[0] > 0
                and self._al(x, 2) and self._al(dz, 2) and self._al(out, 4)
                and (colsum_out is None or (colsum_out.dtype == torch.float32 and colsum_out.is_contiguous())))

    def add(self, x, dz, out, accumulate=False, colsum_out=None, colsum_accumulate=False):
        self.items.append((x, dz, out, bool(accumulate), colsum_out, bool(colsum_accumulate)))

    def pending_bytes(self):
        """Activation bytes the queued products keep alive until the launch (x and dz of every product)."""
        return sum(x.numel() * x.element_size() + dz.numel() * dz.element_size() for x, dz, *_ in self.items)

    def __len__(self):
        return len(self.items)

    def take(self):
        """-> (launch, tensors): the queued products leave the group; launch() issues them on the stream current THEN
        (Runtime.run_wgrad defers it while a step is captured), tensors are the operands it reads."""
        items, self.items = self.items, []
        table = None
        if len(items) > 56:      # the product table travels through device memory: one buffer per launch in flight
            if self.table is None:
                self.table = [torch.empty(1024 * 72, dtype=torch.uint8, device=self.device) for _ in range(4)]
                self._table_next = 0
            table = self.table[self._table_next]
            self._table_next = (self._table_next + 1) % len(self.table)
        return (lambda: gemm_wgrad_group(items, table)), [t for it in items for t in it[:2]]

    def launch(self):
        if self.items:
            self.take()[0]()


def gemm_wgrad_group(items, table=None):
    """items: [(x [rows, M], dz [rows, N], out [M, N] f32, accumulate, colsum_out [N] | None, colsum_accumulate)] -> one launch."""
    n = len(items)
    descs = (NstGemmDesc * n)()
    Ap, Bp, Cp = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    flops = 0.0
    for i, (x, dz, out, acc, cs, cs_acc) in enumerate(items):
        d = descs[i]
        d.M, d.N, d.K = x.shape[1], dz.shape[1], x.shape[0]
        d.trans_a, d.trans_b = 1, 0
        d.lda, d.ldb, d.ldc = x.stride(0), dz.stride(0), out.stride(0)
        d.in_dtype, d.out_dtype = NST_BF16, NST_F32
        d.alpha = 1.0
        d.accumulate = int(acc)
        d.split_k = 1
        if cs is not None:
            d.colsum, d.colsum_accumulate = cs.data_ptr(), int(cs_acc)
        Ap[i], Bp[i], Cp[i] = x.data_ptr(), dz.data_ptr(), out.data_ptr()
        flops += 2.0 * d.M * d.N * d.K
    ev = PROBE.begin("gemm_wgrad_group")
    check(lib.nst_gemm_wgrad_group(descs, Ap, Bp, Cp, n, _p(table), table.numel() if table is not None else 0, _stream()),
          "gemm_wgrad_group")
    if ev is not None:
        PROBE.end(ev, flops, sum(x.numel() * 2 + dz.numel() * 2 + 2 * out.numel() * 4 for x, dz, out, *_ in items))


def splitk_reduce_multi(jobs, n):
    check(lib.nst_splitk_reduce_multi(C.addressof(jobs), n, _stream()), "splitk_reduce_multi")


def gemm(A, B, M, N, K, trans_a=False, trans_b=False, out=None, out_dtype=None, alpha=1.0, bias=None, relu=False,
         dropout_p=0.0, seed=0, stream_id=0, residual=None, gate_src=None, gate_scale=1.0, posenc=None,
         posenc_period=0, emb_scale=1.0, accumulate=False, split_k=1, colsum_out=None, colsum_accumulate=False, batch=None,
         rowdot=None):
    """C[M,N] = epilogue(alpha * op(A) @ op(B)); A/B are 2-D views with unit inner stride (see neurst_hip.h).
    colsum_out [N] f32 (+)= column sums of B over the reduction (the bias gradient of a weight-gradient GEMM).
    rowdot=(src [M, N], dst [M/rows, N/64, rows] f32, rows): dst = per-head (64 columns) row sums of C o src, from the same
    epilogue (bf16 stream kernel only; see rowdot_supported)."""
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    assert A.dtype == B.dtype
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or A.dtype, device=A.device)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] == M and out.shape[1] == N
    d = NstGemmDesc()
    d.M, d.N, d.K = M, N, K
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.lda, d.ldb, d.ldc = A.stride(0), B.stride(0), out.stride(0)
    d.in_dtype, d.out_dtype = _dt(A), _dt(out)
    d.alpha = alpha
    d.bias = _p(bias)
    d.relu = int(relu)
    d.dropout_p = dropout_p
    d.seed, d.stream_id = seed, stream_id
    if residual is not None:
        assert residual.dtype == out.dtype and residual.stride(1) == 1
        d.residual, d.ldr = _p(residual), residual.stride(0)
    if gate_src is not None:
        assert gate_src.dtype == out.dtype and gate_src.stride(1) == 1
        d.gate_src, d.ldg = _p(gate_src), gate_src.stride(0)
    d.gate_scale = gate_scale
    if posenc is not None:
        assert posenc.dtype == torch.float32 and posenc.is_contiguous() and posenc.shape[1] == N
        d.posenc, d.posenc_period = _p(posenc), posenc_period or posenc.shape[0]
    d.emb_scale = emb_scale
    d.accumulate = int(accumulate)
    d.split_k = split_k
    if rowdot is not None:
        src, dst, rows = rowdot
        assert src.stride(-1) == 1 and dst.dtype == torch.float32 and dst.is_contiguous() and dst.numel() == M * (N // 64)
        d.rowdot_src, d.ldrs, d.rowdot_dst = src.data_ptr(), src.stride(-2), dst.data_ptr()
        d.rowdot_rows, d.rowdot_heads = rows, N // 64
    deferred = False
    if split_k > 1:
        need = split_k * (M + 1) * N * 4
        reg = batch.region(need) if batch is not None else None
        if batch is not None and reg is None:      # full: reduce what is queued, then there is room
            batch.flush()
            reg = batch.region(need)
        if reg is not None:                        # deferred second stage: one reduce launch per batch
            d.workspace, d.workspace_bytes = reg
            d.reduce_job_out = C.addressof(batch.jobs[batch.n])
            deferred = True
        else:
            ws = _workspace(need, A.device)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    if colsum_out is not None:
        assert colsum_out.dtype == torch.float32 and colsum_out.numel() == N and colsum_out.is_contiguous()
        d.colsum, d.colsum_accumulate = colsum_out.data_ptr(), int(colsum_accumulate)
    ev = PROBE.begin("gemm")
    if deferred:
        batch.jobs[batch.n].slabs = None
    check(lib.nst_gemm(C.byref(d), _p(A), _p(B), _p(out), _stream()), "gemm")
    if deferred and batch.jobs[batch.n].slabs:     # the library took the deferred path (it may decline: odd shapes)
        batch.n += 1
    if ev is not None:
        esz, osz = A.element_size(), out.element_size()
        nbytes = (M * K + K * N) * esz + M * N * osz * (2 if accumulate else 1)
        nbytes += M * N * osz * ((residual is not None) + (gate_src is not None))
        PROBE.end(ev, 2.0 * M * N * K, nbytes)
    return out


# ------------------------------------------------------------------------------------------------ whole-row products (d_model 256)
def rowgemm_supported(A, n, k):
    """Whether the whole-row products exist for A [rows, k] -> [rows, n] (nst_rowgemm_supported: bf16, n = 256, k % 64 == 0)."""
    return (A.dtype == torch.bfloat16 and A.dim() == 2 and A.stride(1) == 1 and A.stride(0) % 8 == 0 and A.data_ptr() % 16 == 0
            and A.shape[0] > 0 and bool(lib.nst_rowgemm_supported(int(n), int(k), NST_BF16)))


def _rowgemm_desc(A, W, trans_b, dropout_p=0.0, seed=0, stream_id=0, eps=0.0):
    rows, k = A.shape
    assert W.dim() == 2 and W.stride(1) == 1 and W.dtype == torch.bfloat16 and A.dtype == torch.bfloat16 and A.stride(1) == 1
    n = W.shape[0] if trans_b else W.shape[1]
    assert (W.shape[1] if trans_b else W.shape[0]) == k, f"rowgemm operand shapes {tuple(A.shape)} x {tuple(W.shape)} trans_b={trans_b}"
    d = NstRowGemmDesc()
    d.rows, d.n, d.k, d.trans_b, d.dtype = rows, n, k, int(bool(trans_b)), NST_BF16
    d.lda, d.ldb = A.stride(0), W.stride(0)
    d.dropout_p, d.eps, d.seed, d.stream_id = float(dropout_p), float(eps), int(seed), int(stream_id)
    return d, rows, n, k


def gemm_add_layernorm_fwd(A, W, x, gamma, beta, eps, bias=None, trans_b=False, dropout_p=0.0, seed=0, stream_id=0, want_sum=True):
    """nst_gemm_add_layernorm_fwd: delta = bf16(dropout(A @ W + bias)); xs = x + delta (f32); y = LayerNorm(xs) (bf16).
    A [rows, k] bf16, W [k, 256] (or [256, k] with trans_b), x [rows, 256] f32 -> (y, xs | None, mean, rstd): what
    gemm(.., bias, dropout) followed by add_layernorm_fwd returns, in one launch."""
    d, rows, n, k = _rowgemm_desc(A, W, trans_b, dropout_p, seed, stream_id, eps)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == rows * n
    y = torch.empty(rows, n, dtype=torch.bfloat16, device=A.device)
    xs = torch.empty(rows, n, dtype=torch.float32, device=A.device) if want_sum else None
    mean = torch.empty(rows, dtype=torch.float32, device=A.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=A.device)
    ev = PROBE.begin("gemm_rows")
    check(lib.nst_gemm_add_layernorm_fwd(C.byref(d), _p(A), _p(W), _p(bias), _p(x), _p(xs), _p(gamma), _p(beta), _p(y), _p(mean),
                                         _p(rstd), _stream()), "gemm_add_layernorm_fwd")
    if ev is not None:
        PROBE.end(ev, 2.0 * rows * n * k, (rows * k + k * n) * 2 + rows * n * (4 + 2 + (4 if want_sum else 0)))
    return y, xs, mean, rstd


def gemm_layernorm_bwd(A, W, x, gamma, mean, rstd, dgamma, dbeta, accumulate=False, dres=None, emit_dropout=None, batch=None,
                       trans_b=True):
    """nst_gemm_layernorm_bwd: g = bf16(A @ W^T) (trans_b: W is the dense kernel [256, k] as stored), then exactly
    layernorm_bwd(g, x, ...) with the f32 saved input: dx (+ dres), optionally (dx, dz) with emit_dropout=(p, seed, site);
    dgamma / dbeta finished here or by the batch's next flush."""
    d, rows, n, k = _rowgemm_desc(A, W, trans_b)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == rows * n
    assert dres is None or (dres.is_contiguous() and dres.dtype == torch.bfloat16 and dres.numel() == rows * n)
    dx = torch.empty(rows, n, dtype=torch.bfloat16, device=A.device)
    dz = None
    if emit_dropout is not None:
        d.dropout_p, d.seed, d.stream_id = float(emit_dropout[0]), int(emit_dropout[1]), int(emit_dropout[2])
        dz = torch.empty_like(dx)
    slot = batch.ln_slot(n) if batch is not None else None
    if slot is not None:
        ws_ptr, ws_bytes, job = slot
    else:
        ws = _workspace(64 << 20, A.device)
        ws_ptr, ws_bytes, job = ws.data_ptr(), ws.numel(), None
    ev = PROBE.begin("gemm_rows")
    check(lib.nst_gemm_layernorm_bwd(C.byref(d), _p(A), _p(W), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dz),
                                     _p(dgamma), _p(dbeta), int(accumulate), ws_ptr, ws_bytes, job, _stream()),
          "gemm_layernorm_bwd")
    if ev is not None:
        PROBE.end(ev, 2.0 * rows * n * k, (rows * k + k * n) * 2 + rows * n * (4 + 2 + 2 * (dres is not None) + 2 * (dz is not None)))
    return (dx, dz) if emit_dropout is not None else dx


def gemm_rowdot256(A, W, rowdot, trans_b=True):
    """nst_gemm_rowdot256: C [rows, 256] = bf16(A @ W^T); rowdot=(src [rows, 256] bf16, dst f32 [rows/T, 4, T], T) leaves the
    per-head row sums of C o src (the attention backward's delta)."""
    d, rows, n, k = _rowgemm_desc(A, W, trans_b)
    out = torch.empty(rows, n, dtype=torch.bfloat16, device=A.device)
    src = dst = None
    T = 0
    if rowdot is not None:
        src, dst, T = rowdot
        assert src.dtype == torch.bfloat16 and src.is_contiguous() and src.numel() == rows * n
        assert dst.dtype == torch.float32 and dst.is_contiguous() and dst.numel() == rows * (n // 64)
    ev = PROBE.begin("gemm_rows")
    check(lib.nst_gemm_rowdot256(C.byref(d), _p(A), _p(W), _p(out), _p(src), _p(dst), int(T), _stream()), "gemm_rowdot256")
    if ev is not None:
        PROBE.end(ev, 2.0 * rows * n * k, (rows * k + k * n) * 2 + rows * n * (2 + 2 * (src is not None)))
    return out


# ------------------------------------------------------------------------------------------------ fused feed-forward
def ffn_supported(d_model, filter_size, dtype):
    """Whether the one-launch feed-forward pair exists for this shape (d_model 256, filter a multiple of 128, bf16)."""
    return dtype == torch.bfloat16 and bool(lib.nst_ffn_supported(int(d_model), int(filter_size), NST_BF16))


def _ffn_desc(rows, d, f, hidden_p, hidden_seed, hidden_site, out_p=0.0, out_seed=0, out_site=0):
    desc = NstFfnDesc()
    desc.rows, desc.d_model, desc.filter_size, desc.dtype = rows, d, f, NST_BF16
    desc.hidden_dropout_p, desc.hidden_seed, desc.hidden_stream_id = hidden_p, hidden_seed, hidden_site
    desc.output_dropout_p, desc.output_seed, desc.output_stream_id = out_p, out_seed, out_site
    return desc


def ffn_fwd(x, w1t, b1, w2t, b2, residual=None, hidden_p=0.0, hidden_seed=0, hidden_site=0, out_p=0.0, out_seed=0,
            out_site=0, save_gate_bits=False):
    """y = residual + dropout_out(dropout_hidden(relu(x @ w1 + b1)) @ w2 + b2) in one launch; w1t [F, d] / w2t [d, F] are the
    TRANSPOSED bf16 copies of dense1/kernel [d, F] / dense2/kernel [F, d].  Returns (y, hidden); hidden is the saved activation.
    save_gate_bits: returns (y, hidden, gate_bits) -- gate_bits is an opaque uint8 tensor holding `hidden > 0` as one bit per
    element for ffn_bwd (None where the library has no bit path for the shape)."""
    rows, d = x.shape
    f = w1t.shape[0]
    assert x.is_contiguous() and w1t.is_contiguous() and w2t.is_contiguous() and w1t.shape == (f, d) and w2t.shape == (d, f)
    assert residual is None or (residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype)
    assert rows * f * 2 < (1 << 32)
    hidden = torch.empty(rows, f, dtype=x.dtype, device=x.device)
    y = torch.empty_like(x)
    desc = _ffn_desc(rows, d, f, hidden_p, hidden_seed, hidden_site, out_p, out_seed, out_site)
    bits = None
    if save_gate_bits:
        nbytes = int(lib.nst_ffn_gate_bits_bytes(C.byref(desc)))
        if nbytes > 0:
            bits = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            desc.gate_bits, desc.gate_bits_bytes = bits.data_ptr(), nbytes
    ev = PROBE.begin("ffn_fwd")
    check(lib.nst_ffn_fwd(C.byref(desc), _p(x), _p(w1t), _p(b1), _p(w2t), _p(b2), _p(residual), _p(hidden), _p(y), _stream()),
          "ffn_fwd")
    PROBE.end(ev, 4.0 * rows * d * f)
    return (y, hidden, bits) if save_gate_bits else (y, hidden)


def ffn_bwd(dy, hidden, w2, w1, hidden_p=0.0, residual=None, gate_bits=None):
    """dhidden = (dy @ w2^T) * gate(hidden); dx = dhidden @ w1^T (+ residual) in one launch; w2 [F, d], w1 [d, F] as stored.
    gate_bits: what ffn_fwd(save_gate_bits=True) returned for this `hidden` (read instead of it).  Returns (dx, dhidden)."""
    rows, d = dy.shape
    f = w2.shape[0]
    assert dy.is_contiguous() and hidden.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()
    assert w2.shape == (f, d) and w1.shape == (d, f) and hidden.shape == (rows, f)
    assert residual is None or (residual.is_contiguous() and residual.shape == dy.shape and residual.dtype == dy.dtype)
    dhidden = torch.empty_like(hidden)
    dx = torch.empty_like(dy)
    desc = _ffn_desc(rows, d, f, hidden_p, 0, 0)
    if gate_bits is not None:
        assert gate_bits.dtype == torch.uint8 and gate_bits.is_contiguous() and gate_bits.device == dy.device
        desc.gate_bits, desc.gate_bits_bytes = gate_bits.data_ptr(), gate_bits.numel()
    ev = PROBE.begin("ffn_bwd")
    check(lib.nst_ffn_bwd(C.byref(desc), _p(dy), _p(hidden), _p(w2), _p(w1), _p(residual), _p(dhidden), _p(dx), _stream()),
          "ffn_bwd")
    PROBE.end(ev, 4.0 * rows * d * f)
    return dx, dhidden


def ffn_ln_supported(rows, d, f):
    """Whether the feed-forward pair carries the wrapper's row stages for this call (nst_ffn_ln_supported: the eight-wave kernel's
    shapes -- >= 20 480 rows, d_model 256 -- with gate bits)."""
    desc = _ffn_desc(int(rows), int(d), int(f), 0.0, 0, 0)
    return int(lib.nst_ffn_ln_supported(C.byref(desc)))      # 0 | 1 (one launch) | 2 (hidden dimension split + a row launch)


def _ffn_slabs(desc, device):
    """Scratch for the split form's f32 partial sums (None when the call is a single launch)."""
    nbytes = int(lib.nst_ffn_ln_slab_bytes(C.byref(desc)))
    return (torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes) if nbytes > 0 else (None, 0)


def ffn_add_layernorm_fwd(x, w1t, b1, w2t, b2, x_res, gamma, beta, eps, hidden_p=0.0, hidden_seed=0, hidden_site=0, out_p=0.0,
                          out_seed=0, out_site=0, want_sum=True):
    """nst_ffn_add_layernorm_fwd: ffn_fwd (no residual, gate bits saved) and add_layernorm_fwd on the float32 stream x_res in one
    launch -> (y bf16 = LayerNorm(x_res + delta), xs f32 | None, mean, rstd, hidden, gate_bits)."""
    rows, d = x.shape
    f = w1t.shape[0]
    assert x.is_contiguous() and w1t.is_contiguous() and w2t.is_contiguous() and w1t.shape == (f, d) and w2t.shape == (d, f)
    assert x_res.dtype == torch.float32 and x_res.is_contiguous() and x_res.numel() == rows * d and rows * f * 2 < (1 << 32)
    hidden = torch.empty(rows, f, dtype=x.dtype, device=x.device)
    y = torch.empty_like(x)
    xs = torch.empty(rows, d, dtype=torch.float32, device=x.device) if want_sum else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    desc = _ffn_desc(rows, d, f, hidden_p, hidden_seed, hidden_site, out_p, out_seed, out_site)
    nbytes = rows * (f // 32) * 4       # (both forms of this entry write the bits; ffn_layernorm_bwd reads them)
    bits = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    desc.gate_bits, desc.gate_bits_bytes = bits.data_ptr(), nbytes
    slabs, slab_bytes = _ffn_slabs(desc, x.device)
    ev = PROBE.begin("ffn_fwd")
    check(lib.nst_ffn_add_layernorm_fwd(C.byref(desc), _p(x), _p(w1t), _p(b1), _p(w2t), _p(b2), _p(x_res), _p(xs), _p(gamma), _p(beta),
                                        float(eps), _p(hidden), _p(y), _p(mean), _p(rstd), _p(slabs), slab_bytes, _stream()),
          "ffn_add_layernorm_fwd")
    PROBE.end(ev, 4.0 * rows * d * f)
    return y, xs, mean, rstd, hidden, bits


def ffn_layernorm_bwd(dy, hidden, w2, w1, x_ln, gamma, mean, rstd, dgamma, dbeta, hidden_p=0.0, gate_bits=None, accumulate=False,
                      dres=None, emit_dropout=None, batch=None):
    """nst_ffn_layernorm_bwd: ffn_bwd followed by layernorm_bwd (f32 saved input x_ln) in one launch
    -> (dx, dz | None, dhidden)."""
    rows, d = dy.shape
    f = w2.shape[0]
    assert dy.is_contiguous() and hidden.is_contiguous() and w1.is_contiguous() and w2.is_contiguous() and gate_bits is not None
    assert w2.shape == (f, d) and w1.shape == (d, f) and hidden.shape == (rows, f)
    assert x_ln.dtype == torch.float32 and x_ln.is_contiguous() and x_ln.numel() == rows * d
    assert dres is None or (dres.is_contiguous() and dres.dtype == dy.dtype and dres.numel() == rows * d)
    dhidden = torch.empty_like(hidden)
    dx = torch.empty_like(dy)
    p, seed, site = emit_dropout if emit_dropout is not None else (0.0, 0, 0)
    dz = torch.empty_like(dy) if emit_dropout is not None else None
    desc = _ffn_desc(rows, d, f, hidden_p, 0, 0)
    desc.gate_bits, desc.gate_bits_bytes = gate_bits.data_ptr(), gate_bits.numel()
    slot = batch.ln_slot(d) if batch is not None else None
    if slot is not None:
        ws_ptr, ws_bytes, job = slot
    else:
        ws = _workspace(64 << 20, dy.device)
        ws_ptr, ws_bytes, job = ws.data_ptr(), ws.numel(), None
    slabs, slab_bytes = _ffn_slabs(desc, dy.device)
    ev = PROBE.begin("ffn_bwd")
    check(lib.nst_ffn_layernorm_bwd(C.byref(desc), _p(dy), _p(hidden), _p(w2), _p(w1), _p(x_ln), _p(gamma), _p(mean), _p(rstd),
                                    _p(dres), _p(dhidden), _p(dx), _p(dz), float(p), int(seed), int(site), _p(dgamma), _p(dbeta),
                                    int(accumulate), ws_ptr, ws_bytes, job, _p(slabs), slab_bytes, _stream()), "ffn_layernorm_bwd")
    PROBE.end(ev, 4.0 * rows * d * f)
    return dx, dz, dhidden


def transpose_bf16(table, njobs, total_tiles):
    """Runs a device-resident table of NstTransposeJob (see ParamStore.finalize): dst[cols, rows] = src[rows, cols]^T."""
    check(lib.nst_transpose_bf16(_p(table), njobs, total_tiles, _stream()), "transpose_bf16")


def pack2d(table, njobs, total_blocks):
    """Runs a device-resident table of NstPack2dJob (see ParamStore._build_packed): strided block copies, one launch."""
    check(lib.nst_pack2d(_p(table), njobs, total_blocks, _stream()), "pack2d")


def grad_clip(grad, table, nentries, seg_first, nseg, pre_scale=1.0, clip_value=None, clip_norm=None):
    """In place on the flat gradient buffer: g *= pre_scale, then clamp to +-clip_value or scale every tensor to an L2 norm
    of at most clip_norm (neurst_hip.h nst_grad_clip; gradaccum_keras_model.py:228-233)."""
    assert grad.dtype == torch.float32 and grad.is_contiguous()
    ws = torch.empty(nentries + nseg, dtype=torch.float32, device=grad.device)
    check(lib.nst_grad_clip(_p(grad), _p(table), nentries, _p(seg_first), nseg, _p(ws), ws.numel(), float(pre_scale),
                            float(clip_value or 0.0), float(clip_norm or 0.0), _stream()), "grad_clip")


def colsum(x, out, accumulate=False):
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32
    ws = _workspace(64 << 20, x.device)
    check(lib.nst_colsum(_p(x), _p(out), x.shape[0], x.shape[1], x.stride(0), _dt(x), int(accumulate), ws.data_ptr(),
                         ws.numel(), _stream()), "colsum")
    return out


# ------------------------------------------------------------------------------------------------ attention
def _attn_desc(q, k, v, out, H, dh, causal, dropout_p, seed, stream_id, causal_offset=0):
    # q [B,Tq,*] k,v [B,Tk,*] views whose last dim starts at this tensor's head 0 (row stride = stride(1))
    d = NstAttnDesc()
    d.B, d.Tq, d.Tk, d.H, d.dh = q.shape[0], q.shape[1], k.shape[1], H, dh
    for t in (q, out):
        assert t.stride(-1) == 1 and t.stride(0) == t.shape[1] * t.stride(1), "batch stride must be T*row stride"
    for t in (k, v):  # k / v may be the filled prefix of a longer cache: any batch stride >= Tk rows
        assert t.stride(-1) == 1 and t.stride(0) >= t.shape[1] * t.stride(1), "k/v batch stride smaller than Tk rows"
    d.ldq, d.ldk, d.ldv, d.ldo = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
    d.bsk, d.bsv = k.stride(0), v.stride(0)
    d.dtype = _dt(q)
    d.scale = float(dh) ** -0.5
    d.causal = int(causal)
    d.causal_offset = int(causal_offset)
    d.float_min = FLOAT_MIN
    d.dropout_p = dropout_p
    d.seed, d.stream_id = seed, stream_id
    return d


def attention_fwd(q, k, v, H, dh, key_bias=None, causal=False, dropout_p=0.0, seed=0, stream_id=0, causal_offset=0):
    """q [B,Tq,H*dh-view], k/v [B,Tk,H*dh-view] (may be column slices of a packed projection).
    Returns (out, lse, drop_mask); drop_mask (keep bits written by the kernel, None when dropout_p == 0) must be
    handed to attention_bwd."""
    B, Tq = q.shape[0], q.shape[1]
    out = torch.empty(B, Tq, H * dh, dtype=q.dtype, device=q.device)
    lse = torch.empty(B, H, Tq, dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, out, H, dh, causal, dropout_p, seed, stream_id, causal_offset)
    mask = None
    if dropout_p > 0:
        mask = torch.empty(lib.nst_attention_dropout_mask_bytes(C.byref(d)) // 8, dtype=torch.int64, device=q.device)
        d.dropout_mask, d.dropout_mask_bytes = mask.data_ptr(), mask.numel() * 8
    ev = PROBE.begin("attention_fwd")
    check(lib.nst_attention_fwd(C.byref(d), _p(q), _p(k), _p(v), _p(key_bias), _p(out), _p(lse), _stream()),
          "attention_fwd")
    PROBE.end(ev, 4.0 * B * H * Tq * k.shape[1] * dh)
    return out, lse, mask


def rowdot_supported(x, n):
    """gemm(rowdot=...) needs the bf16 stream kernel and whole 64-column heads."""
    return x.is_cuda and x.dtype == torch.bfloat16 and n % 64 == 0


def attention_bwd(q, k, v, out, dout, lse, dq, dk, dv, H, dh, key_bias=None, causal=False, dropout_p=0.0, seed=0,
                  stream_id=0, drop_mask=None, causal_offset=0, delta=None):
    """delta [B,H,Tq] f32: rowsum(dout o out) per head if the GEMM that produced dout already computed it (gemm(rowdot=...))."""
    assert dout.is_contiguous() and out.is_contiguous()
    assert dq.stride(1) == q.stride(1) and dk.stride(1) == k.stride(1) and dv.stride(1) == v.stride(1)
    have_delta = delta is not None
    if not have_delta:
        delta = torch.empty_like(lse)
    assert delta.dtype == torch.float32 and delta.is_contiguous() and delta.numel() == lse.numel()
    d = _attn_desc(q, k, v, out, H, dh, causal, dropout_p, seed, stream_id, causal_offset)
    if dropout_p > 0:
        assert drop_mask is not None, "attention_bwd: dropout needs the mask written by attention_fwd"
        d.dropout_mask, d.dropout_mask_bytes = drop_mask.data_ptr(), drop_mask.numel() * 8
    if q.dtype == torch.bfloat16:   # scratch for dS^T: the dQ product then needs no second pass over the scores
        B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
        ds = torch.empty(B * H * ((Tk + 127) // 128 * 128) * ((Tq + 63) // 64 * 64), dtype=torch.bfloat16, device=q.device)
        d.ds_workspace, d.ds_workspace_bytes = ds.data_ptr(), ds.numel() * 2
    ev = PROBE.begin("attention_bwd")
    check(lib.nst_attention_bwd(C.byref(d), _p(q), _p(k), _p(v), _p(key_bias), None if have_delta else _p(out), _p(dout), _p(lse), _p(delta),
                                _p(dq), _p(dk), _p(dv), _stream()), "attention_bwd")
    PROBE.end(ev, 8.0 * q.shape[0] * H * q.shape[1] * k.shape[1] * dh)


# ------------------------------------------------------------------------------------------------ conv front end
def conv1_ln_relu_fwd(src, w1, b1, gamma, beta, layer_norm, eps, out_dtype):
    B, T, F = src.shape
    Cc = w1.shape[-1]
    T1, F1 = (T + 1) // 2, (F + 1) // 2
    out = torch.empty(B, T1, F1, Cc, dtype=out_dtype, device=src.device)
    mean = torch.empty(B * T1 * F1, dtype=torch.float32, device=src.device) if layer_norm else None
    rstd = torch.empty(B * T1 * F1, dtype=torch.float32, device=src.device) if layer_norm else None
    assert src.is_contiguous() and src.dtype == torch.float32 and w1.is_contiguous()
    check(lib.nst_conv1_ln_relu_fwd(_p(src), _p(w1), _p(b1), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), B, T, F,
                                    Cc, int(layer_norm), float(eps), _dt(out), _stream()), "conv1_fwd")
    return out, mean, rstd


def conv1_ln_relu_bwd(src, w1, b1, gamma, beta, mean, rstd, dout, dw1, db1, dgamma, dbeta, layer_norm, eps,
                      accumulate=False):
    B, T, F = src.shape
    Cc = w1.shape[-1]
    assert dout.is_contiguous()
    check(lib.nst_conv1_ln_relu_bwd(_p(src), _p(w1), _p(b1), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dout),
                                    _p(dw1), _p(db1), _p(dgamma), _p(dbeta), B, T, F, Cc, int(layer_norm), float(eps),
                                    _dt(dout), int(accumulate), _stream()), "conv1_bwd")


def conv2_fwd(x, w2, b2, relu=False):
    B, T1, F1, Cc = x.shape
    assert x.is_contiguous() and w2.is_contiguous() and w2.dtype == x.dtype
    y = torch.empty(B, (T1 + 1) // 2, (F1 + 1) // 2, Cc, dtype=x.dtype, device=x.device)
    ev = PROBE.begin("conv2_fwd")
    check(lib.nst_conv2_fwd(_p(x), _p(w2), _p(b2), _p(y), B, T1, F1, Cc, int(relu), _dt(x), _stream()), "conv2_fwd")
    PROBE.end(ev, 2.0 * y.numel() * 9 * Cc)
    return y


def conv2_dgrad(dy, w2, T1, F1):
    B, T2, F2, Cc = dy.shape
    assert dy.is_contiguous() and w2.dtype == dy.dtype
    dx = torch.empty(B, T1, F1, Cc, dtype=dy.dtype, device=dy.device)
    ev = PROBE.begin("conv2_dgrad")
    check(lib.nst_conv2_dgrad(_p(dy), _p(w2), _p(dx), B, T1, F1, Cc, _dt(dy), _stream()), "conv2_dgrad")
    PROBE.end(ev, 2.0 * dy.numel() * 9 * Cc)
    return dx


def conv2_wgrad(x, dy, dw2, db2=None, accumulate=False):
    """dw2 (+)= x (*) dy ; db2 [C] (+)= sum over pixels of dy (same pass over dy)."""
    B, T1, F1, Cc = x.shape
    assert x.is_contiguous() and dy.is_contiguous() and dw2.dtype == torch.float32
    assert db2 is None or (db2.dtype == torch.float32 and db2.numel() == Cc and db2.is_contiguous())
    ws = _workspace(192 << 20, x.device)   # split-K slabs: up to 8 * 16 slices of [9C + 1, C] f32
    ev = PROBE.begin("conv2_wgrad")
    check(lib.nst_conv2_wgrad(_p(x), _p(dy), _p(dw2), _p(db2), B, T1, F1, Cc, _dt(x), int(accumulate), ws.data_ptr(),
                              ws.numel(), _stream()), "conv2_wgrad")
    PROBE.end(ev, 2.0 * dy.numel() * 9 * Cc)


# ------------------------------------------------------------------------------------------------ embedding / elementwise
def embedding_fwd(table, ids, posenc, L, emb_scale, dropout_p=0.0, seed=0, stream_id=0):
    V, d = table.shape
    ids = ids.contiguous()
    rows = ids.numel()
    out = torch.empty(*ids.shape, d, dtype=table.dtype, device=table.device)
    assert ids.dtype == torch.int64
    check(lib.nst_embedding_fwd(_p(table), _p(ids), _p(posenc), _p(out), rows, L, d, V, emb_scale, dropout_p, seed,
                                stream_id, _dt(table), _stream()), "embedding_fwd")
    return out


def embedding_bwd(dout, ids, dtable, emb_scale, dropout_p=0.0, seed=0, stream_id=0):
    V, d = dtable.shape
    assert dout.is_contiguous() and dtable.dtype == torch.float32
    check(lib.nst_embedding_bwd(_p(dout), _p(ids.contiguous()), _p(dtable), ids.numel(), d, V, emb_scale, dropout_p,
                                seed, stream_id, _dt(dout), _stream()), "embedding_bwd")


def scale_posenc_dropout_fwd(x, posenc, period, scale, dropout_p=0.0, seed=0, stream_id=0):
    assert x.is_contiguous()
    d = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.nst_scale_posenc_dropout_fwd(_p(x), _p(posenc), _p(y), x.numel() // d, d, period, scale, dropout_p, seed,
                                           stream_id, _dt(x), _stream()), "scale_posenc_dropout_fwd")
    return y


def scale_dropout_bwd(dy, scale, dropout_p=0.0, seed=0, stream_id=0):
    assert dy.is_contiguous()
    dx = torch.empty_like(dy)
    check(lib.nst_scale_dropout_bwd(_p(dy), _p(dx), dy.numel(), scale, dropout_p, seed, stream_id, _dt(dy), _stream()),
          "scale_dropout_bwd")
    return dx


# ------------------------------------------------------------------------------------------------ criterion / optimizer
def ls_xent_fwd(logits, labels, weights, label_smoothing):
    assert logits.dim() == 2 and logits.stride(1) == 1
    rows, V = logits.shape
    xent = torch.empty(rows, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    check(lib.nst_ls_xent_fwd(_p(logits), _p(labels), _p(weights), _p(xent), _p(lse), rows, V, logits.stride(0),
                              label_smoothing, _dt(logits), _stream()), "ls_xent_fwd")
    return xent, lse


def ls_xent_bwd(logits, labels, weights, lse, label_smoothing, gscale, out=None, gscale_dev=None):
    rows, V = logits.shape
    dlogits = out if out is not None else torch.empty_like(logits)
    assert dlogits.stride(0) == logits.stride(0)
    check(lib.nst_ls_xent_bwd(_p(logits), _p(labels), _p(weights), _p(lse), _p(dlogits), rows, V, logits.stride(0),
                              label_smoothing, gscale, _p(gscale_dev), _dt(logits), _stream()), "ls_xent_bwd")
    return dlogits


def seq_mask(lengths, max_len, on_token, on_padding, halvings=0, stride=2):
    """[B, max_len] f32: on_token where t < len', on_padding elsewhere; len' = `halvings` times ceil(len / stride) (one launch)."""
    lengths = lengths.to(torch.int64).contiguous()
    out = torch.empty(lengths.shape[0], int(max_len), dtype=torch.float32, device=lengths.device)
    check(lib.nst_seq_mask(_p(lengths), _p(out), lengths.shape[0], int(max_len), int(halvings), int(stride), float(on_token),
                           float(on_padding), _stream()), "seq_mask")
    return out


def xent_reduce(xent, weights):
    """xent, weights [B, L] f32 -> (nll_sum [B], n_tokens [B], loss [1], inv_tokens [1]) in one launch."""
    B, L = weights.shape
    assert xent.numel() == B * L and xent.is_contiguous() and weights.is_contiguous()
    assert xent.dtype == torch.float32 and weights.dtype == torch.float32
    o = torch.empty(2 * B + 2, dtype=torch.float32, device=xent.device)
    check(lib.nst_xent_reduce(_p(xent), _p(weights), B, L, o.data_ptr(), o.data_ptr() + 4 * B, o.data_ptr() + 8 * B,
                              o.data_ptr() + 8 * B + 4, _stream()), "xent_reduce")
    return o[:B], o[B:2 * B], o[2 * B:2 * B + 1], o[2 * B + 1:]


def adam_update(p, m, v, g, shadow, lr_t, beta1, beta2, eps, grad_scale=1.0, loss_scale_state=None):
    """lr_t: float, or a 1-element float32 DEVICE tensor read when the kernel runs (graph replay).  loss_scale_state: the
    4-float device state of loss_scale_update (skip on overflow, unscale otherwise)."""
    n = p.numel()
    if torch.is_tensor(lr_t) or loss_scale_state is not None:
        lr_dev = lr_t if torch.is_tensor(lr_t) else None
        assert lr_dev is None or (lr_dev.dtype == torch.float32 and lr_dev.numel() == 1)
        check(lib.nst_adam_update_dev(_p(p), _p(m), _p(v), _p(g), _p(shadow), n, 0.0 if lr_dev is not None else lr_t, _p(lr_dev),
                                      beta1, beta2, eps, grad_scale, _p(loss_scale_state), _stream()), "adam_update_dev")
        return
    check(lib.nst_adam_update(_p(p), _p(m), _p(v), _p(g), _p(shadow), n, lr_t, beta1, beta2, eps, grad_scale,
                              _stream()), "adam_update")


def loss_scale_update(grad, state, growth_steps, multiplier, counter):
    """RevisedDynamicLossScale.update over the flat (already exchanged) gradient buffer; `counter` is a zeroed int32[1]."""
    assert grad.dtype == torch.float32 and state.dtype == torch.float32 and state.numel() == 4 and counter.numel() >= 1
    check(lib.nst_loss_scale_update(_p(grad), grad.numel(), _p(state), float(growth_steps), float(multiplier), _p(counter),
                                    counter.numel() * counter.element_size(), _stream()), "loss_scale_update")


def cast_f32_to_bf16(src, dst):
    check(lib.nst_cast_f32_to_bf16(_p(src), _p(dst), src.numel(), _stream()), "cast_f32_to_bf16")


def probe_mfma(device):
    c16 = torch.zeros(256, dtype=torch.float32, device=device)
    c16f = torch.zeros(256, dtype=torch.float32, device=device)
    tr = torch.zeros(512, dtype=torch.int16, device=device)
    check(lib.nst_probe_mfma(_p(c16), _p(c16f), _p(tr), _stream()), "probe_mfma")
    return c16, c16f, tr
