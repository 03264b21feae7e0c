"""MI355X-native restatement of neurst/layers/common_layers.py.

Every layer owns its parameters (TF variable names, TF layouts) in the runtime's flat ParamStore and
exposes an explicit ``forward`` / ``backward`` pair: the backward pass is hand-scheduled (no autograd
tape), writes parameter gradients straight into the flat gradient buffer and returns the input gradient.
Activations are 2-D ``[B*T, d]`` row-major tensors.

  PrePostProcessingWrapper   neurst/layers/common_layers.py:23-92   (LN -> layer -> dropout -> residual)
  TransformerFFN             neurst/layers/common_layers.py:95-160
  MultiHeadDenseLayer        neurst/layers/common_layers.py:163-295
  PositionEmbeddingWrapper   neurst/layers/common_layers.py:298-446
"""
import math
import os

import torch

from neurst_amd import kernels as K


def glorot_uniform(shape, gen, fan_in=None, fan_out=None):
    """Keras glorot_uniform: U(-l, l), l = sqrt(6/(fan_in+fan_out))."""
    if fan_in is None:
        if len(shape) == 1:
            fan_in = fan_out = shape[0]
        elif len(shape) == 2:
            fan_in, fan_out = shape
        else:  # conv HWIO
            rf = int(math.prod(shape[:-2]))
            fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(*shape, generator=gen, dtype=torch.float64) * 2 - 1).mul_(lim).float()


_WGRAD_UNITS = 256          # workgroups a split-K weight gradient is cut into
_WGRAD_SPLIT8 = True        # split factors in multiples of 8: a K slice per XCD
_WGRAD_UNITS_SMALL = 128    # gradients of fewer than 8 tiles (256 x 256 kernels)
# the one-launch feed-forward wins when its 128-row workgroups fill the chip (encoder: 28 800 rows at the benchmark shape);
# below that (decoder: 9 600 rows) the two persistent GEMMs are as fast or faster (scripts/ffn_bench.py)
_WGRAD_GROUP_MAX_BYTES = 6 << 30    # activations a pending weight-gradient group may keep alive before it is launched early
_FFN_FUSED_MIN_ROWS = int(os.environ.get("NST_FFN_MIN_ROWS", "16384"))
# the backward pair.  It holds all 160 KB of a CU's LDS, so no
# weight-gradient workgroup shares its CUs, but at 97 us against 124 us for the two GEMMs it still wins in the step
# (17.03 vs 17.31 ms, profiles/r02_ffn_*.json)
_FFN_FUSED = True          # the one-launch feed-forward pair where it is supported (tests pin the two-GEMM path with False)
_FFN_FUSED_BWD = True


def _wgrad_split(rows, k_in, n_out, dtype, units=None):
    """split-K factor for dW[k_in, n_out] = X^T dY reduced over `rows`: tiles*split ~ _WGRAD_UNITS workgroups."""
    tiles = ((k_in + 127) // 128) * ((n_out + 127) // 128)
    bk = 64 if dtype == torch.bfloat16 else 32
    kt = (rows + bk - 1) // bk
    if units is None and tiles < 8:
        units = _WGRAD_UNITS_SMALL
    split = max(1, min((units or _WGRAD_UNITS) // max(tiles, 1), kt // 8))
    if split >= 8 and tiles >= 8 and _WGRAD_SPLIT8:   # few tiles per slice: little to share, keep the finer split
        # multiples of 8: the stream kernel then pins every K slice to one XCD (its tiles share the slice's rows in that
        # L2).  The library re-derives the count from ceil(kt / ceil(kt / split)); take the nearest multiple that survives.
        def survives(s):
            return s >= 8 and s <= kt // 8 and -(-kt // -(-kt // s)) == s
        near = (split + 4) // 8 * 8
        for delta in range(0, near, 8):
            hit = next((c for c in (near + delta, near - delta) if survives(c)), None)
            if hit is not None:
                return hit
    return split


class Layer(object):
    """Base: keeps the runtime and a name scope."""

    def __init__(self, rt, name):
        self.rt, self.name = rt, name

    def _site(self):
        return self.rt.new_dropout_site()


class ResidualStream(object):
    """The residual stream between the sub-layers of a pre-norm bf16 stack, carried in float32.

    The reference computes inputs + dropout(layer(LN(inputs))) in float32 (common_layers.py:73-85).  Rounding that sum to bf16
    after every sub-layer is the ONE class of rounding points that puts the bf16 gradients of the 12 + 6-layer model outside
    1e-2 of the reference (scripts/rounding_point_study.py, profiles/r05_rounding_point_study_b*.json: 1.71e-2 -> 1.31e-2 at 32
    utterances with this stream, every other class moves the figure by < 4 %).  So the sum is never rounded: a sub-layer's last
    kernel writes its contribution `delta` = dropout(layer(LN(x))) as bf16 WITHOUT the residual, and the next LayerNorm
    (nst_add_layernorm_fwd) adds it to the float32 `x`, writes the new sum once and normalises it in the same pass.
    `x`: the running sum (f32; in front of the first sub-layer the bf16 embedding output), `delta`: bf16 or None."""
    __slots__ = ("x", "delta")

    def __init__(self, x, delta=None):
        self.x, self.delta = x, delta

    @staticmethod
    def supported(rt, dim, pre_norm=True):
        return bool(pre_norm) and K.add_layernorm_supported(dim, rt.dtype)


# whole-row products carry the wrapper's LayerNorm stages in their epilogue (tests and A/B runs pin the unfused pairs: NST_ROW_FUSION=0)
_ROW_FUSION = os.environ.get("NST_ROW_FUSION", "1") != "0"


class DeferredDelta(object):
    """The LAST product of a sub-layer (attention output projection, feed-forward dense2) that has not been launched: it waits
    for the LayerNorm that consumes the residual stream next, because nst_gemm_add_layernorm_fwd runs the product, the wrapper's
    dropout, the residual add and that LayerNorm in ONE launch (the bf16 `delta` then never travels).  Stands where
    ResidualStream.delta stands; `plain()` launches the product on its own (the pair it replaces) where the fused kernel cannot
    take the call (a bf16 stream in front of the first sub-layer of a stack)."""
    __slots__ = ("A", "dense", "epi")

    def __init__(self, A, dense, epi):
        self.A, self.dense, self.epi = A, dense, epi

    def plain(self):
        return self.dense.forward(self.A, **self.epi)

    def fused(self, x, gamma, beta, eps, want_sum):
        d, e = self.dense, self.epi
        return K.gemm_add_layernorm_fwd(self.A, d.kernel.compute, x, gamma, beta, eps, bias=None if d.bias is None else d.bias.data,
                                        dropout_p=e.get("dropout_p", 0.0), seed=e.get("seed", 0), stream_id=e.get("stream_id", 0),
                                        want_sum=want_sum)


class DeferredFfn(DeferredDelta):
    """The whole feed-forward pair, not launched yet: nst_ffn_add_layernorm_fwd runs both products, both dropouts, the residual add
    and the next LayerNorm in one launch (the eight-wave kernel's shapes, training).  Either way the launch leaves the layer's
    saved state (input, hidden activation, gate bits) behind."""
    __slots__ = ("ffn", "p")

    def __init__(self, ffn, x, p, epi):
        super().__init__(x, None, epi)
        self.ffn, self.p = ffn, p

    def plain(self):
        f, e = self.ffn, self.epi
        y, h, bits = K.ffn_fwd(self.A, f._w1t.t, f.dense1.bias.data, f._w2t.t, f.dense2.bias.data, residual=None, hidden_p=self.p,
                               hidden_seed=f.rt.step_seed, hidden_site=f.site, out_p=e.get("dropout_p", 0.0), out_seed=e.get("seed", 0),
                               out_site=e.get("stream_id", 0), save_gate_bits=True)
        f._saved = (self.A, h, self.p, bits)      # (bits None where nst_ffn_fwd has no bit path: the backward then gates by h)
        return y

    def fused(self, x, gamma, beta, eps, want_sum):
        f, e = self.ffn, self.epi
        y, xs, mean, rstd, h, bits = K.ffn_add_layernorm_fwd(
            self.A, f._w1t.t, f.dense1.bias.data, f._w2t.t, f.dense2.bias.data, x, gamma, beta, eps, hidden_p=self.p,
            hidden_seed=f.rt.step_seed, hidden_site=f.site, out_p=e.get("dropout_p", 0.0), out_seed=e.get("seed", 0),
            out_site=e.get("stream_id", 0), want_sum=want_sum)
        f._saved = (self.A, h, self.p, bits)
        return y, xs, mean, rstd


class LnBackward(object):
    """The LayerNorm backward of a pre-norm wrapper, offered to the wrapped layer's backward: the layer's last input-gradient
    product (N = d_model = 256: q / qkv projection, dense1) then runs it in its epilogue (nst_gemm_layernorm_bwd) and `done` tells
    the wrapper that what came back is already d(inputs)."""
    __slots__ = ("norm", "dres", "consumer", "done")

    def __init__(self, norm, dres, consumer):
        self.norm, self.dres, self.consumer, self.done = norm, dres, consumer, False

    def eligible(self):
        saved = getattr(self.norm, "_saved", None)
        return saved is not None and saved[0].dtype == torch.float32 and saved[0].is_contiguous() \
            and self.dres.dtype == torch.bfloat16 and self.dres.is_contiguous()

    def _begin(self):
        norm, rt = self.norm, self.norm.rt
        x, mean, rstd = norm._saved
        norm._saved = None
        st = rt.store
        acc = st.acc_flag(norm.gamma)
        st.acc_flag(norm.beta)
        p = self.consumer.drop_rate() if self.consumer is not None else 0.0
        emit = (p, rt.step_seed, self.consumer.site) if p > 0 else None
        return x, mean, rstd, acc, emit

    def _end(self, dx, dzz):
        self.done = True
        if dzz is not None:
            dx._nst_dropped = (self.consumer.site, dzz)
        return dx

    def run(self, dz, kernel):
        """dz [rows, out] bf16, kernel [256, out] as stored: d(inputs) = LayerNorm'(dz . kernel^T) + dres."""
        norm, rt = self.norm, self.norm.rt
        x, mean, rstd, acc, emit = self._begin()
        out = K.gemm_layernorm_bwd(dz, kernel, x, norm.gamma.data, mean, rstd, norm.gamma.grad, norm.beta.grad, accumulate=acc,
                                   dres=self.dres, emit_dropout=emit, batch=rt.wgrad_batch(), trans_b=True)
        return self._end(*(out if emit is not None else (out, None)))

    def run_ffn(self, dz, hidden, w2, w1, hidden_p, bits):
        """The feed-forward pair's input gradient with this LayerNorm backward behind it (nst_ffn_layernorm_bwd)
        -> (d(inputs), d(hidden))."""
        norm, rt = self.norm, self.norm.rt
        x, mean, rstd, acc, emit = self._begin()
        dx, dzz, dh = K.ffn_layernorm_bwd(dz, hidden, w2, w1, x, norm.gamma.data, mean, rstd, norm.gamma.grad, norm.beta.grad,
                                          hidden_p=hidden_p, gate_bits=bits, accumulate=acc, dres=self.dres, emit_dropout=emit,
                                          batch=rt.wgrad_batch())
        return self._end(dx, dzz), dh


class LayerNorm(Layer):
    """tf.keras.layers.LayerNormalization(epsilon, dtype=float32): variables <name>/gamma, <name>/beta."""

    def __init__(self, rt, name, dim, epsilon):
        super().__init__(rt, name)
        self.eps = epsilon
        self.gamma = rt.store.add(name + "/gamma", (dim,), torch.ones(dim))
        self.beta = rt.store.add(name + "/beta", (dim,), torch.zeros(dim))

    def forward(self, x, save=True):
        y, mean, rstd = K.layernorm_fwd(x, self.gamma.data, self.beta.data, self.eps)
        if save:
            self._saved = (x, mean, rstd)
        return y

    def forward_stream(self, stream, save=True, want_sum=True):
        """LayerNorm(stream.x + stream.delta) -> (y bf16, the float32 sum -- None when nothing was added or nobody needs it).
        The backward normalises the saved sum again (nst_layernorm_bwd_mixed reads it as f32)."""
        if stream.delta is None:
            return self.forward(stream.x, save=save), None
        delta = stream.delta
        if isinstance(delta, DeferredDelta):
            if stream.x.dtype == torch.float32 and stream.x.is_contiguous():
                y, xs, mean, rstd = delta.fused(stream.x, self.gamma.data, self.beta.data, self.eps, want_sum or save)
                xs = None if xs is None else xs.view(stream.x.shape)
                y = y.view(stream.x.shape)
                if save:
                    self._saved = (xs, mean, rstd)
                return y, xs
            delta = delta.plain()
        y, xs, mean, rstd = K.add_layernorm_fwd(stream.x, delta, self.gamma.data, self.beta.data, self.eps,
                                                want_sum=want_sum or save)
        if save:
            self._saved = (xs, mean, rstd)
        return y, xs

    def backward(self, dy, dres=None, consumer=None):
        """consumer: the dropout site (object with .site and .drop_rate()) that receives dx next; its dropout backward
        is then produced by the same kernel and attached to dx (see dropped_grad)."""
        x, mean, rstd = self._saved
        self._saved = None
        st = self.rt.store
        acc = st.acc_flag(self.gamma)
        st.acc_flag(self.beta)
        p = consumer.drop_rate() if consumer is not None else 0.0
        if p > 0:
            dx, dz = K.layernorm_bwd(dy, x, self.gamma.data, mean, rstd, self.gamma.grad, self.beta.grad, accumulate=acc,
                                     dres=dres, emit_dropout=(p, self.rt.step_seed, consumer.site), batch=self.rt.wgrad_batch())
            dx._nst_dropped = (consumer.site, dz)
            return dx
        return K.layernorm_bwd(dy, x, self.gamma.data, mean, rstd, self.gamma.grad, self.beta.grad, accumulate=acc,
                               dres=dres, batch=self.rt.wgrad_batch())


def dropped_grad(rt, dy, p, site):
    """dropout backward of dy under mask (step seed, site): taken from the tensor if the producing LayerNorm backward
    already emitted it, computed by the element-wise kernel otherwise."""
    if p <= 0:
        return dy
    tag = getattr(dy, "_nst_dropped", None)
    if tag is not None and tag[0] == site:
        return tag[1]
    return K.scale_dropout_bwd(dy, 1.0, p, rt.step_seed, site)


class Dense(Layer):
    """y = x @ kernel + bias with kernel [in, out] (tf.keras Dense / QuantDense layout).
    Variables <name>/kernel, <name>/bias."""

    def __init__(self, rt, name, in_dim, out_dim, gen, use_bias=True):
        super().__init__(rt, name)
        self.in_dim, self.out_dim = in_dim, out_dim
        self.kernel = rt.store.add(name + "/kernel", (in_dim, out_dim), glorot_uniform((in_dim, out_dim), gen))
        self.bias = rt.store.add(name + "/bias", (out_dim,), torch.zeros(out_dim)) if use_bias else None
        self.wgrad_units = None   # workgroups the weight gradient is cut into (None: _WGRAD_UNITS)
        self.wgrad_grouped = True  # the weight gradient may wait for the stack's grouped launch (Runtime.wgrad_group)

    def forward(self, x, defer_ln=False, **epi):
        """defer_ln (the pre-norm wrapper's float32 residual stream): the product is handed back unlaunched (DeferredDelta) when
        the whole-row kernel can run it together with the next LayerNorm."""
        if defer_ln and _ROW_FUSION and x.dim() == 2 and not (set(epi) - {"dropout_p", "seed", "stream_id"}) \
                and K.rowgemm_supported(x, self.out_dim, self.in_dim):
            return DeferredDelta(x, self, epi)
        return K.gemm(x, self.kernel.compute, x.shape[0], self.out_dim, self.in_dim,
                      bias=None if self.bias is None else self.bias.data, **epi)

    def backward_params(self, x, dz):
        """dkernel (+)= x^T dz ; dbias (+)= colsum(dz)."""
        st = self.rt.store
        rows = x.shape[0]
        acc_k = st.acc_flag(self.kernel)
        grp = self.rt.wgrad_group()
        if grp is not None and self.wgrad_grouped and grp.accepts(x, dz, self.kernel.grad, None if self.bias is None else self.bias.grad):
            # waits for the stack's grouped launch (Runtime.launch_wgrad_group): no split-K, no slabs
            grp.add(x, dz, self.kernel.grad, acc_k, None if self.bias is None else self.bias.grad,
                    False if self.bias is None else st.acc_flag(self.bias))
            # the group keeps x and dz of every queued product alive (benchmark shape: ~1.1 GB for 84 products); past the cap
            # (long ragged batches, transformer_big) it is launched early instead of growing with batch x layers
            if grp.pending_bytes() > _WGRAD_GROUP_MAX_BYTES:
                self.rt.launch_wgrad_group()
            return
        bias_kw = {}
        if self.bias is not None:  # dbias rides on the same pass over dz (ones^T.dz inside the MFMA loop)
            bias_kw = dict(colsum_out=self.bias.grad, colsum_accumulate=st.acc_flag(self.bias))
        self.rt.run_wgrad(lambda: K.gemm(
            x, dz, self.in_dim, self.out_dim, rows, trans_a=True, out=self.kernel.grad, accumulate=acc_k,
            split_k=_wgrad_split(rows, self.in_dim, self.out_dim, x.dtype, self.wgrad_units), batch=self.rt.wgrad_batch(),
            **bias_kw), x, dz)

    def backward_input(self, dz, ln_bwd=None, **epi):
        """dx = dz @ kernel^T  (kernel [in,out] read as the [N,K] operand: no transpose copy).
        ln_bwd (LnBackward): the wrapper's LayerNorm backward rides in the epilogue when the whole-row kernel takes the product."""
        if ln_bwd is not None and _ROW_FUSION and not epi and K.rowgemm_supported(dz, self.in_dim, self.out_dim) and ln_bwd.eligible():
            return ln_bwd.run(dz, self.kernel.compute)
        return K.gemm(dz, self.kernel.compute, dz.shape[0], self.in_dim, self.out_dim, trans_b=True, **epi)


class MultiHeadDenseLayer(Dense):
    """MultiHeadDenseLayer (common_layers.py:163-295).  The kernel is stored exactly as the reference stores it
    (non-output: [in, sum(units)] with column blocks q|k|v, each head-major; output: [H*dh, out]); head
    split/merge are views on the 2-D GEMM output, never copies."""

    def __init__(self, rt, name, in_dim, output_units, num_heads, gen, is_output_transform=False):
        units = output_units if isinstance(output_units, (list, tuple)) else [output_units]
        super().__init__(rt, name, in_dim, sum(units), gen)
        self.units, self.num_heads, self.is_output_transform = list(units), num_heads, is_output_transform


class TransformerFFN(Layer):
    """TransformerFFN (common_layers.py:95-160): dense1 + relu -> dropout -> dense2.

    bf16, d_model 256 (K.ffn_supported): ONE launch per direction -- nst_ffn_fwd computes both products, both dropouts and
    the wrapper's residual add with the hidden tile staying on chip between the products (it is written once, as the saved
    activation); nst_ffn_bwd does the same for the input gradient.  The two weight gradients remain reductions over the
    rows on the weight-gradient stream.
    Otherwise: two GEMMs with fused epilogues (bias+relu+dropout; bias + the wrapper's dropout + residual); backward
    recovers relu'/dropout from the saved hidden activation (h > 0)."""

    def __init__(self, rt, name, hidden_size, filter_size, dropout_rate, gen):
        super().__init__(rt, name)
        self.dense1 = Dense(rt, name + "/dense1", hidden_size, filter_size, gen)
        self.dense2 = Dense(rt, name + "/dense2", filter_size, hidden_size, gen)
        self.rate = dropout_rate
        self.site = self._site()
        self.fused = _FFN_FUSED and K.ffn_supported(hidden_size, filter_size, rt.dtype)
        if self.fused:
            self._w1t = rt.store.add_transposed(self.dense1.kernel)
            self._w2t = rt.store.add_transposed(self.dense2.kernel)

    def forward(self, x, is_training, epilogue=None):
        p = self.rate if is_training else 0.0
        epi = dict(epilogue or {})
        defer_ln = epi.pop("defer_ln", False)
        use_fused = self.fused and x.shape[0] >= _FFN_FUSED_MIN_ROWS and x.is_contiguous() \
            and not (set(epi) - {"residual", "dropout_p", "seed", "stream_id"})
        if self.fused and defer_ln and _ROW_FUSION and is_training and x.is_contiguous() \
                and not (set(epi) - {"dropout_p", "seed", "stream_id"}) and K.ffn_ln_supported(x.shape[0], x.shape[1], self.dense1.out_dim):
            # launched by the next LayerNorm (or on its own: DeferredFfn.plain).  Also below _FFN_FUSED_MIN_ROWS: the decoder's
            # 9 600 rows take the pair kernel with the hidden dimension split over workgroups (nst_ffn_ln_supported == 2)
            return DeferredFfn(self, x, p, epi)
        if use_fused:
            y, h, bits = K.ffn_fwd(x, self._w1t.t, self.dense1.bias.data, self._w2t.t, self.dense2.bias.data,
                                   residual=epi.get("residual"), hidden_p=p, hidden_seed=self.rt.step_seed,
                                   hidden_site=self.site, out_p=epi.get("dropout_p", 0.0), out_seed=epi.get("seed", 0),
                                   out_site=epi.get("stream_id", 0), save_gate_bits=bool(is_training))
        else:
            bits = None
            h = self.dense1.forward(x, relu=True, dropout_p=p, seed=self.rt.step_seed, stream_id=self.site)
            y = self.dense2.forward(h, defer_ln=defer_ln, **epi)
        if is_training:
            self._saved = (x, h, p, bits)
        return y

    def backward(self, dz, residual=None, ln_bwd=None):
        """residual (post-norm wrapper): added to the returned input gradient in the last GEMM's epilogue.
        ln_bwd (pre-norm wrapper, LnBackward): its LayerNorm backward may ride on dense1's input gradient (two-GEMM path)."""
        x, h, p, bits = self._saved
        self._saved = None
        self.dense2.backward_params(h, dz)
        if self.fused and _FFN_FUSED_BWD and ln_bwd is not None and _ROW_FUSION and bits is not None and residual is None \
                and dz.is_contiguous() and ln_bwd.eligible() and K.ffn_ln_supported(dz.shape[0], dz.shape[1], self.dense1.out_dim):
            dx, dh = ln_bwd.run_ffn(dz, h, self.dense2.kernel.compute, self.dense1.kernel.compute, p, bits)
            self.dense1.backward_params(x, dh)
            return dx
        if self.fused and _FFN_FUSED_BWD and dz.shape[0] >= _FFN_FUSED_MIN_ROWS and dz.is_contiguous():
            dx, dh = K.ffn_bwd(dz, h, self.dense2.kernel.compute, self.dense1.kernel.compute, hidden_p=p, residual=residual,
                               gate_bits=bits)
            self.dense1.backward_params(x, dh)
            return dx
        dh = self.dense2.backward_input(dz, gate_src=h, gate_scale=K.dropout_inv_keep(p))
        self.dense1.backward_params(x, dh)
        return self.dense1.backward_input(dh, ln_bwd=ln_bwd, **({} if residual is None else {"residual": residual}))


class PrePostProcessingWrapper(Layer):
    """PrePostProcessingWrapper (common_layers.py:73-92).
    pre-norm  (:73-85):  inputs + dropout(layer(LN(inputs)))
    post-norm (:86-92):  LN(inputs + dropout(layer(inputs)))
    Either way the dropout and the residual add are fused into the wrapped layer's last GEMM epilogue.  In the pre-norm
    backward the residual gradient is fused into the LayerNorm backward kernel; in the post-norm backward it is fused
    into the epilogue of the wrapped layer's last input-gradient GEMM."""

    def __init__(self, rt, name, layer, dim, dropout_rate, epsilon, pre_norm=True):
        super().__init__(rt, name)
        self.layer = layer
        self.norm = LayerNorm(rt, name + "/ln", dim, epsilon)
        self.rate = dropout_rate
        self.pre_norm = pre_norm
        self.site = self._site()

    def forward(self, x, is_training, **kwargs):
        p = self.rate if is_training else 0.0
        self._p = p
        if isinstance(x, ResidualStream):       # pre-norm bf16 stacks: float32 residual stream (see ResidualStream)
            assert self.pre_norm
            y, xs = self.norm.forward_stream(x, save=is_training)
            delta = self.layer.forward(y, is_training=is_training,
                                       epilogue=dict(dropout_p=p, seed=self.rt.step_seed, stream_id=self.site, defer_ln=True), **kwargs)
            return ResidualStream(x.x if xs is None else xs, delta)
        epi = dict(residual=x, dropout_p=p, seed=self.rt.step_seed, stream_id=self.site)
        if not self.pre_norm:
            s = self.layer.forward(x, is_training=is_training, epilogue=epi, **kwargs)
            return self.norm.forward(s, save=is_training)
        y = self.norm.forward(x, save=is_training)
        return self.layer.forward(y, is_training=is_training, epilogue=epi, **kwargs)

    def drop_rate(self):
        return self._p

    def backward(self, dy, consumer=None):
        """consumer: the dropout site that meets the returned gradient next -- on the pre-norm residual chain only; a
        post-norm wrapper hands its gradient to the previous wrapper's LayerNorm, no mask in between."""
        if not self.pre_norm:
            ds = self.norm.backward(dy, consumer=self)     # d(inputs + dropout(layer)); its masked copy rides along
            dz = dropped_grad(self.rt, ds, self._p, self.site)
            out = self.layer.backward(dz, residual=ds)     # layer'(dz) + ds in the last dgrad epilogue
            self.rt.sublayer_boundary()
            return out
        dz = dropped_grad(self.rt, dy, self._p, self.site)
        ln = LnBackward(self.norm, dy, consumer) if _ROW_FUSION else None
        dn = self.layer.backward(dz, ln_bwd=ln) if ln is not None and ln.eligible() else self.layer.backward(dz)
        self.rt.sublayer_boundary()
        if ln is not None and ln.done:      # the layer's last input-gradient product carried the LayerNorm backward
            return dn
        return self.norm.backward(dn, dres=dy, consumer=consumer)


class PositionEmbeddingWrapper(Layer):
    """PositionEmbeddingWrapper, timing="sinusoids" (common_layers.py:415-434): emb*sqrt(d) + signal.
    The wrapped embedding layer applies scale+signal inside its own kernel (GEMM epilogue for the audio
    front end, gather kernel for word embeddings); mode="linear" bypasses the timing like the reference."""

    def __init__(self, rt, name, embedding_layer, timing="sinusoids"):
        super().__init__(rt, name)
        assert timing in (None, "sinusoids"), f"Unknown position embedding type: \"{timing}\""
        self.embedding_layer, self.timing = embedding_layer, timing

    @property
    def embedding_dim(self):
        return self.embedding_layer.embedding_dim

    def forward(self, inputs, mode="embedding", **kw):
        if mode != "embedding":
            return self.embedding_layer.forward(inputs, mode=mode, **kw)
        return self.embedding_layer.forward(inputs, timing=self.timing, **kw)

    def backward(self, dy, mode="embedding"):
        return self.embedding_layer.backward(dy, mode=mode)
