"""MultiHeadAttention / MultiHeadSelfAttention (neurst/layers/attentions/multi_head_attention.py:21-290): the training
path, and the incremental decoding path with a cache (`compute_qkv` cache branches, :151-174 and :254-290).

decode : `cache` is a dict owned by the decoder.  Cross attention projects the memory ONCE (cache["kv"]); self
         attention appends the new position's k|v to preallocated [B', Tmax, d] buffers (the reference concatenates)
         and attends over the filled prefix -- the same fused attention kernel with Tq = 1.

forward : packed projection GEMM (bias fused) -> fused flash attention kernel reading q|k|v as strided column
          views of the projection output (no split / transpose copies) -> output projection GEMM whose epilogue
          carries the wrapper's dropout + residual.
backward: hand scheduled; dq|dk|dv are written by the attention backward kernels directly into one packed
          buffer that feeds the projection wgrad / dgrad GEMMs.
"""
import os

import torch

from neurst_amd import kernels as K
from neurst_amd.layers.common_layers import Layer, MultiHeadDenseLayer


_ROWDOT_ROWS = os.environ.get("NST_ROW_FUSION", "1") != "0"    # nst_gemm_rowdot256 where it wins (see _output_backward_input)


class MultiHeadAttention(Layer):
    """Cross attention: q from `query`, k|v from `memory` (variables q_transform, kv_transform, output_transform)."""

    def __init__(self, rt, name, num_heads, num_units, attention_dropout_rate, gen, attention_type="dot_product",
                 output_depth=None, input_depth=None, memory_depth=None):
        super().__init__(rt, name)
        if attention_type != "dot_product":
            raise NotImplementedError(f"att_fn for \"{attention_type}\" not implemented.")
        if num_units % num_heads != 0:
            raise ValueError("query depth ({}) must be divisible by the number of "
                             "attention heads ({}).".format(num_units, num_heads))
        self.num_heads, self.num_units = num_heads, num_units
        self.dh = num_units // num_heads
        self.rate = attention_dropout_rate
        self.site = self._site()
        self.input_depth, self.memory_depth = input_depth or num_units, memory_depth or input_depth or num_units
        self.output_depth = output_depth or num_units
        self._build_projections(gen, self.output_depth)

    def _build_projections(self, gen, output_depth):
        d, H = self.num_units, self.num_heads
        # creation order follows the reference: output_transform in __init__, the others in build()
        self.output_transform = MultiHeadDenseLayer(self.rt, self.name + "/output_transform", d, output_depth, H, gen,
                                                    is_output_transform=True)
        self.q_transform = MultiHeadDenseLayer(self.rt, self.name + "/q_transform", self.input_depth, d, H, gen)
        self.kv_transform = MultiHeadDenseLayer(self.rt, self.name + "/kv_transform", self.memory_depth, [d, d], H, gen)
        # reduces over the ENCODER's rows (3 x the decoder's): long tiles that do not fit next to the encoder stack's 240 in one
        # round of the grouped launch -- split-K path on the weight-gradient stream, like the front dense layer
        self.kv_transform.wgrad_grouped = False

    def forward(self, query, memory, B, Tq, Tk, memory_bias=None, is_training=True, epilogue=None, cache=None, lagging=None):
        """query [B*Tq, d], memory [B*Tk, d]; memory_bias [B,Tk] f32 (padding*FLOAT_MIN) or None.
        cache (decoding only): the projected memory is computed at the first step and reused.
        lagging (wait-k, transformer_decoder.py:76-92 + layer_utils.py:56-78): query i only sees memory positions
        j <= i + lagging - 1 -- the kernel's causal mask shifted by lagging - 1, on top of the padding bias."""
        d, H, dh = self.num_units, self.num_heads, self.dh
        p = self.rate if is_training else 0.0
        q = self.q_transform.forward(query)
        if cache is not None and "len" in cache:
            # streaming memory (memorize()): the projected memory lives in a preallocated [B, Tmax, 2d] buffer
            Tk = cache["len"]
            kv = cache["kv"]
            kv3 = kv[:, :Tk]
        else:
            if cache is not None:
                if "kv" not in cache:
                    cache["kv"] = self.kv_transform.forward(memory)
                kv = cache["kv"]
            elif getattr(self, "_kv_pre", None) is not None:
                # the decoder projected the memory for ALL its layers in one GEMM over the packed kv_transform kernels
                # (TransformerDecoder._project_memory): this layer's k|v are a column block of that output (row stride = all blocks)
                kv, self._kv_pre = self._kv_pre, None
            else:
                kv = self.kv_transform.forward(memory)
            kv3 = kv.view(B, Tk, 2 * d)
        q3 = q.view(B, Tq, d)
        lag = None if lagging is None else max(int(lagging) - 1, 0)
        ctx, lse, dmask = K.attention_fwd(q3, kv3[..., :d], kv3[..., d:], H, dh, key_bias=memory_bias, causal=lag is not None,
                                   causal_offset=lag or 0, dropout_p=p, seed=self.rt.step_seed, stream_id=self.site)
        out = self.output_transform.forward(ctx.view(B * Tq, d), **(epilogue or {}))
        if is_training:
            self._saved = (query, memory, q, kv, ctx, (lse, dmask), memory_bias, B, Tq, Tk, p, lag)
        return out

    def memorize(self, memory_chunk, cache, B, n):
        """Streaming input (TransformerDecoderLayer.memorize_memory + the concat of update_incremental_cache,
        transformer_decoder.py:159-168): projects `n` new memory positions [B*n, d] and appends them to the layer's
        preallocated key|value buffer cache["kv"] [B, Tmax, 2d]; cache["len"] counts the filled positions."""
        t = cache["len"]
        if t + n > cache["kv"].shape[1]:
            raise RuntimeError(f"memory cache of {cache['kv'].shape[1]} positions is full")
        cache["kv"][:, t:t + n] = self.kv_transform.forward(memory_chunk).view(B, n, 2 * self.num_units)
        cache["len"] = t + n

    def _output_backward_input(self, dz, ctx2, lse, Tq):
        """d(context) = dz . Wo^T; with 64-wide heads on the bf16 path the same GEMM epilogue also leaves
        delta = rowsum(d(context) o context) per head for the attention backward (no separate pass over both tensors)."""
        if self.dh == 64 and K.rowdot_supported(dz, self.num_units):
            delta = torch.empty_like(lse)
            ot = self.output_transform
            # the decoder's row count (fewer rows than fill the chip with 128 x 128 tiles): the whole-row kernel's 48-row
            # workgroups take the product (7.3 against 8.7 us at 9 600 rows; at the encoder's 28 800 the stream kernel wins)
            if _ROWDOT_ROWS and self.num_units == 256 and ctx2.is_contiguous() and dz.shape[0] < 64 * 224 \
                    and K.rowgemm_supported(dz, ot.in_dim, ot.out_dim):
                return K.gemm_rowdot256(dz, ot.kernel.compute, rowdot=(ctx2, delta, Tq), trans_b=True), delta
            return ot.backward_input(dz, rowdot=(ctx2, delta, Tq)), delta
        return self.output_transform.backward_input(dz), None

    def backward(self, dz, dmemory=None, dmemory_accumulate=False, residual=None, ln_bwd=None):
        """Returns d(query) (+ residual, post-norm wrapper); d(memory) is written (or accumulated) into `dmemory`
        [B*Tk, d]."""
        query, memory, q, kv, ctx, (lse, dmask), bias, B, Tq, Tk, p, lag = self._saved
        self._saved = None
        d, H, dh = self.num_units, self.num_heads, self.dh
        ctx2 = ctx.view(B * Tq, d)
        self.output_transform.backward_params(ctx2, dz)
        dctx, delta = self._output_backward_input(dz, ctx2, lse, Tq)
        dq = torch.empty_like(q)
        grouped = getattr(self, "_dkv_out", None) is not None    # d(k|v) goes into the decoder's packed buffer; the decoder
        dkv = self._dkv_out if grouped else torch.empty_like(kv)  # turns all layers' blocks into d(memory) with ONE GEMM
        self._dkv_out = None
        q3, kv3, dq3, dkv3 = q.view(B, Tq, d), kv.view(B, Tk, 2 * d), dq.view(B, Tq, d), dkv.view(B, Tk, 2 * d)
        K.attention_bwd(q3, kv3[..., :d], kv3[..., d:], ctx, dctx.view(B, Tq, d), lse, dq3, dkv3[..., :d],
                        dkv3[..., d:], H, dh, key_bias=bias, causal=lag is not None, causal_offset=lag or 0, dropout_p=p,
                        seed=self.rt.step_seed, stream_id=self.site, drop_mask=dmask, delta=delta)
        self.q_transform.backward_params(query, dq)
        self.kv_transform.backward_params(memory, dkv)
        if dmemory is not None and not grouped:
            self.kv_transform.backward_input(dkv, out=dmemory, accumulate=dmemory_accumulate)
        return self.q_transform.backward_input(dq, ln_bwd=ln_bwd, **({} if residual is None else {"residual": residual}))


class MultiHeadSelfAttention(MultiHeadAttention):
    """Self attention with one packed qkv_transform (multi_head_attention.py:226-290)."""

    def _build_projections(self, gen, output_depth):
        d, H = self.num_units, self.num_heads
        self.output_transform = MultiHeadDenseLayer(self.rt, self.name + "/output_transform", d, output_depth, H, gen,
                                                    is_output_transform=True)
        self.qkv_transform = MultiHeadDenseLayer(self.rt, self.name + "/qkv_transform", self.input_depth, [d, d, d], H, gen)

    def forward(self, x, B, T, bias=None, causal=False, is_training=True, epilogue=None, cache=None):
        """x [B*T, d]; bias [B,T] f32 key-padding bias or None; causal=True is the decoder's lower-triangle bias.
        cache (inference only): {"keys", "values": [B, Tmax, d] buffers, "len": filled positions}; T == 1 for a decoding
        step, T >= 1 for a chunk of the streaming encoder."""
        d, H, dh = self.num_units, self.num_heads, self.dh
        p = self.rate if is_training else 0.0
        qkv = self.qkv_transform.forward(x)
        if cache is not None:
            # T == 1: one decoding step.  T > 1: a chunk of a streaming (monotonic) encoder -- position i of the chunk
            # sees the t cached positions and the chunk up to itself: the kernel's causal mask shifted by t
            # (transformer_encoder.py:138-175 builds lower_triangle_attention_bias(t + T)[:, :, -T:]).
            assert not is_training
            t = cache["len"]
            if t + T > cache["keys"].shape[1]:
                raise RuntimeError(f"attention cache of {cache['keys'].shape[1]} positions is full")
            v3 = qkv.view(B, T, 3 * d)
            cache["keys"][:, t:t + T] = v3[..., d:2 * d]
            cache["values"][:, t:t + T] = v3[..., 2 * d:]
            cache["len"] = t + T
            ctx, _, _ = K.attention_fwd(v3[..., :d], cache["keys"][:, :t + T], cache["values"][:, :t + T], H, dh,
                                        key_bias=None, causal=T > 1, causal_offset=t if T > 1 else 0)
            return self.output_transform.forward(ctx.view(B * T, d), **(epilogue or {}))
        v3 = qkv.view(B, T, 3 * d)
        ctx, lse, dmask = K.attention_fwd(v3[..., :d], v3[..., d:2 * d], v3[..., 2 * d:], H, dh, key_bias=bias, causal=causal,
                                   dropout_p=p, seed=self.rt.step_seed, stream_id=self.site)
        out = self.output_transform.forward(ctx.view(B * T, d), **(epilogue or {}))
        if is_training:
            self._saved = (x, qkv, ctx, (lse, dmask), bias, causal, B, T, p)
        return out

    def backward(self, dz, residual=None, ln_bwd=None):
        x, qkv, ctx, (lse, dmask), bias, causal, B, T, p = self._saved
        self._saved = None
        d, H, dh = self.num_units, self.num_heads, self.dh
        self.output_transform.backward_params(ctx.view(B * T, d), dz)
        dctx, delta = self._output_backward_input(dz, ctx.view(B * T, d), lse, T)
        dqkv = torch.empty_like(qkv)
        v3, g3 = qkv.view(B, T, 3 * d), dqkv.view(B, T, 3 * d)
        K.attention_bwd(v3[..., :d], v3[..., d:2 * d], v3[..., 2 * d:], ctx, dctx.view(B, T, d), lse, g3[..., :d],
                        g3[..., d:2 * d], g3[..., 2 * d:], H, dh, key_bias=bias, causal=causal, dropout_p=p,
                        seed=self.rt.step_seed, stream_id=self.site, drop_mask=dmask, delta=delta)
        self.qkv_transform.backward_params(x, dqkv)
        return self.qkv_transform.backward_input(dqkv, ln_bwd=ln_bwd, **({} if residual is None else {"residual": residual}))
