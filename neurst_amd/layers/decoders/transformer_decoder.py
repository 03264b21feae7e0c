"""TransformerDecoder (neurst/layers/decoders/transformer_decoder.py:23-228): the training branch
(cache["decoding_states"] is None) and incremental decoding with per-layer caches (wait-k lagging is not built)."""

import torch

from neurst_amd import kernels as K
from neurst_amd.layers import layer_utils
from neurst_amd.layers.common_layers import LayerNorm, ResidualStream, dropped_grad
from neurst_amd.layers.decoders.decoder import Decoder, register_decoder
from neurst_amd.layers.transformer_layers import TransformerDecoderLayer


KV_GROUP = True     # one GEMM for the cross-attention k|v projections of all layers (see TransformerDecoder.build)


@register_decoder
class TransformerDecoder(Decoder):
    def __init__(self, num_layers, hidden_size, num_attention_heads, filter_size, ffn_activation="relu",
                 attention_dropout_rate=0., attention_type="dot_product", ffn_dropout_rate=0.,
                 layer_postprocess_dropout_rate=0., layer_postprocess_epsilon=1e-6, post_normalize=False,
                 no_cross_attn_layer_list=None, name=None):
        super().__init__(num_layers=num_layers, hidden_size=hidden_size, num_attention_heads=num_attention_heads,
                         filter_size=filter_size, ffn_activation=ffn_activation,
                         attention_dropout_rate=attention_dropout_rate, attention_type=attention_type,
                         ffn_dropout_rate=ffn_dropout_rate,
                         layer_postprocess_dropout_rate=layer_postprocess_dropout_rate,
                         layer_postprocess_epsilon=layer_postprocess_epsilon, post_normalize=post_normalize,
                         no_cross_attn_layer_list=no_cross_attn_layer_list or [])
        self.name = name or self.__class__.__name__

    def build(self, rt, gen):
        p = self._params
        self.rt = rt
        self._stacking_layers = [
            TransformerDecoderLayer(rt, f"{self.name}/layer_{i}", p["hidden_size"], p["num_attention_heads"],
                                    p["filter_size"], gen, p["ffn_activation"], p["attention_dropout_rate"],
                                    p["attention_type"], p["ffn_dropout_rate"], p["layer_postprocess_dropout_rate"],
                                    p["layer_postprocess_epsilon"], p["post_normalize"],
                                    with_cross_attention=(i not in p["no_cross_attn_layer_list"]))
            for i in range(p["num_layers"])]
        # post-norm stacks end with the last wrapper's LayerNorm: no output_ln (transformer_decoder.py:98-101)
        self._output_norm_layer = None if p["post_normalize"] else LayerNorm(
            rt, f"{self.name}/output_ln", p["hidden_size"], p["layer_postprocess_epsilon"])
        self._site = rt.new_dropout_site()
        self._stream32 = ResidualStream.supported(rt, p["hidden_size"], pre_norm=not p["post_normalize"])
        # Training: every layer's cross attention projects the SAME encoder output with its own kv_transform
        # (transformer_layers.py:213-234, multi_head_attention.py:166-223).  The kernels are kept side by side in a packed
        # copy [d, n * 2d] (refreshed once per optimizer step), so the projection is ONE GEMM whose output the layers read as
        # column blocks, and d(memory) = [d(k|v) of all layers] . packed^T is ONE GEMM with K = n * 2d instead of n
        # accumulating ones.  The weight gradients stay per layer (the data-parallel reducer ships a layer's gradients as soon
        # as that layer's backward is queued).  KV_GROUP = False: per-layer projections (tests).
        self._kv_group, self._kv_atts = None, []
        atts = [l._cross.att for l in self._stacking_layers if l._with_cross_attention]
        if KV_GROUP and len(atts) >= 2 \
                and len({tuple(a.kv_transform.kernel.shape) for a in atts}) == 1:
            self._kv_group = rt.store.add_packed([a.kv_transform.kernel for a in atts], [a.kv_transform.bias for a in atts])
            self._kv_atts = atts
        return self

    def _project_memory(self, mem2):
        """One GEMM for the k|v of every layer; hands each layer its column block."""
        g = self._kv_group
        kv_all = K.gemm(mem2, g.w, mem2.shape[0], g.w.shape[1], g.w.shape[0], bias=g.b)
        for a, c0 in zip(self._kv_atts, g.col0):
            a._kv_pre = kv_all[:, c0:c0 + a.kv_transform.out_dim]
        return kv_all

    def create_decoding_internal_cache(self, encoder_outputs, encoder_inputs_padding, is_inference=False,
                                       decode_padded_length=None):
        """transformer_decoder.py:105-147.  Training: {"decoding_states": None, "memory", "memory_bias"}.  Inference adds
        per-layer decoding states: self-attention key / value buffers of `decode_padded_length` positions (filled one
        position per step) and the cross attention's projected memory (filled at the first step)."""
        cache = dict(decoding_states=None)
        if encoder_inputs_padding is not None:
            cache["memory"] = encoder_outputs
            cache["memory_bias"] = layer_utils.input_padding_to_bias(encoder_inputs_padding)
        if is_inference:
            if decode_padded_length is None:
                raise ValueError("inference needs decode_padded_length (the maximum number of decoding steps)")
            B, d = encoder_outputs.shape[0], self._params["hidden_size"]
            # one buffer for the keys and values of all layers: a beam re-ordering is ONE gather over it
            kv_all = torch.zeros(self._params["num_layers"], 2, B, decode_padded_length, d, dtype=encoder_outputs.dtype,
                                 device=encoder_outputs.device)
            states = {}
            for i in range(self._params["num_layers"]):
                states[f"layer_{i}"] = {"self_attention": {"keys": kv_all[i, 0], "values": kv_all[i, 1], "len": 0},
                                        "encdec_attention": {}}
            cache["decoding_states"] = states
            cache["self_attention_kv"] = kv_all
        return cache

    def update_incremental_cache(self, cache, encoder_outputs, encoder_inputs_padding, max_source_length=1024,
                                 decode_padded_length=256):
        """Streaming input (transformer_decoder.py:149-169): appends newly encoded source positions `encoder_outputs`
        [B, n, d] (+ their padding [B, n]) to the memory, the memory bias and every layer's projected memory.  The
        reference concatenates tensors; here the cache created at the first call preallocates `max_source_length`
        memory positions and `decode_padded_length` target positions and is filled in place."""
        B, n, d = encoder_outputs.shape
        if not cache:
            L = self._params["num_layers"]
            dt, dev = encoder_outputs.dtype, encoder_outputs.device
            cache = {"decoding_states": {}, "memory_len": 0,
                     "memory_buf": torch.zeros(B, max_source_length, d, dtype=dt, device=dev),
                     "memory_bias_buf": torch.zeros(B, max_source_length, dtype=torch.float32, device=dev),
                     "self_attention_kv": torch.zeros(L, 2, B, decode_padded_length, d, dtype=dt, device=dev),
                     "memory_kv": torch.zeros(L, B, max_source_length, 2 * d, dtype=dt, device=dev)}
            for i in range(L):
                kv = cache["self_attention_kv"]
                cache["decoding_states"][f"layer_{i}"] = {
                    "self_attention": {"keys": kv[i, 0], "values": kv[i, 1], "len": 0},
                    "encdec_attention": {"kv": cache["memory_kv"][i], "len": 0}}
        t = cache["memory_len"]
        if t + n > cache["memory_buf"].shape[1]:
            raise RuntimeError(f"memory cache of {cache['memory_buf'].shape[1]} positions is full")
        cache["memory_buf"][:, t:t + n] = encoder_outputs
        cache["memory_bias_buf"][:, t:t + n] = layer_utils.input_padding_to_bias(encoder_inputs_padding)
        cache["memory_len"] = t + n
        cache["memory"] = cache["memory_buf"][:, :t + n]
        cache["memory_bias"] = cache["memory_bias_buf"][:, :t + n]
        chunk = encoder_outputs.reshape(B * n, d)
        for i, layer in enumerate(self._stacking_layers):
            layer.memorize_memory(chunk, cache["decoding_states"][f"layer_{i}"], B, n)
        return cache

    @staticmethod
    def reorder_cache(cache, beam_ids):
        """tf.gather(cache, beam_ids) of the beam search (beam_search.py:409-410): only the self-attention buffers depend on
        the hypothesis; memory, memory_bias and the projected memory are identical for all beams of a sample (beam_ids
        never leave the sample's block), so they stay in place."""
        n = next(iter(cache["decoding_states"].values()))["self_attention"]["len"]
        kv = cache["self_attention_kv"]
        kv[:, :, :, :n] = kv[:, :, :, :n].index_select(2, beam_ids)
        return cache

    def decode_step(self, decoder_inputs, cache, decode_lagging=None):
        """One incremental step: decoder_inputs [B', d] (embedding of the last generated symbols) -> [B', d].
        decode_lagging (wait-k inference, transformer_decoder.py:86-92): the memory positions this step may see."""
        assert cache.get("decoding_states", None) is not None, "create_decoding_internal_cache(is_inference=True) first"
        Bp, d = decoder_inputs.shape
        memory, memory_bias = cache.get("memory", None), cache.get("memory_bias", None)
        Tm = memory.shape[1] if memory is not None else 0
        streaming = "memory_len" in cache   # update_incremental_cache: the layers hold the projected memory already
        mem2 = memory.reshape(Bp * Tm, d) if memory is not None and not streaming else None
        if decode_lagging is not None and memory_bias is not None:
            seen = (torch.arange(Tm, device=memory_bias.device) < int(decode_lagging)).to(memory_bias.dtype)
            memory_bias = torch.minimum(memory_bias, layer_utils.FLOAT_MIN * (1.0 - seen)[None, :])
        if memory_bias is not None and not memory_bias.is_contiguous():
            memory_bias = memory_bias.contiguous()           # the kernel reads a dense [B, Tk] bias
        x = decoder_inputs
        if self._stream32:
            x = ResidualStream(x if x.is_contiguous() else x.contiguous())
        for i, layer in enumerate(self._stacking_layers):
            x = layer.forward(x, Bp, 1, mem2, Tm, memory_bias, is_training=False, cache=cache["decoding_states"][f"layer_{i}"])
        return self._finish(x, False)

    def _finish(self, x, save):
        """Output of the stack from the residual stream / tensor behind the last layer."""
        if isinstance(x, ResidualStream):
            return self._output_norm_layer.forward_stream(x, save=save, want_sum=False)[0]
        return x if self._output_norm_layer is None else self._output_norm_layer.forward(x, save=save)

    def forward(self, decoder_inputs, cache, decode_lagging=None, is_training=True, decode_loop_step=None):
        """decoder_inputs [B,L,d]; cache from create_decoding_internal_cache -> [B,L,d]."""
        if decode_loop_step is not None:
            raise NotImplementedError("static-shape (padded) decoding caches are not built: the cache is preallocated anyway")
        if cache.get("decoding_states", None) is not None:
            return self.decode_step(decoder_inputs.reshape(-1, decoder_inputs.shape[-1]), cache,
                                    decode_lagging=decode_lagging).view(decoder_inputs.shape)
        B, L, d = decoder_inputs.shape
        memory = cache.get("memory", None)
        memory_bias = cache.get("memory_bias", None)
        Tm = memory.shape[1] if memory is not None else 0
        mem2 = memory.reshape(B * Tm, d) if memory is not None else None
        x = decoder_inputs.reshape(B * L, d)
        p = self._params["layer_postprocess_dropout_rate"] if is_training else 0.0
        self._p = p
        if p > 0:
            x = K.scale_posenc_dropout_fwd(x, None, 1, 1.0, p, self.rt.step_seed, self._site)
        self._grouped = self._kv_group is not None and mem2 is not None and is_training
        self._clear_kv_handoff()      # a forward / backward that aborted half way must not leave another batch's k|v behind
        if self._grouped:
            self._project_memory(mem2)
        if self._stream32:
            x = ResidualStream(x if x.is_contiguous() else x.contiguous())
        try:
            for layer in self._stacking_layers:
                x = layer.forward(x, B, L, mem2, Tm, memory_bias, is_training=is_training, lagging=decode_lagging)
        finally:
            for a in getattr(self, "_kv_atts", ()):
                a._kv_pre = None
        out = self._finish(x, is_training)
        self._shapes = (B, L, d, Tm)
        return out.view(B, L, d)

    __call__ = forward

    def _clear_kv_handoff(self):
        """The grouped cross-attention k|v projection reaches the layers through per-layer attributes (_kv_pre: this batch's
        k|v block, _dkv_out: where its gradient goes); each is consumed by its layer.  Cleared here at the start of every
        forward and backward and again behind them, so an exception between hand-over and use cannot leak them into the next
        call (which would attend over another batch's keys, or write d(k|v) into a dead buffer)."""
        for a in getattr(self, "_kv_atts", ()):
            a._kv_pre = None
            a._dkv_out = None

    def backward(self, dout, layer_done=None):
        """Returns (d decoder_inputs [B,L,d], d memory [B,Tm,d]).  layer_done: see TransformerEncoder.backward."""
        try:
            return self._backward(dout, layer_done)
        finally:
            for a in getattr(self, "_kv_atts", ()):
                a._dkv_out = None

    def _backward(self, dout, layer_done=None):
        B, L, d, Tm = self._shapes
        dmemory = torch.empty(B * Tm, d, dtype=dout.dtype, device=dout.device) if Tm else None
        layers = self._stacking_layers
        dx = dout.reshape(B * L, d)
        if self._output_norm_layer is not None:
            dx = self._output_norm_layer.backward(dx, consumer=layers[-1].first_backward_site if layers else self)
        elif not dx.is_contiguous():
            dx = dx.contiguous()
        first = True
        dkv_all = None
        for a in getattr(self, "_kv_atts", ()):
            a._dkv_out = None
        if getattr(self, "_grouped", False) and Tm:
            g = self._kv_group
            dkv_all = torch.empty(B * Tm, g.w.shape[1], dtype=dout.dtype, device=dout.device)
            for a, c0 in zip(self._kv_atts, g.col0):
                a._dkv_out = dkv_all[:, c0:c0 + a.kv_transform.out_dim]
        for i in range(len(layers) - 1, -1, -1):
            dx = layers[i].backward(dx, dmemory, dmemory_accumulate=not first,
                                    consumer=layers[i - 1].first_backward_site if i > 0 else self)
            if layers[i]._with_cross_attention:
                first = False
            if layer_done is not None:
                extra = [self._output_norm_layer.name + "/"] if (i == len(layers) - 1 and self._output_norm_layer is not None) else []
                layer_done([layers[i].name + "/"] + extra)    # output_ln rides with the top layer (see TransformerEncoder)
        dx = dropped_grad(self.rt, dx, self._p, self._site)
        if dkv_all is not None:     # d(memory) of all layers at once: [B*Tm, n*2d] . packed^T (K = n * 2d)
            g = self._kv_group
            K.gemm(dkv_all, g.w, B * Tm, d, g.w.shape[1], trans_b=True, out=dmemory)
            first = False
        if dmemory is not None and first:
            dmemory.zero_()
        return dx.view(B, L, d), (dmemory.view(B, Tm, d) if dmemory is not None else None)

    # the decoder's input dropout as a dropout site (see LayerNorm.backward)
    @property
    def site(self):
        return self._site

    def drop_rate(self):
        return self._p
