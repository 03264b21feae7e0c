"""CPU oracle: a restatement of the NeurST SpeechTransformer training math.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The shipped path (``neurst_amd``) never calls into this module and
fails loudly when ``libneurst_hip.so`` is missing.

Every function restates, in plain torch-CPU ops (fp32 by default, fp64 on
request, autograd for gradients), the arithmetic of one reference function.
Citations are ``file:line`` relative to ``/root/reference``.  Weights are kept
in the reference's TensorFlow variable layout and addressed by the
reference's TF variable names (``tests/neurst/models/transformer_test.py:44-628``)
so that golden vectors and TF checkpoints map 1:1:

  * dense / conv kernels ``[in, out]`` / HWIO, ``x @ kernel``
  * ``qkv_transform/kernel``  ``[in, 3*H*dh]`` columns ``q | k | v``
  * ``output_transform/kernel``  ``[H*dh, out]``
  * ``shared/weights``  ``[V, d]``, logits = ``x @ W.T + bias``

Pinning status (see tests/test_oracle_golden.py):
  * attention / encoder / decoder / sinusoid position embedding / full
    enc-dec logits are pinned by the literal golden vectors of the
    reference's own tests (tests/golden/*.npz, extracted by
    tests/golden/make_golden.py).
  * the conv front-end is pinned against the reference's own
    ``neurst_pt`` implementation executed under an import shim
    (tests/golden/make_golden.py::gen_neurst_pt_frontend).
  * criterion, gradients, Adam/Noam and data-parallel averaging have no
    reference test: PARITY UNPINNED for those; they are covered by
    closed-form / finite-difference known-answer tests only.
"""
import math

import torch
import torch.nn.functional as F

FLOAT_MIN = -1.0e9  # neurst/utils/compat.py:24


# --------------------------------------------------------------------------
# elementary layers
# --------------------------------------------------------------------------
def layer_norm(x, gamma, beta, eps):
    """tf.keras.layers.LayerNormalization(epsilon=eps, dtype=float32) over the
    last axis (neurst/layers/common_layers.py:64-65).  Biased variance."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def dropout(x, rate, is_training, generator=None, tag=None):
    """tf.nn.dropout (inverted dropout).  `generator` is a torch.Generator (masks drawn here), or -- to compare against a
    path whose masks are known -- an object with mask_for(tag, shape, rate) returning the keep multipliers (0 or
    1/keep) of the dropout site `tag` (the variable-scope name of the layer that owns the site)."""
    if not is_training or rate == 0.0:
        return x
    if hasattr(generator, "mask_for"):
        return x * generator.mask_for(tag, tuple(x.shape), rate).to(x.dtype)
    keep = (torch.rand(x.shape, generator=generator, dtype=x.dtype) >= rate).to(x.dtype)
    return x * keep / (1.0 - rate)


def multi_head_dense(x, kernel, bias, num_heads, output_units, is_output_transform=False):
    """MultiHeadDenseLayer.call (neurst/layers/common_layers.py:249-295).

    non-output: x [B,T,in] @ kernel [in, sum(units)] + bias, split on the last
    axis into ``output_units`` blocks, each reshaped [B,T,H,units/H].
    output: x [B,T,H,dh] with kernel [H*dh, out] viewed [H,dh,out].
    """
    if is_output_transform:
        b, t = x.shape[0], x.shape[1]
        out = x.reshape(b, t, -1) @ kernel
        if bias is not None:
            out = out + bias
        return out
    out = x @ kernel
    if bias is not None:
        out = out + bias
    units = output_units if isinstance(output_units, (list, tuple)) else [output_units]
    outs = torch.split(out, list(units), dim=-1)
    outs = [o.reshape(*o.shape[:-1], num_heads, u // num_heads) for o, u in zip(outs, units)]
    return outs if isinstance(output_units, (list, tuple)) else outs[0]


def attention_core(q, k, v, bias, attention_dropout_rate=0.0, is_training=False, generator=None, tag=None):
    """att_fn + weighted sum (neurst/layers/attentions/multi_head_attention.py:124-164, 203-215).

    q [B,F,H,dh] (NOT yet scaled), k,v [B,T,H,dh]; bias [B,T] (key padding),
    [B,F,T] or [1,1,F,T].  Returns [B,F,H,dh].
    """
    dh = q.shape[-1]
    q = q * (dh ** -0.5)                                   # :203
    logits = torch.einsum("bthd,bfhd->bhft", k, q)         # :145
    if bias is not None:                                   # :147-156
        if bias.dim() == 2:
            bias = bias[:, None, None, :]
        elif bias.dim() == 3:
            bias = bias[:, None]
        logits = logits + bias
    weights = torch.softmax(logits, dim=-1)                # :160
    weights = dropout(weights, attention_dropout_rate, is_training, generator, tag)  # :207-208; [B, H, Tq, Tk]
    return torch.einsum("bhft,bthd->bfhd", weights, v)     # :215


def self_attention(x, W, prefix, num_heads, bias, rate=0.0, is_training=False, generator=None):
    """MultiHeadSelfAttention.call (multi_head_attention.py:226-290)."""
    d_in = W[prefix + "/qkv_transform/kernel"].shape[1] // 3
    q, k, v = multi_head_dense(x, W[prefix + "/qkv_transform/kernel"], W.get(prefix + "/qkv_transform/bias"),
                               num_heads, [d_in, d_in, d_in])
    ctx = attention_core(q, k, v, bias, rate, is_training, generator, prefix)
    return multi_head_dense(ctx, W[prefix + "/output_transform/kernel"], W.get(prefix + "/output_transform/bias"),
                            num_heads, None, is_output_transform=True)


def cross_attention(x, memory, W, prefix, num_heads, memory_bias, rate=0.0, is_training=False, generator=None):
    """MultiHeadAttention.call (multi_head_attention.py:166-223)."""
    dq = W[prefix + "/q_transform/kernel"].shape[1]
    dkv = W[prefix + "/kv_transform/kernel"].shape[1] // 2
    q = multi_head_dense(x, W[prefix + "/q_transform/kernel"], W.get(prefix + "/q_transform/bias"), num_heads, dq)
    k, v = multi_head_dense(memory, W[prefix + "/kv_transform/kernel"], W.get(prefix + "/kv_transform/bias"),
                            num_heads, [dkv, dkv])
    ctx = attention_core(q, k, v, memory_bias, rate, is_training, generator, prefix)
    return multi_head_dense(ctx, W[prefix + "/output_transform/kernel"], W.get(prefix + "/output_transform/bias"),
                            num_heads, None, is_output_transform=True)


_RELU_GATES = None


class relu_gates(object):
    """Diagnostic context (tests only): inside it the oracle's ReLUs use the 0/1 gates of ANOTHER path instead of the sign
    of their own pre-activation -- `provider.gate_for(tag, shape)` returns the gate tensor of the ReLU owned by the
    variable scope `tag` (an FFN's scope, or "<modality>/conv1|conv2"), or None to keep the oracle's own.  Running the
    oracle under a reduced-precision path's gates separates that path's rounding error from the discrete ReLU flips of
    near-zero pre-activations."""

    def __init__(self, provider):
        self.provider = provider

    def __enter__(self):
        global _RELU_GATES
        self._old, _RELU_GATES = _RELU_GATES, self.provider
        return self

    def __exit__(self, *exc):
        global _RELU_GATES
        _RELU_GATES = self._old


def relu(x, tag=None):
    if _RELU_GATES is not None and tag is not None:
        g = _RELU_GATES.gate_for(tag, tuple(x.shape))
        if g is not None:
            return x * g.to(x.dtype)
    return F.relu(x)


def ffn(x, W, prefix, rate=0.0, is_training=False, generator=None):
    """TransformerFFN.call (neurst/layers/common_layers.py:145-160), relu."""
    h = relu(x @ W[prefix + "/dense1/kernel"] + W[prefix + "/dense1/bias"], prefix)
    h = dropout(h, rate, is_training, generator, prefix)
    return h @ W[prefix + "/dense2/kernel"] + W[prefix + "/dense2/bias"]


def prepost(x, fn, W, prefix, eps, rate=0.0, is_training=False, generator=None, pre_norm=True):
    """PrePostProcessingWrapper.call (common_layers.py:73-92).  pre-norm: LN -> layer -> dropout -> residual;
    post-norm (pre_norm=False, :86-92): layer -> dropout -> residual -> LN."""
    if not pre_norm:
        y = dropout(fn(x), rate, is_training, generator, prefix)
        return layer_norm(x + y, W[prefix + "/ln/gamma"], W[prefix + "/ln/beta"], eps)
    y = layer_norm(x, W[prefix + "/ln/gamma"], W[prefix + "/ln/beta"], eps)
    y = fn(y)
    y = dropout(y, rate, is_training, generator, prefix)
    return x + y


def _get(W, name, default):
    return W[name] if name in W else default


def fill_default_biases(W, cfg=None):
    """Golden vectors pin kernels only; biases stay at zero and LN at
    gamma=1 / beta=0.  Fill whatever is missing accordingly."""
    W = dict(W)
    for name in list(W.keys()):
        if name.endswith("/kernel"):
            b = name[:-len("kernel")] + "bias"
            if b not in W:
                W[b] = torch.zeros(W[name].shape[-1], dtype=W[name].dtype)
    return W


def _ensure_ln(W, prefix, d, dtype):
    if prefix + "/gamma" not in W:
        W[prefix + "/gamma"] = torch.ones(d, dtype=dtype)
        W[prefix + "/beta"] = torch.zeros(d, dtype=dtype)


# --------------------------------------------------------------------------
# encoder / decoder stacks
# --------------------------------------------------------------------------
def input_padding_to_bias(padding):
    """neurst/layers/layer_utils.py:19-32."""
    return padding * FLOAT_MIN


def lower_triangle_attention_bias(length, dtype=torch.float32):
    """neurst/layers/layer_utils.py:35-53 -> [1,1,L,L]."""
    tril = torch.tril(torch.ones(length, length, dtype=dtype))
    return (FLOAT_MIN * (1.0 - tril)).reshape(1, 1, length, length)


def waitk_attention_bias(memory_length, waitk_lagging, query_length, dtype=torch.float32):
    """neurst/layers/layer_utils.py:56-78, training form: [query_length, memory_length], 0 where key j <= query i + lagging - 1
    (tf.linalg.band_part(ones, -1, min(lagging - 1, memory_length))), FLOAT_MIN elsewhere."""
    upper = min(waitk_lagging - 1, memory_length)
    i = torch.arange(query_length)[:, None]
    j = torch.arange(memory_length)[None, :]
    keep = (j - i <= upper).to(dtype)
    return FLOAT_MIN * (1.0 - keep)


def transformer_encoder(x, padding, W, scope, num_layers, num_heads, eps=1e-6,
                        att_rate=0.0, ffn_rate=0.0, post_rate=0.0, is_training=False, generator=None, monotonic=False,
                        post_normalize=False):
    """TransformerEncoder.call (neurst/layers/encoders/transformer_encoder.py:104-136)
    with TransformerEncoderLayer.call (transformer_layers.py:90-98); attention_monotonic :121-123;
    post_normalize: post-norm wrappers (transformer_layers.py:73,85) and no output_ln (:97-100, 131-134)."""
    pre = not post_normalize
    bias = input_padding_to_bias(padding)
    if monotonic:
        bias = torch.minimum(bias[:, None, None, :], lower_triangle_attention_bias(x.shape[1], x.dtype))
    x = dropout(x, post_rate, is_training, generator, scope)
    for i in range(num_layers):
        p = f"{scope}/layer_{i}"
        _ensure_ln(W, p + "/self_attention_prepost_wrapper/ln", x.shape[-1], x.dtype)
        _ensure_ln(W, p + "/ffn_prepost_wrapper/ln", x.shape[-1], x.dtype)
        x = prepost(x, lambda y: self_attention(y, W, p + "/self_attention_prepost_wrapper/self_attention",
                                                num_heads, bias, att_rate, is_training, generator),
                    W, p + "/self_attention_prepost_wrapper", eps, post_rate, is_training, generator, pre)
        x = prepost(x, lambda y: ffn(y, W, p + "/ffn_prepost_wrapper/ffn", ffn_rate, is_training, generator),
                    W, p + "/ffn_prepost_wrapper", eps, post_rate, is_training, generator, pre)
    if post_normalize:
        return x
    _ensure_ln(W, scope + "/output_ln", x.shape[-1], x.dtype)
    return layer_norm(x, W[scope + "/output_ln/gamma"], W[scope + "/output_ln/beta"], eps)


def transformer_decoder(x, memory, memory_padding, W, scope, num_layers, num_heads, eps=1e-6,
                        att_rate=0.0, ffn_rate=0.0, post_rate=0.0, is_training=False, generator=None, decode_lagging=None,
                        post_normalize=False):
    """TransformerDecoder.call, training branch (neurst/layers/decoders/transformer_decoder.py:171-228)
    with TransformerDecoderLayer.call (transformer_layers.py:213-234).  Cross-attention
    K/V are projected from ``memory`` directly (the encoder's output_ln output).  post_normalize as in the encoder
    (transformer_decoder.py:98-101, 224-227)."""
    pre = not post_normalize
    memory_bias = input_padding_to_bias(memory_padding) if memory_padding is not None else None
    if memory_bias is not None and isinstance(decode_lagging, (list, tuple, torch.Tensor)):
        # streaming decode: target position i was decoded when only decode_lagging[i] memory positions were visible
        vis = torch.as_tensor(decode_lagging).reshape(-1, 1)
        keep = (torch.arange(memory_bias.shape[1])[None, :] < vis).to(x.dtype)
        memory_bias = torch.minimum(memory_bias[:, None, :], (FLOAT_MIN * (1.0 - keep))[None, :, :])[:, None, :, :]
    elif memory_bias is not None and decode_lagging is not None:  # transformer_decoder.py:76-85 (3-d inputs)
        memory_bias = torch.minimum(memory_bias[:, None, :], waitk_attention_bias(
            memory_bias.shape[1], decode_lagging, x.shape[1], x.dtype)[None, :, :])[:, None, :, :]
    causal = lower_triangle_attention_bias(x.shape[1], x.dtype)
    x = dropout(x, post_rate, is_training, generator, scope)
    for i in range(num_layers):
        p = f"{scope}/layer_{i}"
        for w in ("self_attention_prepost_wrapper", "encdec_attention_prepost_wrapper", "ffn_prepost_wrapper"):
            _ensure_ln(W, f"{p}/{w}/ln", x.shape[-1], x.dtype)
        x = prepost(x, lambda y: self_attention(y, W, p + "/self_attention_prepost_wrapper/self_attention",
                                                num_heads, causal, att_rate, is_training, generator),
                    W, p + "/self_attention_prepost_wrapper", eps, post_rate, is_training, generator, pre)
        x = prepost(x, lambda y: cross_attention(y, memory, W, p + "/encdec_attention_prepost_wrapper/encdec_attention",
                                                 num_heads, memory_bias, att_rate, is_training, generator),
                    W, p + "/encdec_attention_prepost_wrapper", eps, post_rate, is_training, generator, pre)
        x = prepost(x, lambda y: ffn(y, W, p + "/ffn_prepost_wrapper/ffn", ffn_rate, is_training, generator),
                    W, p + "/ffn_prepost_wrapper", eps, post_rate, is_training, generator, pre)
    if post_normalize:
        return x
    _ensure_ln(W, scope + "/output_ln", x.shape[-1], x.dtype)
    return layer_norm(x, W[scope + "/output_ln/gamma"], W[scope + "/output_ln/beta"], eps)


# --------------------------------------------------------------------------
# modalities
# --------------------------------------------------------------------------
def sinusoid_signal(length, channels, time=0, dtype=torch.float32, min_timescale=1.0, max_timescale=1.0e4):
    """add_sinusoids_timing_signal (neurst/layers/common_layers.py:356-413):
    concat(sin, cos) over channels//2 timescales, zero pad if channels is odd."""
    position = torch.arange(time, time + length, dtype=dtype)
    nts = channels // 2
    inc = math.log(float(max_timescale) / float(min_timescale)) / (float(nts) - 1)
    inv = min_timescale * torch.exp(torch.arange(nts, dtype=dtype) * -inc)
    scaled = position[:, None] * inv[None, :]
    sig = torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=1)
    if channels % 2:
        sig = F.pad(sig, (0, 1))
    return sig


def position_embedding(emb, time=None):
    """PositionEmbeddingWrapper.call, timing="sinusoids" (common_layers.py:415-434):
    emb * sqrt(d) + signal."""
    d = emb.shape[-1]
    emb = emb * (d ** 0.5)
    if emb.dim() == 3:
        return emb + sinusoid_signal(emb.shape[1], d, time or 0, emb.dtype)[None]
    return emb + sinusoid_signal(1, d, time, emb.dtype)


def word_embedding(ids, table):
    """WordEmbeddingSharedWeights._bottom (text_modalities.py:84-93)."""
    return table[ids.long()]


def tied_logits(x, table, bias):
    """WordEmbeddingSharedWeights._top (text_modalities.py:95-113)."""
    out = x @ table.t()
    return out + bias if bias is not None else out


def audio_conv_subsample(src, W, scope, layer_norm_on=True, kernel_size=3, strides=2):
    """AudioConv2dSubsamplingLayer.call (neurst/layers/modalities/audio_modalities.py:84-109).

    src [B,T,F,C] NHWC.  For each of the two layers: manual zero pad k//2 on T and
    F, Conv2D(k x k, stride s, VALID, bias) with HWIO kernel, LayerNorm over
    channels (eps 1e-6), ReLU.  Then reshape [B,T',F'*C] (F-major, C-minor) and Dense.
    Padding frames are NOT masked inside the conv.
    """
    x = src.permute(0, 3, 1, 2)  # NCHW
    pad = kernel_size // 2
    for i in (1, 2):
        k = W[f"{scope}/conv{i}/kernel"].permute(3, 2, 0, 1)  # HWIO -> OIHW
        x = F.conv2d(x, k, W[f"{scope}/conv{i}/bias"], stride=strides, padding=pad)
        if layer_norm_on:
            x = layer_norm(x.permute(0, 2, 3, 1), W[f"{scope}/ln{i}/gamma"], W[f"{scope}/ln{i}/beta"], 1e-6)
            x = x.permute(0, 3, 1, 2)
        x = relu(x.permute(0, 2, 3, 1), f"{scope}/conv{i}").permute(0, 3, 1, 2)
    x = x.permute(0, 2, 3, 1)  # [B,T',F',C]
    x = x.reshape(x.shape[0], x.shape[1], -1)
    return x @ W[f"{scope}/output_dense/kernel"] + W[f"{scope}/output_dense/bias"]


def length_after_conv(length, strides=2):
    """SpeechTransformer.get_symbols_to_logits_fn (neurst/models/speech_transformer.py:182-183)."""
    return ((length + strides - 1) // strides + strides - 1) // strides


def length_to_padding(lengths, maxlen, dtype=torch.float32):
    """model_utils.input_length_to_padding (neurst/models/model_utils.py:44-75): 1.0 = pad."""
    ar = torch.arange(maxlen)[None, :]
    return (ar >= lengths.long()[:, None]).to(dtype)


# --------------------------------------------------------------------------
# full models
# --------------------------------------------------------------------------
def _target_table(W):
    """Target embedding table: `shared/weights` when tied to the softmax, `emb/weights` otherwise
    (text_modalities.py:61-68)."""
    return W["target_symbol_modality/shared/weights"] if "target_symbol_modality/shared/weights" in W \
        else W["target_symbol_modality/emb/weights"]


def output_logits(dec, W):
    """EncoderDecoderModel.output_logits_layer (encoder_decoder_model.py:180-185): the tied table, or the separate Keras
    Dense `softmax_linear` (:64-67) when modality.share_embedding_and_softmax_weights is off."""
    if "softmax_linear/kernel" in W:
        return dec @ W["softmax_linear/kernel"] + W["softmax_linear/bias"]
    return tied_logits(dec, W["target_symbol_modality/shared/weights"], W.get("target_symbol_modality/shared/bias"))


def speech_transformer_logits(inputs, W, cfg, is_training=False, generator=None, return_intermediates=False):
    """SpeechTransformer.call (speech_transformer.py:179-189 + encoder_decoder_model.py:211-279).

    inputs: src [B,T,F,C] float, src_length [B] int, trg_input [B,L] int.
    cfg keys: num_enc, num_dec, num_heads, strides, kernel_size, layer_norm, eps,
              dropout (single rate used for every site, as the hparams sets do), timing.
    """
    strides = cfg.get("strides", 2)
    rate = cfg.get("dropout", 0.0) if is_training else 0.0
    src = inputs["src"]
    src_pad = length_to_padding(length_after_conv(inputs["src_length"], strides),
                                length_after_conv(src.shape[1], strides), src.dtype)
    emb = audio_conv_subsample(src, W, "input_audio_modality", cfg.get("layer_norm", True),
                               cfg.get("kernel_size", 3), strides)
    if cfg.get("timing", "sinusoids"):
        emb = position_embedding(emb)
    enc = transformer_encoder(emb, src_pad, W, "TransformerEncoder", cfg["num_enc"], cfg["num_heads"],
                              cfg.get("eps", 1e-6), rate, rate, rate, is_training, generator,
                              post_normalize=cfg.get("encoder_post_normalize", False))
    table = _target_table(W)
    temb = word_embedding(inputs["trg_input"], table)
    if cfg.get("timing", "sinusoids"):
        temb = position_embedding(temb)
    dec = transformer_decoder(temb, enc, src_pad, W, "TransformerDecoder", cfg["num_dec"], cfg["num_heads"],
                              cfg.get("eps", 1e-6), rate, rate, rate, is_training, generator,
                              post_normalize=cfg.get("decoder_post_normalize", False))
    logits = output_logits(dec, W)
    if return_intermediates:
        return logits, {"src_emb": emb, "enc_out": enc, "dec_out": dec, "src_padding": src_pad}
    return logits


def transformer_logits(inputs, W, cfg, is_training=False, generator=None):
    """Transformer (text) forward: EncoderDecoderModel.call with embedding on both
    sides (neurst/models/transformer.py + encoder_decoder_model.py:211-279).
    Used to pin the stack against tests/neurst/models/transformer_test.py:23-666."""
    rate = cfg.get("dropout", 0.0) if is_training else 0.0
    src_table = W["input_symbol_modality/emb/weights"]
    emb = word_embedding(inputs["src"], src_table)
    if cfg.get("timing", "sinusoids"):
        emb = position_embedding(emb)
    enc = transformer_encoder(emb, inputs["src_padding"], W, "TransformerEncoder", cfg["num_enc"],
                              cfg["num_heads"], cfg.get("eps", 1e-6), rate, rate, rate, is_training, generator,
                              monotonic=cfg.get("attention_monotonic", False),
                              post_normalize=cfg.get("encoder_post_normalize", False))
    table = _target_table(W)
    temb = word_embedding(inputs["trg_input"], table)
    if cfg.get("timing", "sinusoids"):
        temb = position_embedding(temb)
    dec = transformer_decoder(temb, enc, inputs["src_padding"], W, "TransformerDecoder", cfg["num_dec"],
                              cfg["num_heads"], cfg.get("eps", 1e-6), rate, rate, rate, is_training, generator,
                              decode_lagging=cfg.get("wait_k", None),
                              post_normalize=cfg.get("decoder_post_normalize", False))
    return output_logits(dec, W)


# --------------------------------------------------------------------------
# criterion
# --------------------------------------------------------------------------
def label_smoothed_cross_entropy(logits, labels, trg_length, label_smoothing):
    """LabelSmoothedCrossEntropy.__call__ (neurst/criterions/label_smoothed_cross_entropy.py:94-157).

    Returns (nll_sum [B], n_samples [1], n_tokens [B]) computed in fp32 (or the
    dtype of ``logits`` when it is fp64).
    """
    dt = torch.float64 if logits.dtype == torch.float64 else torch.float32
    logits = logits.to(dt)
    V = logits.shape[-1]
    confidence = 1.0 - label_smoothing
    low = label_smoothing / float(V - 1)
    soft = torch.full_like(logits, low)
    soft.scatter_(-1, labels.long().unsqueeze(-1), confidence)
    xent = -(soft * torch.log_softmax(logits, dim=-1)).sum(-1)
    if label_smoothing:
        norm = -(confidence * math.log(confidence) + float(V - 1) * low * math.log(low + 1e-20))
        xent = xent - norm
    weights = 1.0 - length_to_padding(trg_length, labels.shape[1], dt)
    nll_sum = (xent * weights).sum(1)
    n_samples = torch.tensor([float(labels.shape[0])], dtype=dt)
    n_tokens = weights.sum(1)
    return nll_sum, n_samples, n_tokens


def reduce_loss(nll_sum, n_tokens):
    """LabelSmoothedCrossEntropy.reduce_loss (label_smoothed_cross_entropy.py:46-53)."""
    return nll_sum.sum() / n_tokens.sum()


# --------------------------------------------------------------------------
# task glue
# --------------------------------------------------------------------------
def deduce_text_length_eos_as_padding(trg, pad_id):
    """model_utils.deduce_text_length, EOS_AS_PADDING (neurst/models/model_utils.py:23-41):
    argmin(trg != pad) + 1."""
    ne = (trg != pad_id).to(torch.int32)
    return torch.argmin(ne, dim=-1) + 1


def example_to_input(audio, audio_length, transcript, feature_dim, channels, bos_id, pad_id):
    """SpeechToText.example_to_input, training mode (neurst/tasks/speech2text.py:135-161)."""
    b = audio.shape[0]
    src = audio.reshape(b, -1, feature_dim, channels)
    bos = torch.full((b, 1), bos_id, dtype=transcript.dtype)
    return {"src": src, "src_length": audio_length, "trg": transcript,
            "trg_length": deduce_text_length_eos_as_padding(transcript, pad_id),
            "trg_input": torch.cat([bos, transcript[:, :-1]], dim=1)}


# --------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------
def noam_lr(global_step, dmodel, warmup_steps, initial_factor=1.0, end_factor=None,
            start_decay_at=0, decay_steps=None, initial_step=0):
    """NoamSchedule.__call__ (neurst/optimizers/schedules/noam_schedule.py:76-97), float32 math in python floats."""
    if end_factor is None or start_decay_at is None or decay_steps is None:
        end_factor, start_decay_at, decay_steps = initial_factor, 0, 1
    s = float(global_step) + float(initial_step) + 1.0
    step_factor = max(min(s - start_decay_at, decay_steps), 0.0)
    lr = end_factor + (initial_factor - end_factor) * (1.0 - step_factor / decay_steps)
    lr *= dmodel ** -0.5
    lr *= min(1.0, s / warmup_steps)
    lr /= math.sqrt(max(s, warmup_steps))
    return lr


def keras_adam_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.98, eps=1e-9):
    """Keras Adam (non-amsgrad) dense update, iteration ``t`` counted from 1:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps).
    (epsilon OUTSIDE the bias correction -- differs from torch.optim.Adam.)"""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    p = p - lr_t * m / (torch.sqrt(v) + eps)
    return p, m, v


def clip_gradients(grads, clip_value=None, clip_norm=None):
    """GradAccumKerasModel.train_step (neurst/training/gradaccum_keras_model.py:228-233): tf.clip_by_value on every
    gradient, else tf.clip_by_norm PER gradient tensor (t * clip_norm / max(||t||_2, clip_norm))."""
    out = {}
    for n, g in grads.items():
        if clip_value:
            out[n] = g.clamp(-clip_value, clip_value)
        elif clip_norm:
            out[n] = g * (clip_norm / torch.maximum(g.norm(), torch.tensor(float(clip_norm), dtype=g.dtype)))
        else:
            out[n] = g
    return out


def average_gradients(per_rank_grads):
    """hvd.Average over ranks (neurst/training/hvd_utils.py:46-62): elementwise mean of
    the per-rank gradients (each rank's gradient is of its LOCAL token-mean loss)."""
    n = len(per_rank_grads)
    return [sum(gs) / n for gs in zip(*per_rank_grads)]


# --------------------------------------------------------------------------
# init helpers (used by tests and the CPU baseline)
# --------------------------------------------------------------------------
def init_speech_transformer_weights(cfg, vocab_size, feature_dim=80, in_channels=1, seed=42, dtype=torch.float32):
    """Random weights with the reference's initialisers: glorot_uniform
    kernels, zero biases, N(0, d^-0.5) embedding, LN gamma=1 beta=0."""
    g = torch.Generator().manual_seed(seed)
    d, C, ffn_dim = cfg["d_model"], cfg["channels"], cfg["ffn"]
    W = {}

    def glorot(*shape, fan_in=None, fan_out=None):
        fi = fan_in if fan_in is not None else shape[0]
        fo = fan_out if fan_out is not None else shape[-1]
        lim = math.sqrt(6.0 / (fi + fo))
        return ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(dtype)

    a = "input_audio_modality"
    W[f"{a}/conv1/kernel"] = glorot(3, 3, in_channels, C, fan_in=9 * in_channels, fan_out=9 * C)
    W[f"{a}/conv1/bias"] = torch.zeros(C, dtype=dtype)
    W[f"{a}/conv2/kernel"] = glorot(3, 3, C, C, fan_in=9 * C, fan_out=9 * C)
    W[f"{a}/conv2/bias"] = torch.zeros(C, dtype=dtype)
    for i in (1, 2):
        W[f"{a}/ln{i}/gamma"] = torch.ones(C, dtype=dtype)
        W[f"{a}/ln{i}/beta"] = torch.zeros(C, dtype=dtype)
    fprime = length_after_conv(feature_dim)
    W[f"{a}/output_dense/kernel"] = glorot(fprime * C, d)
    W[f"{a}/output_dense/bias"] = torch.zeros(d, dtype=dtype)
    W["target_symbol_modality/shared/weights"] = (torch.randn(vocab_size, d, generator=g, dtype=torch.float64)
                                                  * d ** -0.5).to(dtype)
    W["target_symbol_modality/shared/bias"] = glorot(vocab_size, fan_in=vocab_size, fan_out=vocab_size)

    def ln(p):
        W[p + "/gamma"] = torch.ones(d, dtype=dtype)
        W[p + "/beta"] = torch.zeros(d, dtype=dtype)

    def dense(p, i, o):
        W[p + "/kernel"] = glorot(i, o)
        W[p + "/bias"] = torch.zeros(o, dtype=dtype)

    for i in range(cfg["num_enc"]):
        p = f"TransformerEncoder/layer_{i}"
        ln(p + "/self_attention_prepost_wrapper/ln")
        dense(p + "/self_attention_prepost_wrapper/self_attention/qkv_transform", d, 3 * d)
        dense(p + "/self_attention_prepost_wrapper/self_attention/output_transform", d, d)
        ln(p + "/ffn_prepost_wrapper/ln")
        dense(p + "/ffn_prepost_wrapper/ffn/dense1", d, ffn_dim)
        dense(p + "/ffn_prepost_wrapper/ffn/dense2", ffn_dim, d)
    ln("TransformerEncoder/output_ln")
    for i in range(cfg["num_dec"]):
        p = f"TransformerDecoder/layer_{i}"
        ln(p + "/self_attention_prepost_wrapper/ln")
        dense(p + "/self_attention_prepost_wrapper/self_attention/qkv_transform", d, 3 * d)
        dense(p + "/self_attention_prepost_wrapper/self_attention/output_transform", d, d)
        ln(p + "/encdec_attention_prepost_wrapper/ln")
        dense(p + "/encdec_attention_prepost_wrapper/encdec_attention/q_transform", d, d)
        dense(p + "/encdec_attention_prepost_wrapper/encdec_attention/kv_transform", d, 2 * d)
        dense(p + "/encdec_attention_prepost_wrapper/encdec_attention/output_transform", d, d)
        ln(p + "/ffn_prepost_wrapper/ln")
        dense(p + "/ffn_prepost_wrapper/ffn/dense1", d, ffn_dim)
        dense(p + "/ffn_prepost_wrapper/ffn/dense2", ffn_dim, d)
    ln("TransformerDecoder/output_ln")
    return W


def train_step_reference(W, inputs, cfg, label_smoothing, is_training=False, generator=None):
    """One forward + loss + backward on the oracle.  Returns (loss, logits, grads by name).  With is_training and
    cfg["dropout"] > 0, `generator` supplies the masks (see dropout())."""
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    logits = speech_transformer_logits(inputs, Wg, cfg, is_training=is_training, generator=generator)
    nll, _, ntok = label_smoothed_cross_entropy(logits, inputs["trg"], inputs["trg_length"], label_smoothing)
    loss = reduce_loss(nll, ntok)
    names = list(Wg.keys())
    grads = torch.autograd.grad(loss, [Wg[n] for n in names], allow_unused=True)
    return loss.detach(), logits.detach(), {n: (g if g is not None else torch.zeros_like(W[n]))
                                            for n, g in zip(names, grads)}


def text_train_step_reference(W, inputs, cfg, label_smoothing, is_training=False, generator=None):
    """train_step_reference for the text Transformer (transformer_logits): inputs carry src ids, src_length,
    trg, trg_length, trg_input; the source padding follows EncoderDecoderModel.get_symbols_to_logits_fn /
    call (encoder_decoder_model.py:211-224: padding = 1 - sequence_mask(src_length))."""
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    inp = dict(inputs)
    inp["src_padding"] = length_to_padding(inputs["src_length"], inputs["src"].shape[1]).to(next(iter(W.values())).dtype)
    if "input_symbol_modality/emb/weights" not in Wg:  # shared source/target embedding
        Wg2 = dict(Wg)
        Wg2["input_symbol_modality/emb/weights"] = Wg["shared_symbol_modality/shared/weights"]
        Wg2["target_symbol_modality/shared/weights"] = Wg["shared_symbol_modality/shared/weights"]
        if "shared_symbol_modality/shared/bias" in Wg:
            Wg2["target_symbol_modality/shared/bias"] = Wg["shared_symbol_modality/shared/bias"]
    else:
        Wg2 = Wg
    logits = transformer_logits(inp, Wg2, cfg, is_training=is_training, generator=generator)
    nll, _, ntok = label_smoothed_cross_entropy(logits, inputs["trg"], inputs["trg_length"], label_smoothing)
    loss = reduce_loss(nll, ntok)
    names = list(Wg.keys())
    grads = torch.autograd.grad(loss, [Wg[n] for n in names], allow_unused=True)
    return loss.detach(), logits.detach(), {n: (g if g is not None else torch.zeros_like(W[n]))
                                            for n, g in zip(names, grads)}
