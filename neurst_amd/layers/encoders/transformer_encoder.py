"""TransformerEncoder (neurst/layers/encoders/transformer_encoder.py:23-136), training/eval path."""
import torch

from neurst_amd import kernels as K
from neurst_amd.layers import layer_utils
from neurst_amd.layers.common_layers import LayerNorm, ResidualStream, dropped_grad
from neurst_amd.layers.encoders.encoder import Encoder, register_encoder
from neurst_amd.layers.transformer_layers import TransformerEncoderLayer


@register_encoder
class TransformerEncoder(Encoder):
    def __init__(self, num_layers, hidden_size, num_attention_heads, filter_size, ffn_activation="relu",
                 attention_dropout_rate=0., attention_type="dot_product", ffn_dropout_rate=0.,
                 layer_postprocess_dropout_rate=0., layer_postprocess_epsilon=1e-6, post_normalize=False,
                 attention_monotonic=False, return_all_layers=False, name=None):
        super().__init__(num_layers=num_layers, hidden_size=hidden_size, num_attention_heads=num_attention_heads,
                         filter_size=filter_size, ffn_activation=ffn_activation,
                         attention_dropout_rate=attention_dropout_rate, attention_type=attention_type,
                         ffn_dropout_rate=ffn_dropout_rate,
                         layer_postprocess_dropout_rate=layer_postprocess_dropout_rate,
                         layer_postprocess_epsilon=layer_postprocess_epsilon, post_normalize=post_normalize,
                         attention_monotonic=attention_monotonic, return_all_layers=return_all_layers)
        assert post_normalize or not return_all_layers, \
            "`return_all_layers` is only available when `post_normalize`=True."   # transformer_encoder.py:74-75
        if return_all_layers:
            raise NotImplementedError("return_all_layers (BERT-style feature extraction) is off the path")
        self.name = name or self.__class__.__name__
        self._built = False

    def build(self, rt, gen):
        """Creates variables (the reference does this lazily in Keras build(); here it is explicit)."""
        p = self._params
        self.rt = rt
        self._stacking_layers = [
            TransformerEncoderLayer(rt, f"{self.name}/layer_{i}", p["hidden_size"], p["num_attention_heads"],
                                    p["filter_size"], gen, p["ffn_activation"], p["attention_dropout_rate"],
                                    p["attention_type"], p["ffn_dropout_rate"], p["layer_postprocess_dropout_rate"],
                                    p["layer_postprocess_epsilon"], p["post_normalize"])
            for i in range(p["num_layers"])]
        # post-norm stacks end with the last wrapper's LayerNorm: no output_ln (transformer_encoder.py:97-100)
        self._output_norm_layer = None if p["post_normalize"] else LayerNorm(
            rt, f"{self.name}/output_ln", p["hidden_size"], p["layer_postprocess_epsilon"])
        self._site = rt.new_dropout_site()
        self._stream32 = ResidualStream.supported(rt, p["hidden_size"], pre_norm=not p["post_normalize"])
        self._built = True
        return self

    def _finish(self, x, save):
        """Output of the stack from the residual stream / tensor behind the last layer."""
        if isinstance(x, ResidualStream):
            return self._output_norm_layer.forward_stream(x, save=save, want_sum=False)[0]
        return x if self._output_norm_layer is None else self._output_norm_layer.forward(x, save=save)

    def forward(self, inputs, inputs_padding, is_training=True):
        """inputs [B,T,d] (device, compute dtype), inputs_padding [B,T] float (1.0 = pad) -> [B,T,d]."""
        B, T, d = inputs.shape
        bias = layer_utils.input_padding_to_bias(inputs_padding)
        x = inputs.reshape(B * T, d)
        p = self._params["layer_postprocess_dropout_rate"] if is_training else 0.0
        self._p = p
        if p > 0:
            x = K.scale_posenc_dropout_fwd(x, None, 1, 1.0, p, self.rt.step_seed, self._site)
        # attention_monotonic (transformer_encoder.py:121-123): min(padding bias, lower-triangle bias) = the kernel's
        # key bias + causal flag together
        causal = bool(self._params["attention_monotonic"])
        if self._stream32:
            x = ResidualStream(x if x.is_contiguous() else x.contiguous())
        for layer in self._stacking_layers:
            x = layer.forward(x, B, T, bias, is_training=is_training, causal=causal)
        return self._finish(x, is_training).view(B, T, d)

    __call__ = forward

    def create_incremental_cache(self, batch, max_length, dtype=None):
        """Per-layer self-attention key / value buffers for streaming input (transformer_layers.py:100-108 creates
        empty tensors and concatenates; here `max_length` positions are preallocated and filled in place)."""
        d = self._params["hidden_size"]
        kv = torch.zeros(self._params["num_layers"], 2, batch, max_length, d, dtype=dtype or self.rt.dtype,
                         device=self.rt.device)
        return {f"layer_{i}": {"self_attention": {"keys": kv[i, 0], "values": kv[i, 1], "len": 0}}
                for i in range(self._params["num_layers"])}

    def incremental_encode(self, inputs, cache, time=None, max_length=1024):
        """Encoding of streaming input (transformer_encoder.py:138-175): `inputs` [B, n, d] (or [B, d]) are the embedded
        positions time .. time+n-1; every layer attends over its cached keys / values of the earlier positions plus
        the chunk itself under the causal mask.  Only for attention_monotonic encoders -- there the result equals the
        rows of the full forward.  Returns (outputs [B, n, d], cache)."""
        assert self._params["attention_monotonic"], \
            "function `incremental_encode` only available when attention_monotonic=True"
        x3 = inputs[:, None, :] if inputs.dim() == 2 else inputs
        B, n, d = x3.shape
        if not cache:
            cache = self.create_incremental_cache(B, max_length, x3.dtype)
        filled = cache["layer_0"]["self_attention"]["len"] if self._stacking_layers else 0
        if time is not None and int(time) != filled:
            raise ValueError(f"incremental_encode: chunk starts at time {time} but {filled} positions are cached")
        x = x3.reshape(B * n, d).contiguous()
        if self._stream32:
            x = ResidualStream(x)
        for i, layer in enumerate(self._stacking_layers):
            x = layer.forward(x, B, n, None, is_training=False, cache=cache[f"layer_{i}"])
        return self._finish(x, False).view(B, n, d), cache

    def backward(self, dout, layer_done=None):
        """layer_done(prefixes): called after each layer's backward has been QUEUED (its gradients are complete once the
        streams that were current until then have drained) -- the data-parallel reducer's per-layer buckets."""
        B, T, d = dout.shape
        layers = self._stacking_layers
        # every gradient on the residual chain next meets a dropout mask: the LayerNorm backward that produces it also
        # emits the masked copy (consumer = that dropout site), saving one element-wise pass per sublayer
        dx = dout.reshape(B * T, d)
        if self._output_norm_layer is not None:
            dx = self._output_norm_layer.backward(dx, consumer=layers[-1].first_backward_site if layers else self)
        elif not dx.is_contiguous():
            dx = dx.contiguous()
        for i in range(len(layers) - 1, -1, -1):
            dx = layers[i].backward(dx, consumer=layers[i - 1].first_backward_site if i > 0 else self)
            if layer_done is not None:
                # output_ln is registered right behind the top layer: it travels with that layer's report (alone it would
                # be a 2 KB all-reduce of its own when the reducer sweeps up what no report covered)
                extra = [self._output_norm_layer.name + "/"] if (i == len(layers) - 1 and self._output_norm_layer is not None) else []
                layer_done([layers[i].name + "/"] + extra)
        dx = dropped_grad(self.rt, dx, self._p, self._site)
        return dx.view(B, T, d)

    # the encoder's input dropout as a dropout site (see LayerNorm.backward)
    @property
    def site(self):
        return self._site

    def drop_rate(self):
        return self._p
