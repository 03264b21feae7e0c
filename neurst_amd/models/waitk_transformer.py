"""WaitkTransformer (neurst/models/waitk_transformer.py:25-139): the text Transformer trained / decoded under a wait-k
policy -- a monotonic (causal) encoder and a decoder whose cross attention at target position i sees the first
i + k source positions only.

  training   wait_k: an int, or a list of laggings one of which is drawn per step (:97-100; the draw here is a
             deterministic function of the runtime's step seed instead of tf.random);
  inference  with the full source available (the reference's offline generation path, get_decoder_output with `time`):
             step t sees the first k + t positions (:103-104);
  streaming  incremental_encode / incremental_decode (:117-139): source chunks are encoded as they arrive (monotonic
             encoder over cached keys / values), appended to the decoder's memory, and one target position is
             decoded against whatever has been read; neurst_amd/utils/simuleval_agents drives them.
"""
import numpy as np
import torch

from neurst_amd.models.model import register_model
from neurst_amd.models.transformer import Transformer
from neurst_amd.utils.flags_core import Flag


@register_model
class WaitkTransformer(Transformer):
    wait_k = None

    @staticmethod
    def class_or_method_args():
        flags = Transformer.class_or_method_args()
        flags.append(Flag("wait_k", dtype=Flag.TYPE.STRING, default=None,
                          help="The lagging k, or a list of laggings sampled per training step (the reference passes it "
                               "through the task as `waitk_lagging`)."))
        return flags

    @classmethod
    def new(cls, args, src_meta, trg_meta, name=None, waitk_lagging=None, **kwargs):
        args = dict(args)
        args["encoder.attention_monotonic"] = True
        model = super().new(args, src_meta, trg_meta, name=name, **kwargs)
        lag = waitk_lagging if waitk_lagging is not None else args.get("wait_k", None)
        if isinstance(lag, str):
            import yaml
            lag = yaml.safe_load(lag)
        model.wait_k = lag
        return model

    @classmethod
    def build_model_args_by_name(cls, name):
        """waitk_transformer.py:75-85: waitk_transformer_<set> / waitktransformer_<set>."""
        args = None
        if name.startswith("waitk_transformer_"):
            args = Transformer.build_model_args_by_name(name[6:])
        elif name.startswith("waitktransformer_"):
            args = Transformer.build_model_args_by_name(name[5:])
        if args is not None:
            args["model.class"] = cls.__name__
            args["model.params"]["encoder.attention_monotonic"] = True
        return args

    # ------------------------------------------------------------------ streaming (waitk_transformer.py:117-139)
    def incremental_encode(self, inputs, encoder_cache, decoder_cache, time=None, max_source_length=1024,
                           decode_padded_length=256):
        """inputs: {"src": ids [B, n] (or [B] with `time`) of the NEW source positions time .. time+n-1, "src_length"
        [B] (valid positions of the chunk) or "src_padding" [B, n]}.  Returns (encoder_cache, decoder_cache)."""
        dev = self.rt.device
        src = torch.as_tensor(inputs["src"], dtype=torch.int64, device=dev)
        assert not (src.dim() == 1 and time is None)
        if src.dim() == 1:
            src = src[:, None]
        B, n = src.shape
        time = 0 if time is None else int(time)
        emb = self._src_modality.forward(src, is_training=False, time=time)   # signal rows time .. time+n-1
        src_padding = inputs.get("src_padding", None)
        if src_padding is None:
            from neurst_amd.models.model_utils import input_length_to_padding
            src_padding = input_length_to_padding(torch.as_tensor(inputs["src_length"], device=dev), n)
        src_padding = torch.as_tensor(src_padding, dtype=torch.float32, device=dev)
        enc_out, encoder_cache = self._encoder.incremental_encode(emb, encoder_cache, time, max_length=max_source_length)
        decoder_cache = self._decoder.update_incremental_cache(decoder_cache, enc_out, src_padding,
                                                               max_source_length=max_source_length,
                                                               decode_padded_length=decode_padded_length)
        return encoder_cache, decoder_cache

    def incremental_decode(self, symbols, cache, time=None):
        """One target position against the memory read so far; symbols: ids [B] (the previous output, or BOS / EOS at
        time 0).  Returns (logits [B, V], cache).  The lagging is wait_k + time as in offline decoding -- positions
        that have not been read yet simply do not exist in the memory."""
        ids = torch.as_tensor(symbols, dtype=torch.int64, device=self.rt.device).reshape(-1)
        time = 0 if time is None else int(time)
        dec_in = self._trg_modality.forward(ids, is_training=False, time=time)
        hidden = self._decoder.decode_step(dec_in, cache, decode_lagging=self.decode_lagging(False, time))
        return self.output_logits_layer(hidden, is_training=False), cache

    def decode_lagging(self, is_training, time):
        lag = self.wait_k
        if lag is None:
            return None
        if isinstance(lag, (list, tuple)):
            if is_training:
                lag = lag[int(np.random.RandomState(self.rt.step_seed & 0x7FFFFFFF).randint(len(lag)))]
            else:
                lag = lag[0]
        lag = int(lag)
        if time is not None:
            lag += int(time)
        return lag
