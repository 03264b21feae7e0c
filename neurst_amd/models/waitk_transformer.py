"""WaitkTransformer (neurst/models/waitk_transformer.py:25-139): the text Transformer trained / decoded under a wait-k
policy -- a monotonic (causal) encoder and a decoder whose cross attention at target position i sees the first
i + k source positions only.

  training   wait_k: an int, or a list of laggings one of which is drawn per step (:97-100; the draw here is a
             deterministic function of the runtime's step seed instead of tf.random);
  inference  with the full source available (the reference's offline generation path, get_decoder_output with `time`):
             step t sees the first k + t positions (:103-104).  The streaming agent (incremental_encode /
             SimulEval) is not built.
"""
import numpy as np

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
