"""SpeechTransformer (neurst/models/speech_transformer.py:27-280) registered under the same names/flags, built on
the MI355X-native layers."""
import torch

from neurst_amd.layers.common_layers import PositionEmbeddingWrapper
from neurst_amd.layers.decoders import build_decoder
from neurst_amd.layers.encoders import build_encoder
from neurst_amd.layers.modalities.audio_modalities import AudioConv2dSubsamplingLayer
from neurst_amd.models.encoder_decoder_model import EncoderDecoderModel, _timing_name
from neurst_amd.models.model import register_model
from neurst_amd.models.model_utils import input_length_to_padding
from neurst_amd.utils.flags_core import Flag


@register_model
class SpeechTransformer(EncoderDecoderModel):
    """Defines the Speech Transformer model."""

    # Flag table: the reference's names, types and defaults (models/speech_transformer.py:35-106), one row per flag.
    _I, _F, _S, _B = Flag.TYPE.INTEGER, Flag.TYPE.FLOAT, Flag.TYPE.STRING, Flag.TYPE.BOOLEAN
    _MODALITY_FLAGS = (
        ("share_embedding_and_softmax_weights", _B, False, "tie the target embedding table and the logits projection"),
        ("dim", _I, None, "embedding width of both sides"), ("source.dim", _I, None, "source embedding width"),
        ("target.dim", _I, None, "target embedding width"), ("timing", _S, None, "position signal of both sides"),
        ("source.timing", _S, None, "source position signal"), ("target.timing", _S, None, "target position signal"),
        ("source.kernel_size", _I, 3, "conv kernel size"), ("source.strides", _I, 2, "conv stride"),
        ("source.channels", _I, 256, "conv channels"), ("source.layer_norm", _B, False, "LayerNorm after each conv layer"))
    _STACK_FLAGS = (
        ("num_layers", _I, None, "layers"), ("hidden_size", _I, None, "model width"),
        ("num_attention_heads", _I, None, "attention heads"), ("filter_size", _I, None, "FFN width"),
        ("ffn_activation", _S, "relu", "FFN activation"), ("attention_dropout_rate", _F, 0., "dropout on attention probabilities"),
        ("attention_type", _S, "dot_product", "attention function"), ("ffn_dropout_rate", _F, 0., "dropout on the FFN hidden layer"),
        ("post_normalize", _B, False, "LayerNorm after (not before) each sub-layer"),
        ("attention_monotonic", _B, False, "causal encoder self attention (streaming / wait-k); encoder only"),
        ("layer_postprocess_dropout_rate", _F, 0., "dropout on every sub-layer output"),
        ("layer_postprocess_epsilon", _F, 1e-6, "LayerNorm epsilon"))

    @staticmethod
    def class_or_method_args():
        cls = SpeechTransformer
        flags = [Flag("modality." + n, dtype=t, default=d, help=h) for n, t, d, h in cls._MODALITY_FLAGS]
        for side in ("encoder", "decoder"):
            flags += [Flag(f"{side}.{n}", dtype=t, default=d, help=f"{side}: {h}") for n, t, d, h in cls._STACK_FLAGS
                      if not (side == "decoder" and n == "attention_monotonic")]
        return flags

    @classmethod
    def build_modalities(cls, rt, gen, model_args, src_meta, trg_meta):
        """speech_transformer.py:108-140 (variable creation order: target modality first, then the audio one)."""
        src_dim = model_args["modality.source.dim"] or model_args["modality.dim"]
        trg_dim = model_args["modality.target.dim"] or model_args["modality.dim"]
        input_name, target_name = "input_audio_modality", "target_symbol_modality"
        target_modality = cls.build_modality(
            rt, gen, vocab_size=trg_meta["vocab_size"], emb_dim=trg_dim, name=target_name,
            timing=(model_args["modality.target.timing"] or model_args["modality.timing"]),
            share_embedding_and_softmax_weights=model_args["modality.share_embedding_and_softmax_weights"])
        input_modality = AudioConv2dSubsamplingLayer(
            rt, input_name, embedding_dim=src_dim, input_dimension=src_meta["audio_feature_dim"], gen=gen,
            input_channels=src_meta.get("audio_feature_channels", 1),
            kernel_size=model_args["modality.source.kernel_size"], strides=model_args["modality.source.strides"],
            channels=model_args["modality.source.channels"], layer_norm=model_args["modality.source.layer_norm"])
        src_timing = _timing_name(model_args["modality.source.timing"] or model_args["modality.timing"])
        if src_timing:
            input_modality = PositionEmbeddingWrapper(rt, input_name + "_posenc_wrapper", input_modality,
                                                      timing=src_timing)
        return input_modality, target_modality

    @classmethod
    def new(cls, args, src_meta, trg_meta, name=None, **kwargs):
        """speech_transformer.py:142-177.  Extra keyword arguments select the runtime:
        runtime=Runtime(...) or device= / dtype= / seed= / init_seed=."""
        rt, gen = cls._runtime(kwargs)
        src_modality, trg_modality = cls.build_modalities(rt, gen, args, src_meta, trg_meta)
        encoder_params, decoder_params = {}, {}
        for f in cls.class_or_method_args():
            if f.name in args:
                if f.name.startswith("encoder."):
                    encoder_params[f.name[8:]] = args[f.name]
                elif f.name.startswith("decoder."):
                    decoder_params[f.name[8:]] = args[f.name]
        encoder = build_encoder({"encoder.class": "TransformerEncoder", "encoder.params": encoder_params}).build(rt, gen)
        decoder = build_decoder({"decoder.class": "TransformerDecoder", "decoder.params": decoder_params}).build(rt, gen)
        model = cls(args, src_meta, trg_meta, src_modality, trg_modality, encoder, decoder, name=name, rt=rt, gen=gen)
        return model.finalize()

    def _src_padding(self, inputs, embedded_inputs):
        """speech_transformer.py:179-189: padding mask at the conv-subsampled rate."""
        strides = self.args["modality.source.strides"]

        def _length_after_conv(_l):
            return ((_l + strides - 1) // strides + strides - 1) // strides

        if strides >= 1:      # the two ceil-divisions of the lengths ride inside the mask kernel
            return input_length_to_padding(inputs["src_length"], _length_after_conv(inputs["src"].shape[1]), halvings=2,
                                           stride=strides)
        return input_length_to_padding(_length_after_conv(inputs["src_length"]),
                                       _length_after_conv(inputs["src"].shape[1]))

    @classmethod
    def build_model_args_by_name(cls, name):
        """speech_transformer.py:191-280."""
        if not name.startswith("speech_transformer"):
            return None
        table = {  # dmodel, heads, enc layers, dec layers, enc filter, dec filter, channels
            "speech_transformer_toy": (8, 2, 2, 2, 10, 10, 5),
            "speech_transformer_s": (256, 4, 12, 6, 2048, 2048, 256),
            "speech_transformer_m": (512, 8, 12, 6, 2048, 2048, 256),
            "speech_transformer_l": (1024, 16, 12, 6, 4096, 4096, 512),
        }
        if name not in table:
            return None
        dmodel, num_heads, n_enc, n_dec, f_enc, f_dec, channels = table[name]
        dropout_rate = 0.1
        return {
            "model.class": cls.__name__,
            "model.params": {
                "modality.source.kernel_size": 3,
                "modality.source.strides": 2,
                "modality.source.channels": channels,
                "modality.source.layer_norm": True,
                "modality.dim": dmodel,
                "modality.share_embedding_and_softmax_weights": True,
                "modality.timing": "sinusoids",
                "encoder.num_layers": n_enc,
                "encoder.hidden_size": dmodel,
                "encoder.num_attention_heads": num_heads,
                "encoder.filter_size": f_enc,
                "encoder.attention_dropout_rate": dropout_rate,
                "encoder.attention_type": "dot_product",
                "encoder.ffn_activation": "relu",
                "encoder.ffn_dropout_rate": dropout_rate,
                "encoder.layer_postprocess_dropout_rate": dropout_rate,
                "decoder.num_layers": n_dec,
                "decoder.hidden_size": dmodel,
                "decoder.num_attention_heads": num_heads,
                "decoder.filter_size": f_dec,
                "decoder.attention_dropout_rate": dropout_rate,
                "decoder.attention_type": "dot_product",
                "decoder.ffn_activation": "relu",
                "decoder.ffn_dropout_rate": dropout_rate,
                "decoder.layer_postprocess_dropout_rate": dropout_rate,
            },
            "optimizer.class": "Adam",
            "optimizer.params": {"epsilon": 1.e-9, "beta_1": 0.9, "beta_2": 0.98},
            "lr_schedule.class": "noam",
            "lr_schedule.params": {
                "initial_factor": 5.0 if dmodel > 256 else 3.5,
                "end_factor": 2.0 if dmodel > 256 else 1.5,
                "dmodel": dmodel,
                "warmup_steps": 25000,
                "start_decay_at": 50000,
                "decay_steps": 50000,
            },
        }
