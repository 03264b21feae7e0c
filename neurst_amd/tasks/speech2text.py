"""SpeechToText task (neurst/tasks/speech2text.py:61-161): example_to_input semantics and model construction.
The tf.data bucketing pipeline (speech2text.py:236-384) is replaced by datasets that already yield padded
batches (see neurst_amd/data/datasets/synthetic_speech.py); real TFRecord feeding is §8(f) rank 2."""
import torch

from neurst_amd.models import build_model
from neurst_amd.models.model_utils import deduce_text_length
from neurst_amd.tasks.task import Task, register_task
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


@register_task(["speech2text", "audio2text", "AudioToText"])
class SpeechToText(Task):
    def __init__(self, args):
        super().__init__(args)
        self._audio_feature_dim = args.get("audio_feature_dim", 80) or 80
        self._audio_feature_channels = args.get("audio_feature_channels", 1) or 1
        vocab_size = args.get("vocab_size", None) or 8008
        # TextDataPipeline appends <UNK>, <SEQ_BEG>, <SEQ_END> at the END of the vocabulary and pads with EOS
        # (neurst/data/data_pipelines/text_data_pipeline.py:71-92)
        self._trg_meta = {"vocab_size": vocab_size, "unk_id": vocab_size - 3, "bos_id": vocab_size - 2,
                          "eos_id": vocab_size - 1, "pad_id": vocab_size - 1,
                          "padding_mode": compat.PaddingMode.EOS_AS_PADDING}

    @staticmethod
    def class_or_method_args():
        return [
            Flag("audio_feature_dim", dtype=Flag.TYPE.INTEGER, default=80, help="The dimension of audio features."),
            Flag("audio_feature_channels", dtype=Flag.TYPE.INTEGER, default=1, help="The channels of audio features."),
            Flag("vocab_size", dtype=Flag.TYPE.INTEGER, default=8008,
                 help="Target vocabulary size INCLUDING <UNK>,<SEQ_BEG>,<SEQ_END> (stands in for the reference's "
                      "transcript_data_pipeline vocabulary files)."),
            Flag("max_src_len", dtype=Flag.TYPE.INTEGER, default=None, help="Maximum source length (frames)."),
            Flag("max_trg_len", dtype=Flag.TYPE.INTEGER, default=None, help="Maximum target length."),
            Flag("batch_size", dtype=Flag.TYPE.INTEGER, default=None, help="Global batch size in FRAMES."),
            Flag("batch_size_per_gpu", dtype=Flag.TYPE.INTEGER, default=None, help="Per-GPU batch size in FRAMES."),
            Flag("experimental_frame_transcript_ratio", dtype=Flag.TYPE.INTEGER, default=None,
                 help="The ratio of the number of frames and its transcript for training batch bucket."),
        ]

    @property
    def trg_meta(self):
        return self._trg_meta

    def get_config(self):
        return {"audio_feature_dim": self._audio_feature_dim, "audio_feature_channels": self._audio_feature_channels,
                "vocab_size": self._trg_meta["vocab_size"]}

    def build_model(self, args, name=None, **kwargs):
        return build_model(args, {"audio_feature_dim": self._audio_feature_dim,
                                  "audio_feature_channels": self._audio_feature_channels},
                           self._trg_meta, name=name, **kwargs)

    def example_to_input(self, batch_of_data, mode):
        """speech2text.py:135-161.  batch_of_data: {"audio" [B, T*F*C] f32, "audio_length" [B], "transcript" [B,L]}."""
        audio = batch_of_data["audio"]
        batch = audio.shape[0]
        input_dict = {"src": audio.reshape(batch, -1, self._audio_feature_dim, self._audio_feature_channels),
                      "src_length": batch_of_data["audio_length"]}
        bos = torch.full((batch,), self._trg_meta["bos_id"], dtype=torch.int64, device=audio.device)
        if mode == compat.ModeKeys.INFER:
            input_dict["trg_input"] = bos
        else:
            tr = batch_of_data["transcript"]
            input_dict["trg"] = tr
            input_dict["trg_length"] = deduce_text_length(tr, self._trg_meta["pad_id"], self._trg_meta["padding_mode"])
            input_dict["trg_input"] = torch.cat([bos[:, None], tr[:, :-1]], dim=1)
        return input_dict
