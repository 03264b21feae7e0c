"""SpeechToText task (neurst/tasks/speech2text.py:61-400): example_to_input semantics, model construction, and the
training data feed -- per-example preprocessing (truncation, SpecAugment, transcript ids; :163-234) and the
frame-bucketed, padded batching of the reference's tf.data pipeline (:236-384) as plain-Python iterators over
neurst_amd/data/batching.py.  Datasets that already yield padded batches (synthetic_speech) bypass the bucketing."""
import numpy as np
import torch

from neurst_amd.data import batching
from neurst_amd.data.text_pipeline import TextDataPipeline
from neurst_amd.utils.audio_lib import SpecAugment

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
        self._args = dict(args)
        pipeline_params = args.get("transcript_data_pipeline.params", None) or {}
        if pipeline_params.get("vocab_path", None):
            self._trg_data_pipeline = TextDataPipeline(**pipeline_params)
            self._trg_meta = self._trg_data_pipeline.meta
        else:
            self._trg_data_pipeline = None
            vocab_size = args.get("vocab_size", None) or 8008
            # TextDataPipeline appends <UNK>, <SEQ_BEG>, <SEQ_END> at the END of the vocabulary and pads with EOS
            # (neurst/data/data_pipelines/text_data_pipeline.py:71-92)
            self._trg_meta = {"vocab_size": vocab_size, "unk_id": vocab_size - 3, "bos_id": vocab_size - 2,
                              "eos_id": vocab_size - 1, "pad_id": vocab_size - 1,
                              "padding_mode": compat.PaddingMode.EOS_AS_PADDING}
        self._specaug = SpecAugment.build(args.get("specaug", None))

    @staticmethod
    def class_or_method_args():
        return [
            Flag("audio_feature_dim", dtype=Flag.TYPE.INTEGER, default=80, help="The dimension of audio features."),
            Flag("audio_feature_channels", dtype=Flag.TYPE.INTEGER, default=1, help="The channels of audio features."),
            Flag("vocab_size", dtype=Flag.TYPE.INTEGER, default=8008,
                 help="Target vocabulary size INCLUDING <UNK>,<SEQ_BEG>,<SEQ_END> (stands in for the reference's "
                      "transcript_data_pipeline vocabulary files)."),
            Flag("transcript_data_pipeline.params", dtype=Flag.TYPE.STRING, default=None,
                 help="Parameters of the target TextDataPipeline (vocab_path, ...); overrides vocab_size."),
            Flag("max_src_len", dtype=Flag.TYPE.INTEGER, default=None, help="Maximum source length (frames)."),
            Flag("min_src_bucket_boundary", dtype=Flag.TYPE.INTEGER, default=128,
                 help="The minimum source length of the training bucket (audio frames)."),
            Flag("max_trg_len", dtype=Flag.TYPE.INTEGER, default=None, help="Maximum target length."),
            Flag("truncate_src", dtype=Flag.TYPE.BOOLEAN, default=None, help="Whether to truncate source to max_src_len."),
            Flag("truncate_trg", dtype=Flag.TYPE.BOOLEAN, default=None, help="Whether to truncate target to max_trg_len."),
            Flag("batch_size", dtype=Flag.TYPE.INTEGER, default=None, help="Global batch size in FRAMES."),
            Flag("batch_size_per_gpu", dtype=Flag.TYPE.INTEGER, default=None, help="Per-GPU batch size in FRAMES."),
            Flag("shuffle_buffer", dtype=Flag.TYPE.INTEGER, default=0, help="The buffer size for dataset shuffle."),
            Flag("experimental_frame_transcript_ratio", dtype=Flag.TYPE.INTEGER, default=None,
                 help="The ratio of the number of frames and its transcript for training batch bucket."),
            Flag("specaug", dtype=Flag.TYPE.STRING, default=None,
                 help="The arguments for spec augment: a predefined setting (LB, LD, SM, SS) or a dict of arguments."),
            Flag("disable_batch_efficiency", dtype=Flag.TYPE.BOOLEAN, default=None,
                 help="Whether to disable rounding the bucket batch sizes up to multiples of 8."),
        ]

    @property
    def trg_meta(self):
        return self._trg_meta

    def get_config(self):
        cfg = {"audio_feature_dim": self._audio_feature_dim, "audio_feature_channels": self._audio_feature_channels,
               "vocab_size": self._trg_meta["vocab_size"]}
        if self._trg_data_pipeline is not None:
            cfg["transcript_data_pipeline.class"] = "TextDataPipeline"
            cfg["transcript_data_pipeline.params"] = self._trg_data_pipeline.get_config()
        return cfg

    def get_data_preprocess_fn(self, mode, data_status, args=None):
        """speech2text.py:171-234: one example {"audio", "transcript"} -> {"audio" flat f32, "audio_length", "transcript" ids}."""
        args = dict(self._args, **(args or {}))
        trunc_audio, max_audio_len = args.get("truncate_src", None), args.get("max_src_len", None)
        trunc_trg, max_trg_len = args.get("truncate_trg", None), args.get("max_trg_len", None)
        fdim = self._audio_feature_dim * self._audio_feature_channels
        if data_status["audio"] != compat.DataStatus.PROJECTED:
            raise RuntimeError("We recommend one to preprocess the audio in advance.")

        def _process_audio(audio):
            audio = np.asarray(audio, dtype=np.float32)
            if trunc_audio and max_audio_len:
                audio = audio[:max_audio_len * fdim]
            if self._specaug is not None and mode == compat.ModeKeys.TRAIN:
                audio = self._specaug(audio.reshape(-1, fdim).copy()).reshape(-1)
            return audio

        def _process_and_truncate_text(text):
            if data_status["transcript"] == compat.DataStatus.RAW:
                if self._trg_data_pipeline is None:
                    raise RuntimeError("raw transcripts need transcript_data_pipeline.params (vocab_path)")
                text = self._trg_data_pipeline.encode(text, is_processed=False)
            else:
                assert data_status["transcript"] == compat.DataStatus.PROJECTED
                text = [int(x) for x in text]
            if mode == compat.ModeKeys.TRAIN and trunc_trg and max_trg_len and len(text) > max_trg_len:
                text = text[:(max_trg_len - 1)] + text[-1:]
            return text

        def data_proc(data, with_label):
            feature = _process_audio(data["audio"])
            ret = {"audio": feature, "audio_length": np.asarray(feature.shape[0] // fdim, dtype=np.int64)}
            if with_label:
                ret["transcript"] = np.asarray(_process_and_truncate_text(data["transcript"]), dtype=np.int64)
            return ret

        if mode == compat.ModeKeys.INFER:
            return lambda data: data_proc(data, False)
        return lambda data: data_proc(data, True)

    def create_and_batch(self, ds, mode, args=None, num_replicas_in_sync=1, shard_id=0, total_shards=1, seed=1234):
        """speech2text.py:236-384 (`create_and_batch_tfds`) as a generator of padded numpy batches.

        TRAIN: file-sharded pass -> preprocess -> length filter -> shuffle buffer -> frame buckets -> padded batches
        (audio padded to the bucket bound, transcripts to the longest of the batch or to the bucket's fixed length,
        incomplete windows dropped), repeated over epochs.  EVAL / INFER: one ordered pass, fixed number of utterances."""
        args = dict(self._args, **(args or {}))
        fdim = self._audio_feature_dim * self._audio_feature_channels
        pad = {"audio": np.float32(0), "audio_length": np.int64(0), "transcript": np.int64(self._trg_meta["pad_id"])}
        prep = self.get_data_preprocess_fn(mode, ds.status, args)
        if mode != compat.ModeKeys.TRAIN:
            bs = batching.adjust_batch_size(args.get("batch_size", None), args.get("batch_size_per_gpu", None), num_replicas_in_sync)

            def eval_gen():
                window = []
                for ex in ds.build_iterator(map_func=prep, shard_id=shard_id, total_shards=total_shards)():
                    window.append(ex)
                    if len(window) == bs:
                        yield batching.pad_batch(window, {}, pad)
                        window = []
                if window:
                    yield batching.pad_batch(window, {}, pad)
            return eval_gen()
        plan = batching.speech_bucket_plan(args.get("max_src_len", None), args.get("max_trg_len", None),
                                           args.get("min_src_bucket_boundary", None), args.get("batch_size", None),
                                           args.get("batch_size_per_gpu", None), num_replicas_in_sync,
                                           bool(args.get("disable_batch_efficiency", None)),
                                           args.get("experimental_frame_transcript_ratio", None))
        bounds, sizes, trans = plan["audio_bounds"], plan["batch_sizes"], plan["trans_bounds"]
        nb = len(bounds)
        # every rank forms ITS share of the reference's global batch (one process per GPU)
        local_sizes = [max(1, b // num_replicas_in_sync) for b in sizes]
        limits = {"audio": args["max_src_len"] * fdim, "audio_length": -1, "transcript": args["max_trg_len"]}

        def key_fn(ex):
            n = int(ex["audio_length"])
            if trans is None:
                for b, bound in enumerate(bounds):
                    if n <= bound:
                        return b
                return None
            tl = int(ex["transcript"].shape[0])
            for b, bound in enumerate(bounds):
                for j in range(2):
                    if n <= bound and tl <= trans[b][j]:
                        return b * nb + j
            return None

        def window_size_fn(key):
            return local_sizes[key] if trans is None else local_sizes[key // nb]

        def padded_lengths_fn(key):
            if trans is None:
                return {"audio": bounds[key] * fdim, "transcript": None}
            return {"audio": bounds[key // nb] * fdim, "transcript": trans[key // nb][key % nb]}

        def train_gen():
            rng = np.random.RandomState(seed + shard_id)
            epoch = 0
            while True:
                stream = ds.build_iterator(map_func=prep, shard_id=shard_id, total_shards=total_shards, shuffle=True, epoch=epoch)()
                stream = batching.clean_by_length(stream, limits)
                stream = batching.shuffle_buffer(stream, args.get("shuffle_buffer", 0), rng)
                n = 0
                for batch in batching.group_by_window_padded_batch(stream, key_fn, window_size_fn, padded_lengths_fn, pad):
                    n += 1
                    yield batch
                if n == 0:
                    raise RuntimeError("the training data produced no complete batch: check batch_size / max_src_len")
                epoch += 1
        return train_gen()

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
