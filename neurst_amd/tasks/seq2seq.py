"""Seq2Seq / Translation task (neurst/tasks/seq2seq.py:39-283, neurst/tasks/translation.py:59-143): input conventions of
the text Transformer (§8(f) rank 1) and its training data feed (rank 2): per-example preprocessing (ids, truncation;
:146-191) and the token-bucketed, padded batching of `create_and_batch_tfds` (:193-271) as plain-Python iterators.
Configured either with vocabulary files (`src/trg_data_pipeline.params: {vocab_path: ...}` -> TextDataPipeline) or with
vocabulary SIZES; the special ids follow the pipeline's convention (<UNK>, <SEQ_BEG>, <SEQ_END> appended at the end,
pad_id == eos_id: data_pipelines/text_data_pipeline.py:71-92).  The tokenizers themselves are out of scope."""
import numpy as np
import torch

from neurst_amd.data import batching
from neurst_amd.data.text_pipeline import TextDataPipeline

from neurst_amd.models import build_model
from neurst_amd.models.model_utils import deduce_text_length
from neurst_amd.tasks.task import Task, register_task
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


def _text_meta(vocab_size):
    return {"vocab_size": vocab_size, "unk_id": vocab_size - 3, "bos_id": vocab_size - 2, "eos_id": vocab_size - 1,
            "pad_id": vocab_size - 1, "padding_mode": compat.PaddingMode.EOS_AS_PADDING}


@register_task(["seq2seq", "translation", "Translation"])
class Seq2Seq(Task):
    def __init__(self, args):
        super().__init__(args)
        self._args = dict(args)
        self._src_data_pipeline = self._trg_data_pipeline = None
        sp = args.get("src_data_pipeline.params", None) or {}
        tp = args.get("trg_data_pipeline.params", None) or {}
        if sp.get("vocab_path", None):
            self._src_data_pipeline = TextDataPipeline(**sp)
        if tp.get("vocab_path", None):
            self._trg_data_pipeline = TextDataPipeline(**tp)
        self._src_meta = self._src_data_pipeline.meta if self._src_data_pipeline else _text_meta(args.get("src_vocab_size", None) or 32003)
        self._trg_meta = self._trg_data_pipeline.meta if self._trg_data_pipeline else _text_meta(args.get("trg_vocab_size", None) or 32003)
        tb = args.get("target_begin_of_sentence", None) or "bos"
        assert tb in ("bos", "eos"), "target_begin_of_sentence must be 'bos' or 'eos'"
        self._target_begin_of_sentence = tb

    @staticmethod
    def class_or_method_args():
        return [
            Flag("src_vocab_size", dtype=Flag.TYPE.INTEGER, default=32003,
                 help="Source vocabulary size INCLUDING <UNK>,<SEQ_BEG>,<SEQ_END>."),
            Flag("trg_vocab_size", dtype=Flag.TYPE.INTEGER, default=32003,
                 help="Target vocabulary size INCLUDING <UNK>,<SEQ_BEG>,<SEQ_END>."),
            Flag("target_begin_of_sentence", dtype=Flag.TYPE.STRING, default="bos", choices=["bos", "eos"],
                 help="The begin of sentence symbol for target side."),
            Flag("max_src_len", dtype=Flag.TYPE.INTEGER, default=80, help="The maximum source length of training data."),
            Flag("max_trg_len", dtype=Flag.TYPE.INTEGER, default=80, help="The maximum target length of training data."),
            Flag("batch_size", dtype=Flag.TYPE.INTEGER, default=None, help="Global batch size in tokens."),
            Flag("batch_size_per_gpu", dtype=Flag.TYPE.INTEGER, default=None, help="Per-GPU batch size in tokens."),
            Flag("batch_by_tokens", dtype=Flag.TYPE.BOOLEAN, default=True, help="Whether `batch_size` counts tokens (else sentences)."),
            Flag("truncate_src", dtype=Flag.TYPE.BOOLEAN, default=None, help="Whether to truncate source to max_src_len."),
            Flag("truncate_trg", dtype=Flag.TYPE.BOOLEAN, default=None, help="Whether to truncate target to max_trg_len."),
            Flag("shuffle_buffer", dtype=Flag.TYPE.INTEGER, default=0, help="The buffer size for dataset shuffle."),
            Flag("src_data_pipeline.params", dtype=Flag.TYPE.STRING, default=None,
                 help="Parameters of the source TextDataPipeline (vocab_path, ...); overrides src_vocab_size."),
            Flag("trg_data_pipeline.params", dtype=Flag.TYPE.STRING, default=None,
                 help="Parameters of the target TextDataPipeline (vocab_path, ...); overrides trg_vocab_size."),
        ]

    @property
    def src_meta(self):
        return self._src_meta

    @property
    def trg_meta(self):
        return self._trg_meta

    def get_config(self):
        cfg = {"src_vocab_size": self._src_meta["vocab_size"], "trg_vocab_size": self._trg_meta["vocab_size"],
               "target_begin_of_sentence": self._target_begin_of_sentence}
        for side, dp in (("src", self._src_data_pipeline), ("trg", self._trg_data_pipeline)):
            if dp is not None:
                cfg[f"{side}_data_pipeline.class"] = "TextDataPipeline"
                cfg[f"{side}_data_pipeline.params"] = dp.get_config()
        return cfg

    def build_model(self, args, name=None, **kwargs):
        return build_model(args, self._src_meta, self._trg_meta, name=name, **kwargs)

    def get_data_preprocess_fn(self, mode, data_status=compat.DataStatus.RAW, args=None):
        """seq2seq.py:146-191: text -> ids (+ EOS) unless already projected; in training an over-long side keeps its first
        max_len - 1 ids and its last one (the EOS) when truncation is on."""
        args = dict(self._args, **(args or {}))

        def proc(text, dp, trunc, max_len, side):
            if data_status != compat.DataStatus.PROJECTED:
                if dp is None:
                    raise RuntimeError(f"text input needs {side}_data_pipeline.params (vocab_path)")
                text = dp.encode(text, is_processed=(data_status == compat.DataStatus.PROCESSED))
            else:
                text = [int(x) for x in text]
            if mode == compat.ModeKeys.TRAIN and trunc and max_len and len(text) > max_len:
                text = text[:(max_len - 1)] + text[-1:]
            return np.asarray(text, dtype=np.int64)

        def fn(data):
            out = {"feature": proc(data["feature"], self._src_data_pipeline, args.get("truncate_src", None),
                                   args.get("max_src_len", None), "src")}
            if mode != compat.ModeKeys.INFER:
                out["label"] = proc(data["label"], self._trg_data_pipeline, args.get("truncate_trg", None),
                                    args.get("max_trg_len", None), "trg")
            return out
        return fn

    def create_and_batch(self, ds, mode, args=None, num_replicas_in_sync=1, shard_id=0, total_shards=1, seed=1234):
        """seq2seq.py:193-271 (`create_and_batch_tfds`) as a generator of padded numpy batches {"feature", "label"}.
        TRAIN: length filter -> shuffle buffer -> (source, target) length buckets -> batches padded to the bucket bounds,
        incomplete windows dropped, repeated over epochs.  EVAL / INFER: ordered, `batch_size` sentences, padded to the longest."""
        args = dict(self._args, **(args or {}))
        pad = {"feature": np.int64(self._src_meta["pad_id"]), "label": np.int64(self._trg_meta["pad_id"])}
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
        plan = batching.text_bucket_plan(args.get("max_src_len", None), args.get("max_trg_len", None), args.get("batch_size", None),
                                         args.get("batch_size_per_gpu", None), args.get("batch_by_tokens", True) is not False,
                                         num_replicas_in_sync)
        sb, tb = plan["src_bounds"], plan["trg_bounds"]
        local = [max(1, n // num_replicas_in_sync) for n in plan["batch_sizes"]]
        limits = {"feature": args["max_src_len"], "label": args["max_trg_len"]}

        def key_fn(ex):
            s, t = ex["feature"].shape[0], ex["label"].shape[0]
            for b in range(len(sb)):
                if s <= sb[b] and t <= tb[b]:
                    return b
            return None

        def train_gen():
            rng = np.random.RandomState(seed + shard_id)
            epoch = 0
            while True:
                stream = ds.build_iterator(map_func=prep, shard_id=shard_id, total_shards=total_shards, shuffle=True, epoch=epoch)()
                stream = batching.clean_by_length(stream, limits)
                stream = batching.shuffle_buffer(stream, args.get("shuffle_buffer", 0), rng)
                n = 0
                for batch in batching.group_by_window_padded_batch(stream, key_fn, lambda b: local[b],
                                                                   lambda b: {"feature": sb[b], "label": tb[b]}, pad):
                    n += 1
                    yield batch
                if n == 0:
                    raise RuntimeError("the training data produced no complete batch: check batch_size / max_src_len")
                epoch += 1
        return train_gen()

    def example_to_input(self, batch_of_data, mode):
        """seq2seq.py:110-136.  batch_of_data: {"feature" [B,S] int64, "label" [B,L] int64}."""
        feat = batch_of_data["feature"]
        input_dict = {"src": feat,
                      "src_length": deduce_text_length(feat, self._src_meta["pad_id"], self._src_meta["padding_mode"])}
        bosid = self._trg_meta["eos_id"] if self._target_begin_of_sentence == "eos" else self._trg_meta["bos_id"]
        bos = torch.full((feat.shape[0],), bosid, dtype=torch.int64, device=feat.device)
        if mode == compat.ModeKeys.INFER:
            input_dict["trg_input"] = bos
        else:
            lab = batch_of_data["label"]
            input_dict["trg"] = lab
            input_dict["trg_length"] = deduce_text_length(lab, self._trg_meta["pad_id"], self._trg_meta["padding_mode"])
            input_dict["trg_input"] = torch.cat([bos[:, None], lab[:, :-1]], dim=1)
        return input_dict


@register_task(["waitk_translation", "WaitkTranslation"])
class WaitkTranslation(Seq2Seq):
    """neurst/tasks/waitk_translation.py:22-56: Translation whose models are built with the wait-k lagging (`wait_k`: an int
    or a list of laggings for multi-path training)."""

    def __init__(self, args):
        super().__init__(args)
        k = args.get("wait_k", None)
        if isinstance(k, str):
            import yaml
            k = yaml.safe_load(k)
        assert k, "Must provide wait_k as the decode lagging."
        assert isinstance(k, (list, int)), f"Value error: {k}"
        self._wait_k = k

    def get_config(self):
        cfg = super().get_config()
        cfg["wait_k"] = self._wait_k
        return cfg

    @staticmethod
    def class_or_method_args():
        return Seq2Seq.class_or_method_args() + [Flag("wait_k", dtype=Flag.TYPE.STRING, default=None, help="The lagging k.")]

    def build_model(self, args, name=None, **kwargs):
        return super().build_model(args, name=name, waitk_lagging=self._wait_k, **kwargs)
