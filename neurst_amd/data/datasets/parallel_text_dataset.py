"""Parallel text datasets of the Seq2Seq / Translation task (neurst/data/datasets/parallel_text_dataset.py:34-306).

  parallel_tfrecord  TFRecord shards of {"feature": int64 ids, "label": int64 ids} (already projected, EOS included) -- what
                     neurst/cli/create_tfrecords.py writes and the recipes train from; read without TensorFlow through
                     neurst_amd/data/tfrecord.py with the file set / rank sharding / interleave of `load_tfrecords`;
  parallel_text      a source and a target text file, one sentence per line (`data_is_processed`: already tokenised /
                     sub-tokenised); empty lines are kept so the two sides stay aligned, the task's length filter drops them.
Both yield single examples {"feature", "label"}; the task turns them into ids, buckets and pads (Seq2Seq.create_and_batch).
"""
import numpy as np

from neurst_amd.data import tfrecord
from neurst_amd.data.datasets.dataset import Dataset, register_dataset
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


@register_dataset(["parallel_tfrecord", "ParallelTFRecordDataset"])
class ParallelTFRecordDataset(Dataset):
    batched = False

    def __init__(self, args):
        super().__init__()
        self._data_path = args["data_path"]
        self._shuffle_dataset = bool(args.get("shuffle_dataset", False))
        self._seed = args.get("seed", None) or 1234
        if not tfrecord.list_record_files(self._data_path):
            raise ValueError(f"Fail to read {self._data_path}")

    @staticmethod
    def class_or_method_args():
        return [Flag("data_path", dtype=Flag.TYPE.STRING, help="TFRecord file, directory (dir/*train*) or path prefix."),
                Flag("shuffle_dataset", dtype=Flag.TYPE.BOOLEAN, default=None, help="Shuffle the file list when training."),
                Flag("seed", dtype=Flag.TYPE.INTEGER, default=1234, help="Seed of the file shuffle.")]

    @property
    def status(self):
        return compat.DataStatus.PROJECTED

    def files(self, shard_id=0, total_shards=1, shuffle=False, epoch=0):
        files = tfrecord.list_record_files(self._data_path)
        if total_shards > 1:
            files = files[shard_id::total_shards]
        if shuffle:
            order = np.random.RandomState(self._seed + epoch).permutation(len(files))
            files = [files[i] for i in order]
        return files

    def build_iterator(self, map_func=None, shard_id=0, total_shards=1, shuffle=False, epoch=0, **unused):
        def gen():
            for record in tfrecord.interleave_records(self.files(shard_id, total_shards, shuffle, epoch)):
                ex = tfrecord.parse_example(record)
                data = {"feature": np.asarray(ex["feature"][1], dtype=np.int64), "label": np.asarray(ex["label"][1], dtype=np.int64)}
                yield map_func(data) if map_func is not None else data
        return gen


@register_dataset(["parallel_text", "ParallelTextDataset"])
class ParallelTextDataset(Dataset):
    batched = False

    def __init__(self, args):
        super().__init__()
        self._src_file, self._trg_file = args["src_file"], args.get("trg_file", None)
        self._data_is_processed = bool(args.get("data_is_processed", False))
        self._targets = None

    @staticmethod
    def class_or_method_args():
        return [Flag("src_file", dtype=Flag.TYPE.STRING, help="The source text file."),
                Flag("trg_file", dtype=Flag.TYPE.STRING, default=None, help="The target text file."),
                Flag("data_is_processed", dtype=Flag.TYPE.BOOLEAN, default=None,
                     help="Whether the text data is already processed (tokenised and sub-tokenised).")]

    @property
    def status(self):
        return compat.DataStatus.PROCESSED if self._data_is_processed else compat.DataStatus.RAW

    def build_iterator(self, map_func=None, shard_id=0, total_shards=1, **unused):
        """parallel_text_dataset.py:110-158: line n of both files (whitespace normalised); with shards, contiguous ranges of
        num_samples // total_shards lines by the reference's 1-based counter (shard s takes n in [s * per, (s + 1) * per),
        the last shard runs to the end)."""
        begin, end = 0, None
        if total_shards > 1:
            with open(self._src_file, encoding="utf-8") as fp:
                total = sum(1 for _ in fp)
            per = total // total_shards
            begin = per * shard_id
            end = total + 1 if shard_id == total_shards - 1 else begin + per

        def gen():
            trg = open(self._trg_file, encoding="utf-8") if self._trg_file else None
            with open(self._src_file, encoding="utf-8") as src:
                n = 0
                for line in src:
                    n += 1
                    data = {"feature": " ".join(line.strip().split())}
                    if trg is not None:
                        data["label"] = " ".join(trg.readline().strip().split())
                    if end is not None:
                        if n < begin:
                            continue
                        if n >= end:
                            break
                    yield map_func(data) if map_func is not None else data
            if trg is not None:
                trg.close()
        return gen

    @property
    def targets(self):
        if self._targets is None and self._trg_file:
            with open(self._trg_file, encoding="utf-8") as fp:
                self._targets = [line.strip() for line in fp]
        return self._targets
