"""`audio_tfrecord` dataset (neurst/data/datasets/audio/audio_dataset.py:248-363): speech-to-text examples stored as
tf.train.Example records {feature_key: float_list (extracted features, flattened [frames * dim * channels]) or int64_list
(raw samples), transcript_key: bytes_list (raw text) or int64_list (already projected ids), "src_lang", "uuid"}.

Reading goes through neurst_amd/data/tfrecord.py (no TensorFlow): the file set, rank sharding at FILE level
(`files.shard(num_workers, worker_id)`, dataset_utils.py:295-306) and the 10-way record interleave follow
`load_tfrecords`.  Elements come out as {"audio": float32 / int64 array, "transcript": int64 array | str,
"src_lang": str, "uuid": str} -- the reference's `feature_name_mapping` to "audio" / "transcript" applied.
"""
import numpy as np

from neurst_amd.data import tfrecord
from neurst_amd.data.datasets.dataset import Dataset, register_dataset
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


@register_dataset(["audio_tfrecord", "AudioTFRecordDataset"])
class AudioTFRecordDataset(Dataset):
    batched = False  # yields single examples: the task buckets and pads them (SpeechToText.create_and_batch)

    def __init__(self, args):
        super().__init__()
        self._data_path = args["data_path"]
        self._shuffle_dataset = bool(args.get("shuffle_dataset", False))
        self._feature_key = args.get("feature_key", None) or "audio"
        self._transcript_key = args.get("transcript_key", None) or "transcript"
        self._seed = args.get("seed", None) or 1234
        files = tfrecord.list_record_files(self._data_path)
        if not files:
            raise ValueError(f"Fail to read {self._data_path}")
        first = tfrecord.parse_example(next(tfrecord.read_records(files[0])))
        kind, _ = first.get(self._feature_key, (None, []))
        if kind == "float":
            self._audio_is_extracted = True
        elif kind == "int64":
            self._audio_is_extracted = False
        else:
            raise ValueError(f"record of {files[0]} has no float / int64 feature '{self._feature_key}'")
        self._transcript_is_projected = first.get(self._transcript_key, (None, []))[0] == "int64"
        self._targets = None

    @staticmethod
    def class_or_method_args():
        return [Flag("data_path", dtype=Flag.TYPE.STRING, help="TFRecord file, directory (dir/*train*) or path prefix; comma separated list allowed."),
                Flag("shuffle_dataset", dtype=Flag.TYPE.BOOLEAN, default=None, help="Shuffle the file list when training."),
                Flag("feature_key", dtype=Flag.TYPE.STRING, default="audio", help="The key of the audio features in the TF Record."),
                Flag("transcript_key", dtype=Flag.TYPE.STRING, default="transcript",
                     help="The key of the audio transcript/translation in the TF Record."),
                Flag("seed", dtype=Flag.TYPE.INTEGER, default=1234, help="Seed of the file shuffle.")]

    @property
    def status(self):
        return {"audio": compat.DataStatus.PROJECTED if self._audio_is_extracted else compat.DataStatus.RAW,
                "transcript": compat.DataStatus.PROJECTED if self._transcript_is_projected else compat.DataStatus.RAW}

    def files(self, shard_id=0, total_shards=1, shuffle=False, epoch=0):
        files = tfrecord.list_record_files(self._data_path)
        if total_shards > 1:
            files = files[shard_id::total_shards]
        if shuffle:
            order = np.random.RandomState(self._seed + epoch).permutation(len(files))
            files = [files[i] for i in order]
        return files

    def _element(self, record):
        ex = tfrecord.parse_example(record)

        def text(key):
            kind, vals = ex.get(key, (None, []))
            if kind != "bytes" or not vals:
                return ""
            return vals[0].decode("utf-8")
        kind, audio = ex.get(self._feature_key, (None, np.zeros(0, np.float32)))
        tkind, tr = ex.get(self._transcript_key, (None, []))
        if tkind == "bytes":
            tr = tr[0].decode("utf-8") if tr else ""
        elif tkind is None:
            tr = np.zeros(0, np.int64)
        return {"audio": np.asarray(audio), "transcript": tr, "src_lang": text("src_lang"), "uuid": text("uuid")}

    def build_iterator(self, map_func=None, shard_id=0, total_shards=1, shuffle=False, epoch=0, **unused):
        """One pass over this shard's records (audio_dataset.py:344-360); `shuffle` permutes the FILE order only, as the
        reference does for training (the example-level shuffle is the task's shuffle buffer)."""
        def gen():
            for record in tfrecord.interleave_records(self.files(shard_id, total_shards, shuffle, epoch)):
                data = self._element(record)
                if map_func is not None:
                    data = map_func(data)
                yield data
        return gen

    @property
    def targets(self):
        if self._targets is None:
            assert not self._transcript_is_projected
            self._targets = [x["transcript"] for x in self.build_iterator()()]
        return self._targets
