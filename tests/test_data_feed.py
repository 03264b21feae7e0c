"""Data-feed row: TFRecord / tf.train.Example codec, vocabulary ids, SpecAugment, frame-bucketed
batching -- against records TensorFlow wrote, the reference's own functions (tests/golden/make_golden_data.py) and
google.protobuf.  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pb_example import example_message_class  # noqa: E402

from neurst_amd.data import batching, tfrecord  # noqa: E402
from neurst_amd.data.datasets import build_dataset  # noqa: E402
from neurst_amd.data.text_pipeline import TextDataPipeline  # noqa: E402
from neurst_amd.tasks import build_task  # noqa: E402
from neurst_amd.utils import compat  # noqa: E402
from neurst_amd.utils.audio_lib import SpecAugment  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------------------------------------ CRC-32C / framing
def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors + the classic check value
    assert tfrecord.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert tfrecord.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tfrecord.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfrecord.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert tfrecord.crc32c(b"123456789") == 0xE3069283
    assert tfrecord.crc32c(b"") == 0
    # the library's slicing-by-8 routine (nst_crc32c) and the Python table walk agree, also across chunk boundaries
    rng = np.random.RandomState(0)
    blob = rng.bytes(5000)
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 4999):
        for off in (0, 1, 3):
            assert tfrecord.crc32c(blob[off:off + n]) == tfrecord.crc32c_py(blob[off:off + n])
    assert tfrecord.crc32c(blob[100:], tfrecord.crc32c(blob[:100])) == tfrecord.crc32c_py(blob)
    assert tfrecord.crc32c_py(b"123456789") == 0xE3069283


def _pipeline(gold, side):
    """TextDataPipeline whose vocabulary holds the fixture's tokens at their real ids (the rest are placeholders)."""
    n = int(gold[f"vocab_size_{side}"])
    tokens = [f"<tok{i}>" for i in range(n)]
    for i, t in zip(gold[f"{side}_ids"], gold[f"{side}_tokens"]):
        tokens[int(i)] = str(t)
    return TextDataPipeline(vocab_path=tokens)


def test_tensorflow_written_records():
    """tests/golden/tfrecord_seq2seq_head.bin = the first 12 records of the reference's tests/examples/train.tfrecords-00000-of-00004,
    bytes TensorFlow wrote: both CRCs of every record verify, the Example decodes, and the int64 lists are exactly the
    vocabulary ids (+ EOS) of the text lines the records were made from."""
    gold = np.load(os.path.join(GOLD, "tfrecord_seq2seq_head.npz"))
    recs = list(tfrecord.read_records(os.path.join(GOLD, "tfrecord_seq2seq_head.bin"), check_crc=True))
    assert len(recs) == int(gold["nrec"]) == 12
    src_pipe, trg_pipe = _pipeline(gold, "src"), _pipeline(gold, "trg")
    assert trg_pipe.meta["vocab_size"] == int(gold["vocab_size_trg"]) + 3
    assert trg_pipe.meta["eos_id"] == trg_pipe.meta["pad_id"] == trg_pipe.meta["vocab_size"] - 1
    assert trg_pipe.meta["padding_mode"] == compat.PaddingMode.EOS_AS_PADDING
    Example = example_message_class()
    for rec, src_line, trg_line in zip(recs, gold["src_lines"], gold["trg_lines"]):
        ex = tfrecord.parse_example(rec)
        assert sorted(ex) == ["feature", "label"]
        assert ex["feature"][0] == ex["label"][0] == "int64"
        assert ex["feature"][1].tolist() == src_pipe.encode(str(src_line), is_processed=True)
        assert ex["label"][1].tolist() == trg_pipe.encode(str(trg_line), is_processed=True)
        assert trg_pipe.decode(ex["label"][1]) == str(trg_line)
        msg = Example()
        msg.ParseFromString(rec)  # the independent decoder agrees
        assert list(msg.features.feature["label"].int64_list.value) == ex["label"][1].tolist()
        # and re-encoding the parsed values reproduces TensorFlow's bytes exactly
        assert tfrecord.encode_example({"feature": ex["feature"][1], "label": ex["label"][1]}) == rec
    # re-framing reproduces the file byte for byte
    assert b"".join(tfrecord.frame_record(r) for r in recs) == open(os.path.join(GOLD, "tfrecord_seq2seq_head.bin"), "rb").read()


def test_corruption_is_detected(tmp_path):
    raw = bytearray(open(os.path.join(GOLD, "tfrecord_seq2seq_head.bin"), "rb").read())
    raw[40] ^= 0x01
    p = tmp_path / "bad.tfrecords"
    p.write_bytes(bytes(raw))
    with pytest.raises(tfrecord.TFRecordError):
        list(tfrecord.read_records(str(p)))
    assert len(list(tfrecord.read_records(str(p), check_crc=False))) == 12
    p.write_bytes(bytes(raw[:100]))
    with pytest.raises(tfrecord.TFRecordError):
        list(tfrecord.read_records(str(p), check_crc=False))


def test_example_codec_against_protobuf():
    Example = example_message_class()
    rng = np.random.RandomState(0)
    for trial in range(20):
        audio = rng.randn(rng.randint(0, 300)).astype(np.float32)
        ids = rng.randint(-5, 1 << 40, size=rng.randint(0, 40)).astype(np.int64)
        texts = [("utt-%d" % trial).encode(), "übung ✓".encode("utf-8")][:rng.randint(1, 3)]
        msg = Example()
        msg.features.feature["audio"].float_list.value.extend(audio.tolist())
        msg.features.feature["transcript"].int64_list.value.extend(ids.tolist())
        msg.features.feature["uuid"].bytes_list.value.extend(texts)
        got = tfrecord.parse_example(msg.SerializeToString())
        assert got["audio"][0] == "float" and np.array_equal(got["audio"][1], audio)
        assert got["transcript"][0] == "int64" and np.array_equal(got["transcript"][1], ids)
        assert got["uuid"] == ("bytes", texts)
        back = Example()
        back.ParseFromString(tfrecord.encode_example({"audio": audio, "transcript": ids, "uuid": texts}))
        assert np.array_equal(np.asarray(back.features.feature["audio"].float_list.value, np.float32), audio)
        assert list(back.features.feature["transcript"].int64_list.value) == ids.tolist()
        assert list(back.features.feature["uuid"].bytes_list.value) == texts
    # unpacked repeated scalars (legal protobuf, older writers) are accepted too
    ld = tfrecord._ld
    unpacked = ld(1, ld(1, ld(1, b"a") + ld(2, ld(3, b"\x08\x01\x08\x02\x08\xff\xff\xff\xff\x0f"))))
    assert tfrecord.parse_example(unpacked)["a"][1].tolist() == [1, 2, 0xFFFFFFFF]


def test_file_set_and_interleave(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    counts = {}
    for i in range(13):
        n = 3 + (i % 4)
        counts[i] = n
        tfrecord.write_records(str(d / f"train.tfrecords-{i:05d}-of-00013"),
                               [tfrecord.encode_example({"k": [i * 100 + j]}) for j in range(n)])
    tfrecord.write_records(str(d / "dev.tfrecords"), [tfrecord.encode_example({"k": [-1]})])
    files = tfrecord.list_record_files(str(d))                       # a directory means dir/*train*
    assert [os.path.basename(f) for f in files] == [f"train.tfrecords-{i:05d}-of-00013" for i in range(13)]
    assert tfrecord.list_record_files(str(d / "train.tfrecords-0000")) == files[:10]      # a prefix means prefix*
    assert tfrecord.list_record_files(str(d / "dev.tfrecords")) == [str(d / "dev.tfrecords")]
    assert tfrecord.list_record_files(f"{d}/dev.tfrecords, {d}/train.tfrecords-00012") == [str(d / "dev.tfrecords"), files[12]]
    got = [int(tfrecord.parse_example(r)["k"][1][0]) for r in tfrecord.interleave_records(files)]
    # model of Dataset.interleave(cycle_length=10, block_length=1)
    its = [iter([i * 100 + j for j in range(counts[i])]) for i in range(13)]
    slots, nxt, want, i = its[:10], 10, [], 0
    while slots:
        if i >= len(slots):
            i = 0
        try:
            want.append(next(slots[i]))
            i += 1
        except StopIteration:
            if nxt < 13:
                slots[i] = its[nxt]
                nxt += 1
            else:
                slots.pop(i)
    assert got == want and sorted(got) == sorted(i * 100 + j for i in range(13) for j in range(counts[i]))
    assert got[:11] == [0, 100, 200, 300, 400, 500, 600, 700, 800, 900, 1]


# ------------------------------------------------------------------------------------------------ SpecAugment
@pytest.mark.parametrize("tag", ["LB", "LD", "SM", "SS", "LB_short", "custom"])
def test_specaugment_matches_reference(tag):
    """Outputs of the reference's SpecAugment class under the same numpy seed (masked cells + the fill value)."""
    g = np.load(os.path.join(GOLD, f"specaug_{tag}.npz"))
    shape = tuple(int(v) for v in g["shape"])
    x = np.random.RandomState(int(g["x_seed"])).randn(*shape).astype(np.float32)
    if tag == "custom":
        aug = SpecAugment.build("{time_wrap_w: 0, freq_mask_n: 3, freq_mask_f: 10, time_mask_n: 4, time_mask_t: 30, "
                                "time_mask_p: 0.5, mask_value: 0.25}")
    else:
        aug = SpecAugment.build(tag.split("_")[0])
    np.random.seed(int(g["seed"]))
    y = aug(x.copy())
    mask = np.unpackbits(g["mask"])[:x.size].reshape(shape).astype(bool)
    assert int(mask.sum()) == int(g["n_masked"])
    want = np.where(mask, g["mask_value"], x)
    assert np.array_equal(y, want)
    if tag == "LB_short":
        assert mask.sum() == 0 or mask.any(axis=0).sum() <= 27   # 60 frames < time_mask_t = 100: no time mask at all
    assert SpecAugment.build(None) is None and SpecAugment.build("nonsense") is None


# ------------------------------------------------------------------------------------------------ bucket arithmetic
def test_bucket_boundaries_match_reference():
    g = np.load(os.path.join(GOLD, "bucket_boundaries.npz"))
    for k in g.files:
        if not k.startswith("case"):
            continue
        mx, mn = int(g[k][0]), int(g[k][1])
        assert batching.create_audio_bucket_boundaries(mx, None if mn < 0 else mn) == g[k][2:].tolist()
    for v, f, want in g["minimal_multiple"]:
        assert batching.minimal_multiple(int(v), int(f)) == int(want)


def test_bucket_plan_of_the_mustc_recipe():
    """examples/speech_transformer/must-c/st_training_args.yml: batch_size 80000 frames, max_src_len 3000, max_trg_len 150,
    experimental_frame_transcript_ratio 12 (speech2text.py:293-336 evaluated by hand)."""
    plan = batching.speech_bucket_plan(3000, 150, 128, 80000, None, 1, False, 12)
    b, s, t = plan["audio_bounds"], plan["batch_sizes"], plan["trans_bounds"]
    assert b[0] == 128 and b[-1] == 3008 and b == sorted(b) and len(b) == len(s) == len(t)
    assert s[0] == 632 and s[-1] == 32            # ceil8(80000 // 128), ceil8(80000 // 3008)
    assert all(x % 8 == 0 for x in s)
    assert t[0][0] == 16                          # ceil8(min(int(128 / 12), 152))
    assert t[-1] == [152, 152] and all(p[1] >= p[0] for p in t)
    plain = batching.speech_bucket_plan(3000, 150, 128, None, 20000, 4, True, None)
    assert plain["trans_bounds"] is None and plain["batch_sizes"][0] == (20000 // 128) * 4
    with pytest.raises(AssertionError):
        batching.speech_bucket_plan(3000, 150, 128, 2000, None, 1)   # per-GPU batch must exceed max_src_len
    with pytest.raises(RuntimeError):
        batching.speech_bucket_plan(None, 150, 128, 80000, None, 1)


def test_clean_shuffle_window():
    exs = [{"audio": np.zeros(n * 4, np.float32), "audio_length": np.int64(n), "transcript": np.arange(t)}
           for n, t in [(5, 3), (50, 3), (5, 1), (5, 9), (0, 3), (7, 2)]]
    kept = list(batching.clean_by_length(iter(exs), {"audio": 40, "audio_length": -1, "transcript": 8}))
    assert [int(e["audio_length"]) for e in kept] == [5, 7]    # too long / 1-token transcript / too long transcript / empty audio
    rng = np.random.RandomState(0)
    out = list(batching.shuffle_buffer(iter(range(100)), 16, rng))
    assert sorted(out) == list(range(100)) and out != list(range(100))
    assert all(abs(v - i) <= 100 for i, v in enumerate(out)) and max(out[:10]) < 26   # only the buffer's reach is mixed
    assert list(batching.shuffle_buffer(iter(range(5)), 0, rng)) == list(range(5))


# ------------------------------------------------------------------------------------------------ end to end
def _write_speech_shards(d, n_files=4, per_file=40, fdim=8, seed=0, projected=True):
    rng = np.random.RandomState(seed)
    uid = 0
    total = {}
    for i in range(n_files):
        recs = []
        for _ in range(per_file):
            frames = int(rng.randint(20, 400))
            tr = rng.randint(0, 50, size=max(2, frames // 12)).astype(np.int64)
            tr[-1] = 52  # EOS of a 50 + 3 vocabulary
            feats = {"audio": (rng.randn(frames * fdim) + uid).astype(np.float32), "uuid": [f"utt{uid}"], "src_lang": ["en"],
                     "translation": tr if projected else [" ".join(f"w{t}" for t in tr[:-1])]}
            total[f"utt{uid}"] = (frames, tr)
            recs.append(tfrecord.encode_example(feats))
            uid += 1
        tfrecord.write_records(os.path.join(d, f"train.tfrecords-{i:05d}-of-{n_files:05d}"), recs)
    return total


def _task_and_dataset(d, **task_params):
    params = {"audio_feature_dim": 8, "audio_feature_channels": 1, "vocab_size": 53, "max_src_len": 300, "max_trg_len": 30,
              "batch_size": 2000, "min_src_bucket_boundary": 64, "truncate_src": False}
    params.update(task_params)
    task = build_task({"task.class": "SpeechToText", "task.params": params})
    ds = build_dataset({"dataset.class": "AudioTFRecordDataset",
                        "dataset.params": {"data_path": d, "shuffle_dataset": True, "feature_key": "audio", "transcript_key": "translation"}})
    return task, ds


def test_audio_tfrecord_dataset_and_bucketed_batches(tmp_path):
    d = str(tmp_path)
    total = _write_speech_shards(d)
    task, ds = _task_and_dataset(d)
    assert ds.status == {"audio": compat.DataStatus.PROJECTED, "transcript": compat.DataStatus.PROJECTED}
    # file-level sharding: disjoint, complete
    seen = []
    for r in range(2):
        ids = [e["uuid"] for e in ds.build_iterator(shard_id=r, total_shards=2)()]
        assert len(ids) == 80
        seen += ids
    assert sorted(seen) == sorted(total)
    first = next(ds.build_iterator()())
    assert first["uuid"] == "utt0" and first["src_lang"] == "en" and first["audio"].dtype == np.float32
    assert first["audio"].shape[0] == total["utt0"][0] * 8 and np.array_equal(first["transcript"], total["utt0"][1])

    plan = batching.speech_bucket_plan(300, 30, 64, 2000, None, 1)
    bounds, sizes = plan["audio_bounds"], plan["batch_sizes"]
    it = task.create_and_batch(ds, compat.ModeKeys.TRAIN, seed=3)
    n_utts = 0
    for _ in range(12):
        b = next(it)
        B = b["audio"].shape[0]
        bucket = bounds.index(b["audio"].shape[1] // 8)
        assert B == sizes[bucket] and b["audio"].shape[1] == bounds[bucket] * 8      # padded to the bucket bound
        lo = bounds[bucket - 1] if bucket else 0
        assert (b["audio_length"] <= bounds[bucket]).all() and (b["audio_length"] > lo).all() and (b["audio_length"] <= 300).all()
        assert b["transcript"].shape[1] == max(int((row != 52).sum()) + 1 for row in b["transcript"])  # longest of the batch
        for row, n in zip(b["audio"], b["audio_length"]):
            assert (row[int(n) * 8:] == 0).all() and row[:int(n) * 8].any()
        for row in b["transcript"]:
            k = int((row != 52).sum())
            assert row[k] == 52 and (row[k:] == 52).all()                             # EOS, then EOS as padding
        n_utts += B
        x = task.example_to_input({k: torch.from_numpy(v) for k, v in b.items()}, compat.ModeKeys.TRAIN)
        assert x["src"].shape == (B, bounds[bucket], 8, 1) and x["trg_input"][:, 0].eq(51).all()
        assert torch.equal(x["trg_length"], torch.from_numpy((b["transcript"] != 52).sum(1) + 1))
    assert n_utts > 160 * 0.5                                                         # runs over epochs, drops little

    # evaluation: ordered, fixed utterance count, last batch partial, padded to the longest
    ev = list(task.create_and_batch(ds, compat.ModeKeys.EVAL, args={"batch_size": 64}))
    assert [b["audio"].shape[0] for b in ev] == [64, 64, 32]
    assert ev[0]["audio"].shape[1] == int(ev[0]["audio_length"].max()) * 8


def test_truncation_specaug_and_fixed_transcript_buckets(tmp_path):
    d = str(tmp_path)
    _write_speech_shards(d, n_files=2, per_file=60, seed=5)
    task, ds = _task_and_dataset(d, max_src_len=200, truncate_src=True, truncate_trg=True, max_trg_len=10, specaug="LB",
                                 experimental_frame_transcript_ratio=12, batch_size=1200, min_src_bucket_boundary=64)
    np.random.seed(0)
    it = task.create_and_batch(ds, compat.ModeKeys.TRAIN, seed=1)
    plan = batching.speech_bucket_plan(200, 10, 64, 1200, None, 1, False, 12)
    for _ in range(6):
        b = next(it)
        bucket = plan["audio_bounds"].index(b["audio"].shape[1] // 8)
        assert b["transcript"].shape[1] in plan["trans_bounds"][bucket]               # fixed transcript length of the bucket
        assert (b["audio_length"] <= 200).all()                                      # truncated, not dropped
        real = (b["transcript"] != 52).sum(1) + 1
        assert (real <= 10).all()                                                    # head + EOS kept
    task2, _ = _task_and_dataset(d, specaug="{time_wrap_w: 0, freq_mask_n: 1, freq_mask_f: 4, time_mask_n: 1, time_mask_t: 10, time_mask_p: 1.0}")
    prep = task2.get_data_preprocess_fn(compat.ModeKeys.TRAIN, ds.status)
    ex = next(ds.build_iterator()())
    np.random.seed(4)
    out = prep(dict(ex))
    changed = out["audio"].reshape(-1, 8) != ex["audio"].reshape(-1, 8)
    assert changed.any() and np.unique(out["audio"].reshape(-1, 8)[changed]).size == 1
    evalp = task2.get_data_preprocess_fn(compat.ModeKeys.EVAL, ds.status)
    assert np.array_equal(evalp(dict(ex))["audio"], ex["audio"])                   # no augmentation outside training


def test_raw_transcripts_need_a_pipeline(tmp_path):
    d = str(tmp_path)
    _write_speech_shards(d, n_files=1, per_file=5, projected=False)
    task, ds = _task_and_dataset(d)
    assert ds.status["transcript"] == compat.DataStatus.RAW
    with pytest.raises(RuntimeError):
        next(task.create_and_batch(ds, compat.ModeKeys.TRAIN))
    vocab = [f"w{i}" for i in range(50)]
    task, ds = _task_and_dataset(d, **{"transcript_data_pipeline.params": {"vocab_path": vocab}, "batch_size": 400,
                                       "max_src_len": 399, "min_src_bucket_boundary": 512})
    assert task.trg_meta["vocab_size"] == 53 and task.trg_meta["eos_id"] == 52
    prep = task.get_data_preprocess_fn(compat.ModeKeys.TRAIN, ds.status)
    ex = next(ds.build_iterator()())
    ids = prep(dict(ex))["transcript"]
    assert ids[-1] == 52 and [f"w{i}" for i in ids[:-1]] == ex["transcript"].split()
    assert ds.targets[0] == ex["transcript"]


def test_reference_recipe_yaml_layout_builds_task_and_dataset(tmp_path):
    """The argument layout of examples/speech_transformer/must-c/st_training_args.yml goes through the CLI flag parser:
    entry / dataset / task classes by their reference names, nested *.params, the Noam schedule with a decaying factor."""
    import yaml
    import neurst_amd.cli.run_exp as run_exp
    import neurst_amd.utils.flags_core as fc
    from neurst_amd.optimizers import build_lr_schedule
    train = tmp_path / "asr_st" / "de" / "train"
    train.mkdir(parents=True)
    tfrecord.write_records(str(train / "train.tfrecords-00000-of-00001"),
                           [tfrecord.encode_example({"audio": np.zeros(80 * 50, np.float32), "translation": np.array([1, 2, 5], np.int64)})])
    (tmp_path / "vocab.de").write_text("a\nb\nc\n")
    cfg = {
        "entry.class": "trainer",
        "entry.params": {"train_steps": 200000, "summary_steps": 200, "save_checkpoint_steps": 2000,
                         "criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1},
                         "optimizer.class": "adam", "optimizer.params": {"epsilon": 1.e-9, "beta_1": 0.9, "beta_2": 0.98},
                         "lr_schedule.class": "noam",
                         "lr_schedule.params": {"initial_factor": 3.5, "end_factor": 1.5, "dmodel": 256, "warmup_steps": 25000,
                                                "start_decay_at": 50000, "decay_steps": 50000}},
        "dataset.class": "AudioTFRecordDataset",
        "dataset.params": {"data_path": str(train), "shuffle_dataset": True, "feature_key": "audio", "transcript_key": "translation"},
        "task.class": "SpeechToText",
        "task.params": {"audio_feature_dim": 80, "transcript_data_pipeline.class": "TranscriptDataPipeline",
                        "transcript_data_pipeline.params": {"language": "de", "vocab_path": str(tmp_path / "vocab.de")},
                        "batch_by_frames": True, "batch_size": 80000, "max_src_len": 3000, "max_trg_len": 150, "truncate_src": True,
                        "experimental_frame_transcript_ratio": 12},
    }
    path = tmp_path / "st_training_args.yml"
    path.write_text(yaml.safe_dump(cfg))
    argv = ["--config_paths", str(path), "--hparams_set", "speech_transformer_s"]
    parser = fc.define_flags(run_exp.FLAG_LIST, argv=argv)
    args, _ = fc.intelligent_parse_flags(run_exp.FLAG_LIST, parser, run_exp._pre_load_args, argv=argv)
    assert (args["entry.class"], args["dataset.class"], args["task.class"]) == ("trainer", "AudioTFRecordDataset", "SpeechToText")
    task, ds = build_task(args), build_dataset(args)
    assert task.trg_meta["vocab_size"] == 6 and task.trg_meta["eos_id"] == 5
    assert ds.status == {"audio": compat.DataStatus.PROJECTED, "transcript": compat.DataStatus.PROJECTED}
    assert task.get_config()["transcript_data_pipeline.params"]["language"] == "de"
    lr = build_lr_schedule({"lr_schedule.class": args["entry.params"]["lr_schedule.class"],
                            "lr_schedule.params": args["entry.params"]["lr_schedule.params"]})
    assert abs(lr(24999) - 3.5 * 256 ** -0.5 * 25000 ** -0.5) < 1e-12
    assert abs(lr(99999) - 1.5 * 256 ** -0.5 * 100000 ** -0.5) < 1e-12


# ------------------------------------------------------------------------------------------------ text side (Seq2Seq)
def test_text_bucket_boundaries_match_reference():
    """create_batch_bucket_boundaries / associated_bucket_boundaries: outputs of the reference functions (dataset_utils.py:125-178)."""
    g = np.load(os.path.join(GOLD, "bucket_boundaries.npz"))
    for i in range(5):
        v = g[f"text{i}"].tolist()
        ms, mt, n = v[0], v[1], v[2]
        assert batching.create_batch_bucket_boundaries(ms) == g[f"textraw{i}"].tolist()
        a, b = batching.associated_bucket_boundaries(batching.create_batch_bucket_boundaries(ms), batching.create_batch_bucket_boundaries(mt))
        assert list(a) == v[3:3 + n] and list(b) == v[3 + n:3 + 2 * n]
    plan = batching.text_bucket_plan(80, 80, 4096, None, True, 1)
    assert plan["src_bounds"][0] == 8 and plan["src_bounds"][-1] == 81
    assert plan["batch_sizes"][0] == 4096 // 8 and plan["batch_sizes"][-1] == 4096 // 81
    assert batching.text_bucket_plan(80, 80, None, 32, False, 4)["batch_sizes"][0] == 128
    with pytest.raises(RuntimeError):
        batching.text_bucket_plan(None, 80, 4096, None)


def test_seq2seq_feed_from_tensorflow_written_records(tmp_path):
    """The Translation task trained from `parallel_tfrecord` shards: here the shard is the fixture TensorFlow wrote (the head
    of the reference's tests/examples/train.tfrecords-00000-of-00004), so ids, EOS and lengths are the reference's own."""
    import shutil
    gold = np.load(os.path.join(GOLD, "tfrecord_seq2seq_head.npz"))
    d = tmp_path / "data"
    d.mkdir()
    shutil.copy(os.path.join(GOLD, "tfrecord_seq2seq_head.bin"), d / "train.tfrecords-00000-of-00001")
    vs, vt = int(gold["vocab_size_src"]) + 3, int(gold["vocab_size_trg"]) + 3
    task = build_task({"task.class": "Translation", "task.params": {"src_vocab_size": vs, "trg_vocab_size": vt, "max_src_len": 60,
                                                                    "max_trg_len": 60, "batch_size": 1, "batch_by_tokens": False}})
    ds = build_dataset({"dataset.class": "ParallelTFRecordDataset", "dataset.params": {"data_path": str(d)}})
    assert ds.status == compat.DataStatus.PROJECTED and ds.batched is False
    exs = list(ds.build_iterator()())
    assert len(exs) == 12 and all(e["label"][-1] == vt - 1 and e["feature"][-1] == vs - 1 for e in exs)   # EOS = last id
    it = task.create_and_batch(ds, compat.ModeKeys.TRAIN, seed=0)
    plan = batching.text_bucket_plan(60, 60, 1, None, False)
    for e in exs:                                          # window size 1: every example comes out at once, in order
        b = next(it)
        k = plan["src_bounds"].index(b["feature"].shape[1])
        assert b["label"].shape == (1, plan["trg_bounds"][k]) and len(e["feature"]) <= plan["src_bounds"][k] and len(e["label"]) <= plan["trg_bounds"][k]
        assert k == 0 or len(e["feature"]) > plan["src_bounds"][k - 1] or len(e["label"]) > plan["trg_bounds"][k - 1]   # smallest bucket that fits
        assert b["feature"][0, :len(e["feature"])].tolist() == e["feature"].tolist() and (b["feature"][0, len(e["feature"]):] == vs - 1).all()
        x = task.example_to_input({n: torch.from_numpy(v) for n, v in b.items()}, compat.ModeKeys.TRAIN)
        assert x["trg_length"].tolist() == [len(e["label"])] and x["src_length"].tolist() == [len(e["feature"])]
        assert x["trg_input"][0, 0] == vt - 2 and x["trg_input"][0, 1:len(e["label"])].tolist() == e["label"][:-1].tolist()
    assert np.array_equal(next(it)["feature"][0, :len(exs[0]["feature"])], exs[0]["feature"])               # second epoch
    # token-sized windows on a larger synthetic shard
    rng = np.random.RandomState(1)
    recs = []
    for _ in range(400):
        ls, lt = int(rng.randint(3, 50)), int(rng.randint(3, 50))
        recs.append(tfrecord.encode_example({"feature": np.r_[rng.randint(0, 50, ls - 1), vs - 1], "label": np.r_[rng.randint(0, 50, lt - 1), vt - 1]}))
    d2 = tmp_path / "big"
    d2.mkdir()
    tfrecord.write_records(str(d2 / "train.tfrecords-00000-of-00001"), recs)
    ds2 = build_dataset({"dataset.class": "parallel_tfrecord", "dataset.params": {"data_path": str(d2)}})
    plan = batching.text_bucket_plan(40, 40, 600, None)
    it2 = task.create_and_batch(ds2, compat.ModeKeys.TRAIN, args={"batch_size": 600, "batch_by_tokens": True, "max_src_len": 40,
                                                                 "max_trg_len": 40, "shuffle_buffer": 32}, seed=2)
    for _ in range(10):
        b = next(it2)
        k = plan["src_bounds"].index(b["feature"].shape[1])
        assert b["label"].shape[1] == plan["trg_bounds"][k] and b["feature"].shape[0] == plan["batch_sizes"][k]
        assert b["feature"].shape[0] * max(b["feature"].shape[1], b["label"].shape[1]) <= 600
        real_s = (b["feature"] != vs - 1).sum(1) + 1
        assert (real_s <= 40).all()                                                                      # the length filter ran
    ev = list(task.create_and_batch(ds, compat.ModeKeys.EVAL, args={"batch_size": 5}))
    assert [b["feature"].shape[0] for b in ev] == [5, 5, 2]
    assert ev[0]["feature"].shape[1] == max(len(e["feature"]) for e in exs[:5])


def test_parallel_text_dataset_with_vocabulary_pipelines(tmp_path):
    gold = np.load(os.path.join(GOLD, "tfrecord_seq2seq_head.npz"))
    (tmp_path / "src.txt").write_text("\n".join(str(x) for x in gold["src_lines"]) + "\n", encoding="utf-8")
    (tmp_path / "trg.txt").write_text("\n".join("  " + str(x) + " " for x in gold["trg_lines"]) + "\n", encoding="utf-8")

    def vocab(side):
        n = int(gold[f"vocab_size_{side}"])
        toks = [f"<tok{i}>" for i in range(n)]
        for i, t in zip(gold[f"{side}_ids"], gold[f"{side}_tokens"]):
            toks[int(i)] = str(t)
        return toks
    task = build_task({"task.class": "Seq2Seq", "task.params": {
        "src_data_pipeline.params": {"vocab_path": vocab("src")}, "trg_data_pipeline.params": {"vocab_path": vocab("trg")},
        "max_src_len": 60, "max_trg_len": 60, "batch_size": 400, "truncate_trg": True}})
    assert task.src_meta["vocab_size"] == int(gold["vocab_size_src"]) + 3
    ds = build_dataset({"dataset.class": "ParallelTextDataset",
                        "dataset.params": {"src_file": str(tmp_path / "src.txt"), "trg_file": str(tmp_path / "trg.txt"), "data_is_processed": True}})
    assert ds.status == compat.DataStatus.PROCESSED
    prep = task.get_data_preprocess_fn(compat.ModeKeys.TRAIN, ds.status)
    got = [prep(e) for e in ds.build_iterator()()]
    # the same ids TensorFlow stored for these very sentences
    recs = [tfrecord.parse_example(r) for r in tfrecord.read_records(os.path.join(GOLD, "tfrecord_seq2seq_head.bin"))]
    for g_, r in zip(got, recs):
        assert g_["feature"].tolist() == r["feature"][1].tolist() and g_["label"].tolist() == r["label"][1].tolist()
    # contiguous range sharding by the reference's 1-based counter (parallel_text_dataset.py:121-149)
    parts = [[e["feature"] for e in ds.build_iterator(shard_id=s, total_shards=3)()] for s in range(3)]
    assert [len(p) for p in parts] == [3, 4, 5] and sum(parts, []) == [" ".join(str(x).split()) for x in gold["src_lines"]]
    # truncation keeps the head and the EOS
    short = task.get_data_preprocess_fn(compat.ModeKeys.TRAIN, ds.status, {"max_trg_len": 4})(next(ds.build_iterator()()))
    assert len(short["label"]) == 4 and short["label"][-1] == task.trg_meta["eos_id"] and short["label"][:3].tolist() == recs[0]["label"][1][:3].tolist()
    assert ds.targets[0] == str(gold["trg_lines"][0])


def test_prefetcher_order_errors_and_end():
    import time
    from neurst_amd.data.prefetch import Prefetcher

    def slow():
        for i in range(20):
            time.sleep(0.001)
            yield i
    assert list(Prefetcher(slow(), depth=3)) == list(range(20))

    def boom():
        yield 1
        yield 2
        raise ValueError("bad record")
    it = Prefetcher(boom())
    assert next(it) == 1 and next(it) == 2
    with pytest.raises(ValueError, match="bad record"):
        next(it)
    with pytest.raises(StopIteration):
        next(it)
    # an endless producer is stopped by close(), it does not block on the full queue forever
    def forever():
        i = 0
        while True:
            yield i
            i += 1
    it = Prefetcher(forever(), depth=2)
    assert [next(it) for _ in range(5)] == [0, 1, 2, 3, 4]
    it.close()
    it._thread.join(timeout=2.0)
    assert not it._thread.is_alive()


def test_create_tfrecords_reproduces_tensorflow_bytes(tmp_path):
    """neurst_amd.cli.create_tfrecords on the fixture's sentences (parallel text + the two vocabularies, the layout of the
    reference's tests/examples/example_create_seq2seq_tfrecrods.yml) writes records byte-identical to the ones TensorFlow
    wrote for these sentences."""
    import yaml
    import neurst_amd.cli.create_tfrecords as ct
    gold = np.load(os.path.join(GOLD, "tfrecord_seq2seq_head.npz"))
    (tmp_path / "src.txt").write_text("\n".join(str(x) for x in gold["src_lines"]) + "\n", encoding="utf-8")
    (tmp_path / "trg.txt").write_text("\n".join(str(x) for x in gold["trg_lines"]) + "\n", encoding="utf-8")

    def vocab(side):
        n = int(gold[f"vocab_size_{side}"])
        toks = [f"<tok{i}>" for i in range(n)]
        for i, t in zip(gold[f"{side}_ids"], gold[f"{side}_tokens"]):
            toks[int(i)] = str(t)
        (tmp_path / f"vocab.{side}").write_text("\n".join(toks) + "\n", encoding="utf-8")
        return str(tmp_path / f"vocab.{side}")
    cfg = {"dataset.class": "ParallelTextDataset",
           "dataset.params": {"src_file": str(tmp_path / "src.txt"), "trg_file": str(tmp_path / "trg.txt"), "data_is_processed": True},
           "task.class": "Seq2Seq",
           "task.params": {"src_data_pipeline.class": "TextDataPipeline", "src_data_pipeline.params": {"vocab_path": vocab("src")},
                           "trg_data_pipeline.class": "TextDataPipeline", "trg_data_pipeline.params": {"vocab_path": vocab("trg")}},
           "processor_id": 0, "num_processors": 1, "num_output_shards": 1, "output_range_begin": 0, "output_range_end": 1,
           "output_template": str(tmp_path / "out" / "train.tfrecords-%5.5d-of-%5.5d")}
    (tmp_path / "create.yml").write_text(yaml.safe_dump(cfg))
    paths = ct._main(["--config_paths", str(tmp_path / "create.yml")])
    assert [os.path.basename(p) for p in paths] == ["train.tfrecords-00000-of-00001"] and not os.path.exists(paths[0] + ".incomplete")
    assert open(paths[0], "rb").read() == open(os.path.join(GOLD, "tfrecord_seq2seq_head.bin"), "rb").read()
    # several shards and processors: every example lands in exactly one shard of the processor's range
    cfg.update({"num_output_shards": 4, "output_range_begin": 1, "output_range_end": 3, "seed": 5})
    (tmp_path / "create.yml").write_text(yaml.safe_dump(cfg))
    paths = ct._main(["--config_paths", str(tmp_path / "create.yml")])
    assert [os.path.basename(p) for p in paths] == ["train.tfrecords-00001-of-00004", "train.tfrecords-00002-of-00004"]
    got = sorted(r for p in paths for r in tfrecord.read_records(p))
    assert got == sorted(tfrecord.read_records(os.path.join(GOLD, "tfrecord_seq2seq_head.bin")))
    # audio examples: float features + ids, as the speech recipes store them
    ex = {"audio": np.arange(12, dtype=np.float32) / 7, "transcript": np.array([3, 1, 2]), "uuid": "utt-1"}
    rec = tfrecord.encode_example({k: ct._feature_value(v) for k, v in ex.items()})
    back = tfrecord.parse_example(rec)
    assert back["audio"][0] == "float" and np.array_equal(back["audio"][1], ex["audio"]) and back["uuid"] == ("bytes", [b"utt-1"])


def test_bpe_reproduces_reference_lines(tmp_path):
    """neurst_amd/data/bpe.py on the reference's own merge table (tests/examples/codes.bpe4k.en): de-BPE + BPE of its training
    lines is the identity (all 15190 lines of both languages are checked when the fixture is generated)."""
    from neurst_amd.data.bpe import BPE
    g = np.load(os.path.join(GOLD, "bpe_en.npz"))
    codes = g["codes"].tobytes().decode("utf-8").split("\n")
    lines = g["lines"].tobytes().decode("utf-8").split("\n")
    bpe = BPE(codes)
    assert bpe.version == (0, 2) and len(bpe.ranks) == 4000
    for line in lines:
        raw = bpe.detokenize(line)
        assert "@@" not in raw and bpe.tokenize(raw, return_str=True) == line
    assert bpe.tokenize("maxine", return_str=True) == "max@@ ine"           # train.example.en.tok.bpe.txt line 2
    # through the data pipeline: raw tokenised text -> BPE -> ids -> text again
    (tmp_path / "codes").write_text("\n".join(codes), encoding="utf-8")
    vocab = sorted(set(t for l in lines for t in l.split()))
    dp = TextDataPipeline(vocab_path=vocab, subtokenizer="bpe", subtokenizer_codes=str(tmp_path / "codes"))
    line = lines[0]
    ids = dp.encode(bpe.detokenize(line), is_processed=False)
    assert ids[-1] == dp.meta["eos_id"] and dp.meta["unk_id"] not in ids
    assert dp.decode(ids) == bpe.detokenize(line)
    with pytest.raises(NotImplementedError):
        TextDataPipeline(vocab_path=vocab, tokenizer="moses").encode("a b")
    with pytest.raises(NotImplementedError):
        TextDataPipeline(vocab_path=vocab, subtokenizer="wordpiece").encode("a b")


def test_sentencepiece_subtokenizer(tmp_path):
    """`spm` sub-tokenizer: the pieces are SentencePiece's own (same library as the reference's wrapper, spm.py:28-91)."""
    import sentencepiece as spm
    g = np.load(os.path.join(GOLD, "bpe_en.npz"))
    from neurst_amd.data.bpe import BPE
    raw = [BPE(g["codes"].tobytes().decode("utf-8").split("\n")).detokenize(l) for l in g["lines"].tobytes().decode("utf-8").split("\n")]
    (tmp_path / "corpus.txt").write_text("\n".join(raw * 4) + "\n", encoding="utf-8")
    spm.SentencePieceTrainer.Train(f"--input={tmp_path / 'corpus.txt'} --model_prefix={tmp_path / 'm'} --vocab_size=120 "
                                   "--model_type=bpe --hard_vocab_limit=false --minloglevel=2")
    sp = spm.SentencePieceProcessor()
    sp.Load(str(tmp_path / "m.model"))
    vocab = [sp.IdToPiece(i) for i in range(sp.GetPieceSize())]
    dp = TextDataPipeline(vocab_path=vocab, subtokenizer="spm", subtokenizer_codes=str(tmp_path / "m.model"))
    ids = dp.encode(raw[0], is_processed=False)
    assert [vocab[i] for i in ids[:-1]] == sp.EncodeAsPieces(raw[0]) and ids[-1] == dp.meta["eos_id"]
    assert dp.decode(ids) == raw[0]
