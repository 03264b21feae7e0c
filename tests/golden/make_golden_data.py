#!/usr/bin/env python3
"""Golden vectors for the data-feed row, generated from the reference itself in the build container:

  specaug_*.npz            outputs of the UNMODIFIED reference class neurst/utils/audio_lib.py::SpecAugment (numpy path)
                           under fixed numpy seeds (imported with the tensorflow/absl shim of make_golden.py);
  bucket_boundaries.npz    neurst/tasks/speech2text.py::create_audio_bucket_boundaries and
                           neurst/training/training_utils.py::minimal_multiple (function bodies exec'ed from the files);
  tfrecord_seq2seq_head.bin + .npz   the first records of tests/examples/train.tfrecords-00000-of-00004 -- bytes that
                           TensorFlow wrote -- with the matching lines of the parallel text and the two vocabularies.

usage: python tests/golden/make_golden_data.py   (needs /root/reference; the test suite only reads the outputs)
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
from pb_example import example_message_class  # noqa: E402

REF = mg.REF


def _exec_function(rel, name, extra_globals=None):
    src = open(os.path.join(REF, rel)).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    g = {"math": __import__("math")}
    g.update(extra_globals or {})
    exec(compile(mod, rel, "exec"), g)
    return g[name]


def _save_specaug(tag, x_seed, shape, seed, x, y):
    """y differs from x only in the masked cells, which all hold one value: store the packed mask and that value
    (x is regenerated in the test from RandomState(x_seed).randn(*shape).astype(float32))."""
    changed = (y != x)
    vals = np.unique(y[changed]) if changed.any() else np.zeros(1, np.float32)
    assert vals.size == 1
    assert np.array_equal(np.where(changed, vals[0], x), y)
    mg.save("specaug_" + tag, x_seed=np.int64(x_seed), shape=np.asarray(shape, np.int64), seed=np.int64(seed),
            mask=np.packbits(changed), mask_value=np.float32(vals[0]), n_masked=np.int64(changed.sum()))


def gen_specaug():
    mg._install_shim()
    sys.modules["neurst.utils.compat"].is_tf_tensor = lambda x: False
    al = mg._load("neurst.utils.audio_lib")
    cases = {"LB": (0, (320, 80)), "LD": (1, (451, 80)), "SM": (2, (900, 80)), "SS": (3, (123, 40)), "LB_short": (4, (60, 20))}
    for tag, (seed, shape) in cases.items():
        aug = al.SpecAugment.build(tag.split("_")[0])
        x = np.random.RandomState(100 + seed).randn(*shape).astype(np.float32)
        np.random.seed(seed)
        y = aug(x.copy())
        _save_specaug(tag, 100 + seed, shape, seed, x, np.asarray(y))
    aug = al.SpecAugment.build("{time_wrap_w: 0, freq_mask_n: 3, freq_mask_f: 10, time_mask_n: 4, time_mask_t: 30, time_mask_p: 0.5, mask_value: 0.25}")
    x = np.random.RandomState(7).randn(200, 64).astype(np.float32)
    np.random.seed(11)
    _save_specaug("custom", 7, (200, 64), 11, x, np.asarray(aug(x.copy())))


def gen_buckets():
    f = _exec_function("neurst/tasks/speech2text.py", "create_audio_bucket_boundaries")
    mm = _exec_function("neurst/training/training_utils.py", "minimal_multiple")
    cases = [(3000, 128), (900, 128), (1200, 100), (600, 64), (5000, None), (130, 128), (2048, 256)]
    out = {}
    for i, (mx, mn) in enumerate(cases):
        out[f"case{i}"] = np.asarray([mx, -1 if mn is None else mn] + list(f(mx, mn)), dtype=np.int64)
    vals = [(v, k) for v in (1, 7, 8, 9, 75, 3001, 4096) for k in (8, 3)]
    out["minimal_multiple"] = np.asarray([[v, k, mm(v, k)] for v, k in vals], dtype=np.int64)
    cb = _exec_function("neurst/data/dataset_utils.py", "create_batch_bucket_boundaries",
                        {"_MIN_BUCKET_BOUNDARY": 8, "_BUCKET_BOUNDARY_SCALE": 1.1})
    ab = _exec_function("neurst/data/dataset_utils.py", "associated_bucket_boundaries")
    for i, (ms, mt) in enumerate([(80, 80), (128, 64), (50, 200), (256, 256), (9, 300)]):
        a, b = ab(cb(ms), cb(mt))
        out[f"text{i}"] = np.asarray([ms, mt, len(a)] + list(a) + list(b), dtype=np.int64)
        out[f"textraw{i}"] = np.asarray(cb(ms), dtype=np.int64)
    mg.save("bucket_boundaries", **out)


def gen_tfrecord_head(nrec=12):
    """First `nrec` records of a shard TensorFlow wrote, the text lines they were made from (located by their content:
    create_tfrecords scatters the lines over the shards) and the vocabulary entries those lines use."""
    import struct
    ex = os.path.join(REF, "tests", "examples")
    raw = open(os.path.join(ex, "train.tfrecords-00000-of-00004"), "rb").read()
    off, n = 0, 0
    while n < nrec:
        (ln,) = struct.unpack("<Q", raw[off:off + 8])
        off += 12 + ln + 4
        n += 1
    with open(os.path.join(HERE, "tfrecord_seq2seq_head.bin"), "wb") as fp:
        fp.write(raw[:off])
    src = open(os.path.join(ex, "train.example.zh.jieba.bpe.txt"), encoding="utf-8").read().split("\n")
    trg = open(os.path.join(ex, "train.example.en.tok.bpe.txt"), encoding="utf-8").read().split("\n")
    vocabs = {}
    for lang in ("zh", "en"):
        toks = [l.strip().split()[0] for l in open(os.path.join(ex, "vocab." + lang), encoding="utf-8").read().split("\n") if l.strip()]
        vocabs[lang] = toks
    # locate the lines: google.protobuf (dynamic descriptor of the public tf.train.Example schema) decodes the labels
    Example = example_message_class()
    src_lines, trg_lines = [], []
    off = 0
    for _ in range(nrec):
        (ln,) = struct.unpack("<Q", raw[off:off + 8])
        msg = Example()
        msg.ParseFromString(raw[off + 12:off + 12 + ln])
        off += 12 + ln + 4
        ids = list(msg.features.feature["label"].int64_list.value)
        text = " ".join(vocabs["en"][i] for i in ids[:-1])
        k = trg.index(text)
        trg_lines.append(trg[k])
        src_lines.append(src[k])
    used = {}
    for lang, lines in (("zh", src_lines), ("en", trg_lines)):
        toks = sorted(set(t for l in lines for t in l.split()))
        used[lang] = (np.asarray([vocabs[lang].index(t) for t in toks], np.int64), np.asarray(toks))
    mg.save("tfrecord_seq2seq_head", nrec=np.int64(nrec), src_lines=np.asarray(src_lines), trg_lines=np.asarray(trg_lines),
            vocab_size_src=np.int64(len(vocabs["zh"])), vocab_size_trg=np.int64(len(vocabs["en"])),
            src_ids=used["zh"][0], src_tokens=used["zh"][1], trg_ids=used["en"][0], trg_tokens=used["en"][1])


def gen_bpe(nlines=60):
    """The reference's BPE data: codes.bpe4k.en and lines of train.example.en.tok.bpe.txt.  Checks, here, that re-applying the
    codes to EVERY de-BPE'd line of both languages reproduces the file; commits the English merge table and a sample."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from neurst_amd.data.bpe import BPE
    ex = os.path.join(REF, "tests", "examples")
    for lang, name in (("en", "train.example.en.tok.bpe.txt"), ("zh", "train.example.zh.jieba.bpe.txt")):
        b = BPE(os.path.join(ex, "codes.bpe4k." + lang))
        lines = open(os.path.join(ex, name), encoding="utf-8").read().split("\n")
        bad = [l for l in lines if b.tokenize(b.detokenize(l), return_str=True) != " ".join(l.split())]
        assert not bad, (lang, bad[:3])
    codes = open(os.path.join(ex, "codes.bpe4k.en"), encoding="utf-8").read().split("\n")
    lines = open(os.path.join(ex, "train.example.en.tok.bpe.txt"), encoding="utf-8").read().split("\n")
    sample = [l for l in lines if "@@" in l][:nlines]
    mg.save("bpe_en", codes=np.frombuffer("\n".join(codes).encode("utf-8"), dtype=np.uint8),
            lines=np.frombuffer("\n".join(sample).encode("utf-8"), dtype=np.uint8))


if __name__ == "__main__":
    gen_bpe()
    gen_specaug()
    gen_buckets()
    gen_tfrecord_head()
    print("ok")
