"""Checkpoint row: TensorFlow tensor-bundle reader / writer and the name-based checkpoint manager.
PARITY UNPINNED against TensorFlow-written bytes (none exist in the reference tree): round trips, the format's own
invariants and the reference's naming rules.  CPU only."""
import os
import struct

import numpy as np
import pytest
import torch

from neurst_amd.data.tfrecord import crc32c
from neurst_amd.utils import checkpoints as ck
from neurst_amd.utils import tensor_bundle as tb


def _tensors(rng, n=600):
    out = {}
    for i in range(n):
        shape = tuple(int(x) for x in rng.randint(1, 9, size=rng.randint(0, 4)))
        name = f"Model/layer_{i // 4}/block.{i % 4}/kernel"
        out[tb.checkpoint_key(name)] = np.asarray(rng.randn(*shape), dtype=np.float32)
    out[tb.checkpoint_key("Model/step")] = np.asarray(12345678901, dtype=np.int64)
    out[tb.checkpoint_key("Model/half")] = rng.randn(3, 5).astype(np.float16)
    out[tb.checkpoint_key("Model/big")] = rng.randn(300, 40).astype(np.float32)   # > one 4 KB index block of names follows
    return out


def test_bundle_round_trip_and_format_invariants(tmp_path):
    rng = np.random.RandomState(0)
    tensors = _tensors(rng)
    tensors[tb.OBJECT_GRAPH_KEY] = [b"graph-bytes"]
    prefix = str(tmp_path / "ckpt-7")
    tb.write_bundle(prefix, tensors)
    assert sorted(os.listdir(tmp_path)) == ["ckpt-7.data-00000-of-00001", "ckpt-7.index"]
    back = tb.read_bundle(prefix)
    assert sorted(back) == sorted(tensors)
    for k, v in tensors.items():
        if k == tb.OBJECT_GRAPH_KEY:
            assert back[k] == [b"graph-bytes"]
        else:
            assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57 and len(raw) > 4096      # several data blocks
    # keys are stored in ascending byte order, the header entry (empty key) first
    entries = tb._read_table(prefix + ".index")
    keys = [k for k, _ in entries]
    assert keys[0] == b"" and keys == sorted(keys)
    # data file = the tensors back to back in key order; entry CRCs are the masked CRC-32C of their bytes
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    off = 0
    for k, val in entries[1:]:
        e = tb._parse_entry(val)
        assert e["offset"] == off and e["shard_id"] == 0
        if e["dtype"] != tb.DT_STRING:
            assert e["crc32c"] == tb._mask(crc32c(data[off:off + e["size"]]))
        off += e["size"]
    assert off == len(data)


def test_bundle_detects_corruption(tmp_path):
    prefix = str(tmp_path / "c")
    tb.write_bundle(prefix, {"a": np.arange(100, dtype=np.float32), "b": np.ones((3, 3), np.float32)})
    d = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    d[17] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(d))
    with pytest.raises(tb.BundleError):
        tb.read_bundle(prefix)
    assert tb.read_bundle(prefix, verify=False)["b"].shape == (3, 3)
    i = bytearray(open(prefix + ".index", "rb").read())
    i[5] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(i))
    with pytest.raises(tb.BundleError):
        tb.read_bundle(prefix)
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(tb.BundleError):
        tb.read_bundle(prefix)


def test_bfloat16_entries_are_widened(tmp_path):
    """A bf16 variable (dtype 14) written by a mixed-precision run reads back as the float32 with the same upper bits."""
    prefix = str(tmp_path / "b")
    vals = np.asarray([1.0, -2.5, 3.140625, 0.0], dtype=np.float32)
    tb.write_bundle(prefix, {"w": (vals.view(np.uint32) >> 16).astype(np.uint16)})
    # patch the dtype of the entry from uint16 (17) to bfloat16 (14) and rebuild the index
    entries = tb._read_table(prefix + ".index")
    fixed = [(k, v.replace(b"\x08\x11", b"\x08\x0e", 1) if k == b"w" else v) for k, v in entries]
    tb._write_table(prefix + ".index", fixed)
    assert np.array_equal(tb.read_bundle(prefix)["w"], vals)


def test_object_based_names():
    n = "SpeechTransformer/TransformerEncoder/layer_0/ffn_prepost_wrapper/ffn/dense1/kernel"
    k = tb.checkpoint_key(n)
    assert k == "SpeechTransformer.STransformerEncoder.Slayer_0.Sffn_prepost_wrapper.Sffn.Sdense1.Skernel/.ATTRIBUTES/VARIABLE_VALUE"
    assert tb.variable_name(k) == n
    # neurst/utils/compat.py:152-155 wrapper_var_name agrees on names without dots
    assert k.replace(".S", "/").replace("/.ATTRIBUTES/VARIABLE_VALUE", "") == n
    assert tb.variable_name(tb.checkpoint_key("a.b/c.S")) == "a.b/c.S"


class _Param(object):
    def __init__(self, t):
        self.data, self.shape = t, tuple(t.shape)


class _Store(object):
    def __init__(self, shapes, seed):
        g = torch.Generator().manual_seed(seed)
        self.params = {n: _Param(torch.randn(*s, generator=g)) for n, s in shapes.items()}

    def state_dict(self):
        return {n: p.data.clone() for n, p in self.params.items()}

    def load_state_dict(self, sd, strict=True):
        for n, v in sd.items():
            self.params[n].data.copy_(v.reshape(self.params[n].shape))


class SpeechTransformer(object):
    def __init__(self, seed, extra=False):
        shapes = {"input_audio_modality/conv1/kernel": (3, 3, 1, 4), "TransformerEncoder/layer_0/ln/gamma": (4,),
                  "target_symbol_modality/shared/weights": (11, 4)}
        if extra:
            shapes["TransformerDecoder/new_head/kernel"] = (4, 2)
        self.store = _Store(shapes, seed)


class Renamed(SpeechTransformer):
    pass


class _Opt(object):
    def __init__(self):
        self.iterations, self.m, self.v = 0, torch.zeros(7), torch.zeros(7)

    def state(self):
        return {"step": self.iterations, "m": self.m, "v": self.v}

    def load_state(self, st):
        self.iterations, self.m, self.v = int(st["step"]), st["m"].clone(), st["v"].clone()


def test_checkpoint_manager_save_restore_scope_mapping_and_rotation(tmp_path):
    d = str(tmp_path / "model")
    src, opt = SpeechTransformer(1), _Opt()
    opt.iterations, opt.m = 40, torch.arange(7.0)
    mgr = ck.NameBasedCheckpointManager(src, d, max_to_keep=2, optimizer=opt)
    assert ck.restore_checkpoint_if_possible(SpeechTransformer(2), d) is None          # nothing there yet
    for step in (10, 20, 30):
        src.store.params["TransformerEncoder/layer_0/ln/gamma"].data.fill_(float(step))
        mgr.save(step)
    files = sorted(os.listdir(d))
    assert files == ["checkpoint", "ckpt-20.data-00000-of-00001", "ckpt-20.index", "ckpt-30.data-00000-of-00001", "ckpt-30.index"]
    meta = open(os.path.join(d, "checkpoint")).read().splitlines()
    assert meta[0] == 'model_checkpoint_path: "ckpt-30"' and meta[1:3] == ['all_model_checkpoint_paths: "ckpt-20"',
                                                                            'all_model_checkpoint_paths: "ckpt-30"']
    assert meta[3].startswith("all_model_checkpoint_timestamps: ")
    assert ck.latest_checkpoint(d) == os.path.join(d, "ckpt-30")
    names = dict(ck.list_variables(os.path.join(d, "ckpt-30")))
    assert names["SpeechTransformer/input_audio_modality/conv1/kernel"] == [3, 3, 1, 4]
    assert ck.checkpoint_scope_name(os.path.join(d, "ckpt-30")) == "SpeechTransformer"
    graph = tb.read_bundle(os.path.join(d, "ckpt-30"))[tb.OBJECT_GRAPH_KEY][0]
    assert b"VARIABLE_VALUE" in graph and tb.checkpoint_key("SpeechTransformer/target_symbol_modality/shared/weights").encode() in graph

    # a model with another top scope restores by replacing the scope (checkpoints.py:340-361)
    dst, opt2 = Renamed(3, extra=True), _Opt()
    before = dst.store.params["TransformerDecoder/new_head/kernel"].data.clone()
    assert ck.restore_checkpoint_if_possible(dst, d, optimizer=opt2) == os.path.join(d, "ckpt-30")
    for n in src.store.params:
        assert torch.equal(dst.store.params[n].data, src.store.params[n].data)
    assert torch.equal(dst.store.params["TransformerDecoder/new_head/kernel"].data, before)     # absent in the checkpoint: untouched
    assert opt2.iterations == 40 and torch.equal(opt2.m, torch.arange(7.0))
    # explicit prefix + name filter
    part = SpeechTransformer(4)
    keep = part.store.params["input_audio_modality/conv1/kernel"].data.clone()
    assert ck.restore_checkpoint_if_possible(part, os.path.join(d, "ckpt-20"), var_name_pattern="ln|shared") is not None
    assert torch.equal(part.store.params["input_audio_modality/conv1/kernel"].data, keep)
    assert float(part.store.params["TransformerEncoder/layer_0/ln/gamma"].data[0]) == 20.0
    # shape mismatch = not restored
    bad = SpeechTransformer(5)
    bad.store.params["target_symbol_modality/shared/weights"] = _Param(torch.zeros(12, 4))
    ck.restore_checkpoint_if_possible(bad, d)
    assert float(bad.store.params["target_symbol_modality/shared/weights"].data.abs().sum()) == 0.0


def test_avg_checkpoint_cli(tmp_path):
    """avg_checkpoint.py:49-103: running mean over the checkpoints of a directory, `_`-prefixed variables dropped, one
    ckpt + state file + model_configs.yml written; the result restores through the normal manager path."""
    import numpy as np
    from neurst_amd.cli.avg_checkpoint import average_checkpoints
    from neurst_amd.utils import tensor_bundle as tb
    from neurst_amd.utils.checkpoints import latest_checkpoint, list_variables
    d = tmp_path / "run"
    d.mkdir()
    rng = np.random.RandomState(0)
    vals = []
    lines = []
    for step in (10, 20, 30):
        w, b = rng.randn(4, 3).astype(np.float32), rng.randn(3).astype(np.float32)
        vals.append((w, b))
        tb.write_bundle(str(d / f"ckpt-{step}"), {
            tb.checkpoint_key("SpeechTransformer/enc/kernel"): w, tb.checkpoint_key("SpeechTransformer/enc/bias"): b,
            "_optimizer/step": np.asarray(step, dtype=np.int64), "_optimizer/m": np.ones(5, np.float32),
            tb.OBJECT_GRAPH_KEY: [tb.object_graph_proto(["SpeechTransformer/enc/kernel", "SpeechTransformer/enc/bias"])]})
        lines.append(f'all_model_checkpoint_paths: "ckpt-{step}"')
    (d / "checkpoint").write_text('model_checkpoint_path: "ckpt-30"\n' + "\n".join(lines) + "\n")
    (d / "model_configs.yml").write_text("model.class: SpeechTransformer\n")
    out = tmp_path / "avg"
    prefix, paths = average_checkpoints(str(d), str(out))
    assert len(paths) == 3 and latest_checkpoint(str(out)) == prefix
    got = {tb.variable_name(k): v for k, v in tb.read_bundle(prefix).items() if k != tb.OBJECT_GRAPH_KEY}
    assert set(got) == {"SpeechTransformer/enc/kernel", "SpeechTransformer/enc/bias"}
    np.testing.assert_allclose(got["SpeechTransformer/enc/kernel"], np.mean([w for w, _ in vals], 0), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(got["SpeechTransformer/enc/bias"], np.mean([b for _, b in vals], 0), rtol=1e-6, atol=1e-7)
    assert sorted(n for n, _ in list_variables(prefix)) == sorted(got)
    assert (out / "model_configs.yml").read_text().startswith("model.class")
    # explicit prefixes, comma separated
    prefix2, paths2 = average_checkpoints(f"{d}/ckpt-10,{d}/ckpt-30", str(tmp_path / "avg2"))
    got2 = {tb.variable_name(k): v for k, v in tb.read_bundle(prefix2).items() if k != tb.OBJECT_GRAPH_KEY}
    np.testing.assert_allclose(got2["SpeechTransformer/enc/bias"], (vals[0][1] + vals[2][1]) / 2, rtol=1e-6, atol=1e-7)
