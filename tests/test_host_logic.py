"""CPU tests of the host side: C-ABI surface, registry / flag semantics of the reference, variable naming,
parameter store, schedule, task glue.  No kernel is launched (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ C ABI
def _header_symbols():
    src = open(os.path.join(ROOT, "include", "neurst_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nst_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from neurst_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(_lib.lib, s), f"{s} declared in include/neurst_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.lib.nst_abi_version() == _lib.NST_ABI_VERSION == 10
    out = subprocess.check_output(["nm", "-D", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (nst_[a-z0-9_]+)", out))
    assert exported == set(syms)


def test_comm_entry_points_validate_their_arguments_without_a_device():
    """nst_comm_* (include/neurst_hip.h, ABI v8): argument errors are reported before RCCL or the device is touched."""
    from neurst_amd import _lib
    L = _lib.lib
    small = ctypes.create_string_buffer(16)
    assert L.nst_comm_unique_id(small, len(small)) == -1 and b"128" in L.nst_last_error_string()
    handle = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(_lib.NST_COMM_UNIQUE_ID_BYTES)
    assert L.nst_comm_init(ident, len(ident), 2, 2, ctypes.byref(handle)) == -1 and handle.value is None
    assert b"rank 2 of 2" in L.nst_last_error_string()
    assert L.nst_comm_init(ident, 8, 0, 1, ctypes.byref(handle)) == -1
    fake = ctypes.create_string_buffer(256)      # not a communicator: the magic number is missing
    assert L.nst_comm_allreduce_bucket(fake, None, 0, _lib.NST_F32, None, 0) == -1
    assert b"not a communicator" in L.nst_last_error_string()
    assert L.nst_comm_fence(fake, None) == -1 and L.nst_comm_broadcast(fake, None, 0, _lib.NST_F32, 0, None) == -1
    assert L.nst_comm_destroy(fake) == -1 and L.nst_comm_destroy(None) == 0


def test_stream_create_rejects_unknown_priority_classes_without_a_device():
    from neurst_amd import _lib
    handle = ctypes.c_void_p()
    assert _lib.lib.nst_stream_create(2, ctypes.byref(handle)) == -1 and handle.value is None
    assert b"priority class 2" in _lib.lib.nst_last_error_string()
    assert _lib.lib.nst_stream_create(0, None) == -1
    assert _lib.lib.nst_stream_destroy(None) == 0


def test_struct_layouts_match_header_sizes():
    from neurst_amd import _lib
    # field order/types mirror the header; sizes guard against silent drift (x86-64 SysV layout)
    assert ctypes.sizeof(_lib.NstGemmDesc) == 232
    assert ctypes.sizeof(_lib.NstAttnDesc) == 152
    assert ctypes.sizeof(_lib.NstFfnDesc) == 88 and ctypes.sizeof(_lib.NstTransposeJob) == 32 and ctypes.sizeof(_lib.NstSplitkJob) == 64


def test_kernels_refuse_cpu_tensors():
    from neurst_amd import kernels as K
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.layernorm_fwd(torch.zeros(2, 8), torch.ones(8), torch.zeros(8), 1e-6)


# ------------------------------------------------------------------------------------------------ registry / flags
def test_registry_semantics():
    from neurst_amd.utils.registry import setup_registry

    class Base(object):
        pass

    build, register = setup_registry("thing_t", base_class=Base, backend="pt")

    @register
    class MyThing(Base):
        def __init__(self, a=1, b=2):
            self.a, self.b = a, b

    @register(["alias1", "alias2"])
    class OtherThing(Base):
        def __init__(self, **kw):
            self.kw = kw

    for name in ("MyThing", "mything", "my_thing"):
        assert isinstance(build({"thing_t.class": name, "thing_t.params": {"a": 5}}), MyThing)
    assert build({"class": "alias2", "params": {"x": 1}}).kw == {"x": 1}
    assert build("MyThing", b=7).b == 7
    assert build({"thing_t.class": None}) is None and build("none") is None
    with pytest.raises(ValueError, match="Not registered class name"):
        build({"thing_t.class": "nope"})
    with pytest.raises(ValueError, match="must extend"):
        register(type("Bad", (), {}))
    with pytest.raises(ValueError, match="Cannot register duplicate"):
        register("alias1")(type("Third", (Base,), {}))


def test_flag_precedence_cli_over_config_over_hparams(tmp_path):
    import neurst_amd.cli.run_exp as run_exp
    import neurst_amd.utils.flags_core as fc
    cfg = tmp_path / "cfg.yml"
    cfg.write_text("dtype: float32\nmodel.params:\n  encoder.num_layers: 3\n  decoder.num_layers: 4\n"
                   "entry.class: trainer\nentry.params:\n  train_steps: 77\ntask.class: speech2text\n"
                   "dataset.class: synthetic_speech\ndataset.params:\n  frames: 120\n")
    argv = ["--config_paths", str(cfg), "--hparams_set", "speech_transformer_s", "--encoder.num_layers", "5",
            "--summary_steps", "9", "--batch_per_gpu", "16"]
    parser = fc.define_flags(run_exp.FLAG_LIST, argv=argv)
    args, rest = fc.intelligent_parse_flags(run_exp.FLAG_LIST, parser, run_exp._pre_load_args, argv=argv)
    assert args["dtype"] == "float32"                                  # config file beats the flag default
    assert args["model.class"] == "SpeechTransformer"                  # from the hparams set
    assert args["model.params"]["encoder.num_layers"] == 5             # CLI beats config beats hparams (12)
    assert args["model.params"]["decoder.num_layers"] == 4             # config beats hparams (6)
    assert args["model.params"]["encoder.hidden_size"] == 256          # hparams set
    assert args["entry.params"]["train_steps"] == 77
    assert args["entry.params"]["summary_steps"] == 9                  # nested class flag given flat on the CLI
    assert args["entry.params"]["optimizer.class"] == "Adam"           # hparams set's entry params survive
    assert args["entry.params"]["lr_schedule.params"]["warmup_steps"] == 25000
    assert args["dataset.params"]["frames"] == 120 and args["dataset.params"]["batch_per_gpu"] == 16
    assert rest == []


def test_hparams_sets_match_reference_tables():
    from neurst_amd.models import build_model  # noqa: F401 (registers the models)
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    s = get_hyper_parameters("speech_transformer_s")  # neurst/models/speech_transformer.py:209-216
    p = s["model.params"]
    assert (p["modality.dim"], p["encoder.num_attention_heads"], p["encoder.num_layers"], p["decoder.num_layers"],
            p["encoder.filter_size"], p["modality.source.channels"]) == (256, 4, 12, 6, 2048, 256)
    assert s["lr_schedule.params"]["initial_factor"] == 3.5 and s["optimizer.params"]["beta_2"] == 0.98
    m = get_hyper_parameters("speech_transformer_m")["model.params"]
    assert (m["modality.dim"], m["encoder.num_attention_heads"]) == (512, 8)
    big = get_hyper_parameters("transformer_big")["model.params"]
    assert (big["modality.dim"], big["encoder.num_attention_heads"], big["encoder.attention_dropout_rate"]) == (1024, 16, 0.3)
    assert get_hyper_parameters("no_such_set") == {}


# ------------------------------------------------------------------------------------------------ variables
def test_speech_transformer_variable_names_and_counts():
    """TF variable names / layouts the reference's tests address (tests/neurst_pt/models/speech_transformer_test.py:57-152)."""
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    hp = get_hyper_parameters("speech_transformer_s")
    m = build_model(hp, {"audio_feature_dim": 80, "audio_feature_channels": 1},
                    {"vocab_size": 8008, "eos_id": 8007, "bos_id": 8006, "unk_id": 8005}, device="cpu", dtype="bfloat16")
    P = m.store.params
    assert P["input_audio_modality/conv1/kernel"].shape == (3, 3, 1, 256)
    assert P["input_audio_modality/conv2/kernel"].shape == (3, 3, 256, 256)
    assert P["input_audio_modality/output_dense/kernel"].shape == (5120, 256)
    assert P["target_symbol_modality/shared/weights"].shape == (8008, 256)
    assert P["target_symbol_modality/shared/bias"].shape == (8008,)
    e = "TransformerEncoder/layer_11/self_attention_prepost_wrapper/"
    assert P[e + "self_attention/qkv_transform/kernel"].shape == (256, 768)
    assert P[e + "self_attention/output_transform/kernel"].shape == (256, 256)
    assert P[e + "ln/gamma"].shape == (256,)
    assert P["TransformerEncoder/layer_0/ffn_prepost_wrapper/ffn/dense1/kernel"].shape == (256, 2048)
    d = "TransformerDecoder/layer_5/encdec_attention_prepost_wrapper/encdec_attention/"
    assert P[d + "q_transform/kernel"].shape == (256, 256) and P[d + "kv_transform/kernel"].shape == (256, 512)
    assert "TransformerEncoder/output_ln/gamma" in P and "TransformerDecoder/output_ln/beta" in P
    n = m.store.num_parameters()
    assert abs(n - 29.2e6) < 0.1e6, n              # 29.2 M parameters
    assert m.store.shadow.dtype == torch.bfloat16 and m.store.grad.dtype == torch.float32
    for p in P.values():                             # every parameter 16-byte aligned in the bf16 shadow
        assert p.offset % 8 == 0
    assert torch.equal(m.store.shadow[:100].float(), m.store.master[:100].to(torch.bfloat16).float())


def test_param_store_gradient_bookkeeping():
    from neurst_amd.runtime import ParamStore
    st = ParamStore()
    a = st.add("a", (3,), torch.arange(3.0))
    b = st.add("b", (2, 2), torch.ones(2, 2))
    with pytest.raises(ValueError, match="duplicate"):
        st.add("a", (1,), torch.zeros(1))
    st.finalize("cpu", torch.float32)
    assert a.compute.data_ptr() == a.data.data_ptr()   # fp32 mode: no shadow
    a.grad.fill_(3.0)
    st.begin_backward()                      # zeroes the flat gradient once; every kernel then accumulates
    assert float(st.grad.abs().sum()) == 0.0 and st.acc_flag(a) is True and st.acc_flag(b) is True
    a.grad.fill_(3.0)
    st.begin_backward(accumulate=True)       # gradient accumulation micro step: keep what is there
    assert float(a.grad.sum()) == 9.0 and st.acc_flag(a) is True
    sd = st.state_dict()
    sd["b"] = torch.full((2, 2), 5.0)
    st.load_state_dict(sd)
    assert float(b.data.sum()) == 20.0


def test_noam_schedule_matches_oracle():
    from neurst_amd.optimizers import build_lr_schedule
    from oracle import neurst_oracle as O
    kw = dict(dmodel=256, warmup_steps=25000, initial_factor=3.5, end_factor=1.5, start_decay_at=50000, decay_steps=50000)
    s = build_lr_schedule({"lr_schedule.class": "noam", "lr_schedule.params": kw})
    for step in (0, 1, 100, 24999, 25000, 60000, 99999, 150000):
        assert abs(s(step) - O.noam_lr(step, **kw)) < 1e-15
    s2 = build_lr_schedule({"lr_schedule.class": "noam", "lr_schedule.params": {"dmodel": 8, "warmup_steps": 4000}})
    assert abs(s2(0) - O.noam_lr(0, dmodel=8, warmup_steps=4000)) < 1e-15


def test_example_to_input_matches_oracle():
    from neurst_amd.data.datasets.synthetic_speech import SyntheticSpeechDataset
    from neurst_amd.tasks import build_task
    from neurst_amd.utils import compat
    from oracle import neurst_oracle as O
    task = build_task({"task.class": "AudioToText", "task.params": {"audio_feature_dim": 16, "vocab_size": 30}})
    ds = SyntheticSpeechDataset({"batch_per_gpu": 5, "frames": 48, "feature_dim": 16, "vocab_size": 30, "ragged": True})
    raw = ds.make_batch(torch.Generator().manual_seed(0))
    got = task.example_to_input(raw, compat.ModeKeys.TRAIN)
    ref = O.example_to_input(raw["audio"], raw["audio_length"], raw["transcript"], 16, 1, 28, 29)
    for k in ref:
        assert torch.equal(got[k], ref[k]), k
    assert got["src"].shape == (5, 48, 16, 1) and int(got["trg"][0, -1]) == 29
    # every transcript ends with EOS and is padded with EOS; length counts the first EOS
    assert all(int(got["trg"][i, int(got["trg_length"][i]) - 1]) == 29 for i in range(5))
    # different ranks draw different data, same rank reproduces
    a = next(ds.build_iterator(shard_id=0))["audio"]
    b = next(ds.build_iterator(shard_id=1))["audio"]
    assert not torch.equal(a, b) and torch.equal(a, next(ds.build_iterator(shard_id=0))["audio"])


def test_length_masks():
    from neurst_amd.models.model_utils import deduce_text_length, input_length_to_padding
    from neurst_amd.utils.compat import PaddingMode
    pad = input_length_to_padding(torch.tensor([3, 1]), 4)
    assert pad.tolist() == [[0, 0, 0, 1], [0, 1, 1, 1]]
    t = torch.tensor([[5, 6, 9, 9], [9, 9, 9, 9], [1, 2, 3, 9]])
    assert deduce_text_length(t, 9, PaddingMode.EOS_AS_PADDING).tolist() == [3, 1, 4]
    assert deduce_text_length(t, 9, PaddingMode.DEFAULT).tolist() == [2, 0, 3]


def test_graft_entry_build_runs():
    import __graft_entry__ as g
    g.build()


def test_seq2seq_example_to_input_follows_reference():
    """neurst/tasks/seq2seq.py:110-136: src_length / trg_length from EOS-as-padding, trg_input = [BOS, trg[:-1]]."""
    import torch
    from neurst_amd.tasks import build_task
    from neurst_amd.utils import compat
    task = build_task({"task.class": "translation", "task.params": {"src_vocab_size": 13, "trg_vocab_size": 23}})
    eos_s, eos_t, bos_t = 12, 22, 21
    batch = {"feature": torch.tensor([[1, 2, 3, eos_s, eos_s], [4, 5, 6, 7, eos_s]]),
             "label": torch.tensor([[9, 8, eos_t, eos_t], [1, 2, 3, eos_t]])}
    out = task.example_to_input(batch, compat.ModeKeys.TRAIN)
    assert out["src_length"].tolist() == [4, 5] and out["trg_length"].tolist() == [3, 4]
    assert out["trg_input"].tolist() == [[bos_t, 9, 8, eos_t], [bos_t, 1, 2, 3]]
    assert torch.equal(out["trg"], batch["label"]) and torch.equal(out["src"], batch["feature"])
    inf = task.example_to_input(batch, compat.ModeKeys.INFER)
    assert inf["trg_input"].tolist() == [bos_t, bos_t] and "trg" not in inf
    eos_task = build_task({"task.class": "seq2seq", "task.params": {"src_vocab_size": 13, "trg_vocab_size": 23,
                                                                     "target_begin_of_sentence": "eos"}})
    assert eos_task.example_to_input(batch, compat.ModeKeys.TRAIN)["trg_input"][:, 0].tolist() == [eos_t, eos_t]


def test_synthetic_text_dataset_shapes_and_eos_padding():
    import torch
    from neurst_amd.data.datasets import build_dataset
    ds = build_dataset({"dataset.class": "synthetic_text", "dataset.params": {"batch_per_gpu": 4, "src_len": 9, "trg_len": 7,
                                                                            "src_vocab_size": 50, "trg_vocab_size": 60,
                                                                            "ragged": True, "num_batches": 2}})
    batches = list(ds.build_iterator())
    assert len(batches) == 2
    b = batches[0]
    assert b["feature"].shape == (4, 9) and b["label"].shape == (4, 7)
    assert (b["feature"][:, -1] == 49).all() and (b["label"][:, -1] == 59).all()
    assert int(b["feature"].max()) <= 49 and int(b["label"][:, :-1].min()) >= 0


def test_inverse_sqrt_and_piecewise_schedules():
    """inverse_sqrt_schedule.py:57-70, piecewise_schedule.py:66-82 (known answers from the formulas)."""
    from neurst_amd.optimizers import build_lr_schedule
    s = build_lr_schedule({"lr_schedule.class": "inverse_sqrt",
                           "lr_schedule.params": {"peak_lr": 5e-4, "init_lr": 1e-7, "warmup_steps": 4000}})
    assert abs(s(0) - (1e-7 + 1 * (5e-4 - 1e-7) / 4000)) < 1e-15          # global step 0 -> step 1
    assert abs(s(1998) - (1e-7 + 1999 * (5e-4 - 1e-7) / 4000)) < 1e-15
    assert abs(s(3999) - 5e-4) < 1e-12                                      # step 4000: the peak
    assert abs(s(15999) - 5e-4 * (4000 / 16000) ** 0.5) < 1e-12
    p = build_lr_schedule({"lr_schedule.class": "piecewise",
                           "lr_schedule.params": {"schedule_steps": "[100, 200, 400]", "schedule_lrs": [1e-3, 5e-4, 1e-4, 1e-5]}})
    assert abs(p(0) - 1e-3 / 100) < 1e-15 and abs(p(49) - 0.5e-3) < 1e-15
    # the reference's late-binding closures make every middle segment use the LAST middle rate (see the module docstring)
    assert p(99) == 1e-4 and p(198) == 1e-4 and p(199) == 1e-4 and p(398) == 1e-4 and p(399) == 1e-5 and p(10 ** 6) == 1e-5
    assert p.get_config()["schedule_steps"] == [100, 200, 400]
    from neurst_amd.optimizers.schedules.piecewise_schedule import PiecewiseSchedule
    q = PiecewiseSchedule({"schedule_steps": [100, 200, 400], "schedule_lrs": [1e-3, 5e-4, 1e-4, 1e-5]}, strict=True)
    assert q(99) == 5e-4 and q(198) == 5e-4 and q(199) == 1e-4


def test_lr_schedules_match_the_reference_classes():
    """tests/golden/lr_schedules.json: values of the reference's own NoamSchedule / InverseSquareRootSchedule /
    PiecewiseSchedule classes (executed over the torch-backed TensorFlow stand-in of make_golden.py, float32 scalars), from
    step 0 and resumed at step 1234."""
    import json
    from neurst_amd.optimizers import build_lr_schedule
    from neurst_amd.utils import compat
    from oracle import neurst_oracle as O
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "lr_schedules.json")))
    cls = {"noam_st_s": "noam", "noam_plain": "noam", "inverse_sqrt": "inverse_sqrt", "piecewise": "piecewise"}
    try:
        for key, rec in G.items():
            if key == "steps":
                continue
            name, initial = key.split("@")
            compat.register_initial_step(int(initial))
            sched = build_lr_schedule({"lr_schedule.class": cls[name], "lr_schedule.params": dict(rec["args"])})
            for step, want in zip(G["steps"], rec["values"]):
                assert sched(step) == pytest.approx(want, rel=2e-6, abs=1e-12), (key, step)
                if cls[name] == "noam" and int(initial) == 0:
                    kw = {k: v for k, v in rec["args"].items() if v is not None}
                    assert O.noam_lr(step, **kw) == pytest.approx(want, rel=2e-6, abs=1e-12)
    finally:
        compat.register_initial_step(0)


def test_example_to_input_matches_the_reference_methods():
    """tests/golden/example_to_input_reference.npz: the bodies of the reference's SpeechToText.example_to_input
    (tasks/speech2text.py:135-161), Seq2Seq.example_to_input (tasks/seq2seq.py:110-136) and deduce_text_length
    (models/model_utils.py:23-41), compiled from their source and executed over the TensorFlow stand-in of make_golden.py."""
    import numpy as np
    from neurst_amd.models.model_utils import deduce_text_length
    from neurst_amd.tasks import build_task
    from neurst_amd.utils import compat
    from oracle import neurst_oracle as O
    z = np.load(os.path.join(ROOT, "tests", "golden", "example_to_input_reference.npz"))
    Fd, C, V, bos, eos = (int(v) for v in z["dims"])
    st = build_task({"task.class": "speech2text", "task.params": {"audio_feature_dim": Fd, "vocab_size": V}})
    assert (st.trg_meta["bos_id"], st.trg_meta["eos_id"], st.trg_meta["pad_id"]) == (bos, eos, eos)
    batch = {k: torch.from_numpy(z[k]) for k in ("audio", "audio_length", "transcript")}
    for mode, key in ((compat.ModeKeys.TRAIN, "train"), (compat.ModeKeys.INFER, "infer")):
        got = st.example_to_input(dict(batch), mode)
        want = {k.split(":")[1]: z[k] for k in z.files if k.startswith(f"st_{key}:")}
        assert set(want) <= set(got)
        for k, v in want.items():
            assert np.array_equal(got[k].numpy(), v), (key, k)
    ref = O.example_to_input(batch["audio"], batch["audio_length"], batch["transcript"], Fd, C, bos, eos)
    for k in ("src", "src_length", "trg", "trg_length", "trg_input"):
        assert np.array_equal(ref[k].numpy(), z[f"st_train:{k}"]), k
    for tb in ("bos", "eos"):
        s2s = build_task({"task.class": "translation", "task.params": {"src_vocab_size": V, "trg_vocab_size": V,
                                                                        "target_begin_of_sentence": tb}})
        data = {"feature": torch.from_numpy(z["feature"]), "label": torch.from_numpy(z["transcript"])}
        for mode, key in ((compat.ModeKeys.TRAIN, "train"), (compat.ModeKeys.INFER, "infer")):
            got = s2s.example_to_input(dict(data), mode)
            for k in (k for k in z.files if k.startswith(f"s2s_{tb}_{key}:")):
                assert np.array_equal(got[k.split(":")[1]].numpy(), z[k]), k
    assert deduce_text_length(torch.from_numpy(z["default_pad_ids"]), 0, compat.PaddingMode.DEFAULT).tolist() == \
        z["default_pad_lengths"].tolist()


def test_weight_gradient_split_survives_the_library_rounding_and_pins_to_xcds():
    """layers/common_layers._wgrad_split: the library re-derives the slice count as ceil(kt / ceil(kt / split)); gradients
    with >= 8 tiles get a multiple of 8 that survives that rounding (the stream kernel then runs K slice z on XCD z % 8),
    smaller ones keep a finer split; never more than kt / 8 slices."""
    import torch
    from neurst_amd.layers.common_layers import _wgrad_split
    for rows in (9600, 28800, 115200, 1000):
        for k_in, n_out in ((256, 256), (256, 768), (256, 2048), (2048, 256), (5120, 256), (256, 8008), (64, 128)):
            for dtype in (torch.bfloat16, torch.float32):
                split = _wgrad_split(rows, k_in, n_out, dtype)
                bk = 64 if dtype == torch.bfloat16 else 32
                kt = (rows + bk - 1) // bk
                tiles = ((k_in + 127) // 128) * ((n_out + 127) // 128)
                assert 1 <= split <= max(1, kt // 8)
                if tiles >= 8 and split >= 8:
                    assert split % 8 == 0 and -(-kt // -(-kt // split)) == split, (rows, k_in, n_out, split)
    assert _wgrad_split(28800, 256, 2048, torch.bfloat16) == 8          # FFN: 32 tiles x 8 slices = one slice per XCD
    assert _wgrad_split(28800, 256, 768, torch.bfloat16) == 24
    assert _wgrad_split(28800, 5120, 256, torch.bfloat16, units=512) == 6


def test_bench_autotune_picks_by_the_two_percent_rule_and_survives_bad_candidates(monkeypatch, tmp_path):
    """bench.autotune (the N > 1 self-diagnosis VERDICT r4 item 4 asks for), with the child jobs replaced by canned outputs:
    a crashed candidate and one that prints no JSON are recorded and skipped, a candidate must beat the default by more than
    2 % to replace it, factors the caller fixed in the environment are not varied, the other ranks read rank 0's choice."""
    import argparse
    import json
    import subprocess as sp
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench

    for k in ("GPU_MAX_HW_QUEUES", "NCCL_MAX_NCHANNELS", "NST_DIST_NATIVE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_PORT", "45991")
    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    args = argparse.Namespace(autotune_budget=200.0, dtype="bf16", model="speech_transformer_s", batch=8, frames=96, vocab=64,
                              wire="bf16", graph=None, ragged=False, strong=False)
    outputs = {}

    class FakeProc:
        def __init__(self, cmd, env=None, **kw):
            assert kw.get("start_new_session") is True and "WORLD_SIZE" not in env and env["NST_BENCH_CHILD"] == "1"
            label = next(lb for lb, c in bench.AUTOTUNE_CANDIDATES if all(env.get(k) == v for k, v in c.items()))
            self.out, self.returncode, self.pid = outputs[label], 0, 1

        def communicate(self, timeout=None):
            if isinstance(self.out, Exception):
                raise self.out
            return self.out, "trace\nlast line of stderr"

    monkeypatch.setattr(sp, "Popen", FakeProc)
    line = lambda ms: "noise\n" + json.dumps({"ms_per_step": ms, "exchange": {"exposed_ms": 0.1}}) + "\n"

    # 1.5 % faster is noise: the default stays; the crashed / silent candidates are in the report with their reason
    outputs.update(hwq1_ch16_torch=line(10.0), hwq2_ch16_torch=line(9.85), hwq1_ch32_torch="no json here\n",
                   hwq1_ch16_native=OSError("spawn failed"))
    rep = bench.autotune(args, 0, 2)
    assert rep["chosen"] == "hwq1_ch16_torch" and os.environ["GPU_MAX_HW_QUEUES"] == "1"
    by = {c["label"]: c for c in rep["candidates"]}
    assert "no JSON line" in by["hwq1_ch32_torch"]["error"] and "last line of stderr" in by["hwq1_ch32_torch"]["error"]
    assert by["hwq1_ch16_native"]["error"].startswith("OSError") and by["hwq2_ch16_torch"]["ms_per_step"] == 9.85
    # another rank of the same launch reads the published choice
    for k in rep["chosen_env"]:
        monkeypatch.delenv(k, raising=False)
    rep1 = bench.autotune(args, 1, 2)
    assert rep1["chosen"] == rep["chosen"] and os.environ["NCCL_MAX_NCHANNELS"] == "16"

    # 5 % faster wins
    for k in rep["chosen_env"]:
        monkeypatch.delenv(k, raising=False)
    outputs.update(hwq2_ch16_torch=line(9.5), hwq1_ch32_torch=line(9.9), hwq1_ch16_native=line(10.4))
    rep = bench.autotune(args, 0, 2)
    assert rep["chosen"] == "hwq2_ch16_torch" and os.environ["GPU_MAX_HW_QUEUES"] == "2"

    # a factor the caller fixed is not a factor: the hw-queue candidate collapses onto the default and is not run
    for k in rep["chosen_env"]:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    outputs.clear()
    outputs.update(hwq1_ch16_torch=line(10.0), hwq1_ch32_torch=line(9.0), hwq1_ch16_native=line(9.9))

    class FakeProcFixed(FakeProc):
        def __init__(self, cmd, env=None, **kw):
            assert env["GPU_MAX_HW_QUEUES"] == "4"
            label = next(lb for lb, c in bench.AUTOTUNE_CANDIDATES if lb in outputs and
                         all(env.get(k) == v for k, v in c.items() if k != "GPU_MAX_HW_QUEUES"))
            self.out, self.returncode, self.pid = outputs[label], 0, 1

    monkeypatch.setattr(sp, "Popen", FakeProcFixed)
    rep = bench.autotune(args, 0, 2)
    assert [c["label"] for c in rep["candidates"]] == ["hwq1_ch16_torch", "hwq1_ch32_torch", "hwq1_ch16_native"]
    assert rep["chosen"] == "hwq1_ch32_torch" and rep["fixed_by_caller"] == {"GPU_MAX_HW_QUEUES": "4"}
    assert os.environ["GPU_MAX_HW_QUEUES"] == "4" and os.environ["NCCL_MAX_NCHANNELS"] == "32"
    # no usable candidate at all: the defaults stay and the report says so
    outputs.update(hwq1_ch16_torch="", hwq1_ch32_torch="", hwq1_ch16_native="")
    for k in ("NCCL_MAX_NCHANNELS", "NST_DIST_NATIVE"):
        monkeypatch.delenv(k, raising=False)
    rep = bench.autotune(args, 0, 2)
    assert rep["chosen"] is None and rep["chosen_env"] == {} and "NCCL_MAX_NCHANNELS" not in os.environ
