"""Host-side logic of the path on CPU: the hand-scheduled forward / backward of the layer classes, the gradient
bookkeeping in the flat buffer, the train step and the data-parallel hooks are ordinary Python -- they are exercised
here WITHOUT a GPU by installing oracle/kernel_emulation.py (a float64 restatement of the C-ABI contracts) over
`neurst_amd.kernels` for the duration of a test, and compared with oracle/neurst_oracle.py (autograd).

This does not test the HIP kernels (tests/test_gpu_*.py do, through the C ABI); it pins the orchestration around
them, so that a scheduling mistake in a backward pass shows up in the `-m "not gpu"` suite already."""
import inspect
import os
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import kernel_emulation as E
from oracle import neurst_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def cpu_kernels(monkeypatch):
    return E.install(monkeypatch)


def rel_err(got, ref):
    got, ref = got.detach().double(), ref.detach().double()
    assert got.shape == ref.shape, f"{tuple(got.shape)} vs {tuple(ref.shape)}"
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)


def test_emulation_covers_every_kernel_entry_point(cpu_kernels):
    """Every function of neurst_amd.kernels that reaches the library has an emulated namesake with the same signature."""
    from neurst_amd import kernels as K
    patched = {n for n, f in vars(K).items() if inspect.isfunction(f) and f.__module__ == "oracle.kernel_emulation"}
    assert patched == set(cpu_kernels)
    import ast
    text = open(os.path.join(ROOT, "neurst_amd", "kernels.py")).read()
    tree = ast.parse(text)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and "lib.nst_" in ast.get_source_segment(text, node) \
                and not node.name.startswith("_") and node.name != "probe_mfma":
            assert node.name in cpu_kernels, f"kernels.{node.name} has no CPU emulation"
            want = [a.arg for a in node.args.args]
            got = list(inspect.signature(getattr(E, node.name)).parameters)
            assert got == want, f"{node.name}: emulation signature {got} != {want}"
    assert K.gemm is E.gemm


def _speech_model(case, dropout=0.0, **extra):
    cases = {  # d, H, enc, dec, ffn, C, B, T, F, L, V, ragged
        "toy": (8, 2, 2, 2, 10, 5, 2, 11, 80, 3, 5, False),
        "small": (32, 2, 2, 2, 64, 8, 3, 38, 16, 9, 50, True),
        "row": (256, 4, 2, 2, 128, 8, 2, 22, 16, 5, 40, True),      # d_model 256: the whole-row products take their calls
    }
    d, H, ne, nd, ffn, C, B, T, F, L, V, ragged = cases[case]
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    p = dict(get_hyper_parameters("speech_transformer_toy")["model.params"])
    p.update({"modality.dim": d, "modality.source.channels": C, "encoder.num_layers": ne, "decoder.num_layers": nd,
              "encoder.hidden_size": d, "decoder.hidden_size": d, "encoder.num_attention_heads": H,
              "decoder.num_attention_heads": H, "encoder.filter_size": ffn, "decoder.filter_size": ffn})
    for k in list(p):
        if k.endswith("dropout_rate"):
            p[k] = dropout
    dtype = extra.pop("dtype", "float32")
    p.update(extra)
    model = build_model({"model.class": "SpeechTransformer", "model.params": p},
                        {"audio_feature_dim": F, "audio_feature_channels": 1},
                        {"vocab_size": V, "eos_id": V - 1, "bos_id": V - 2, "unk_id": V - 3}, device="cpu",
                        dtype=dtype, init_seed=3)
    g = torch.Generator().manual_seed(11)
    sd = {}
    for n, prm in model.store.params.items():
        if n.endswith("/bias") or n.endswith("/beta"):
            sd[n] = torch.randn(prm.shape, generator=g) * 0.05
        elif n.endswith("/gamma"):
            sd[n] = 1.0 + torch.randn(prm.shape, generator=g) * 0.1
    model.store.load_state_dict(sd, strict=False)
    cfg = {"num_enc": ne, "num_dec": nd, "num_heads": H, "layer_norm": True}
    return model, cfg, (B, T, F, L, V, ragged)


def _speech_inputs(shape, seed=11):
    B, T, F, L, V, ragged = shape
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(B, T, F, 1, generator=g)
    if ragged:
        src_len = torch.tensor([T - (i * T) // (2 * B) for i in range(B)])
        trg_len = torch.tensor([L - i for i in range(B)]).clamp(min=1)
    else:
        src_len, trg_len = torch.full((B,), T), torch.full((B,), L)
    trg = torch.randint(0, V - 3, (B, L), generator=g)
    trg = torch.where(torch.arange(L)[None] >= (trg_len[:, None] - 1), torch.full_like(trg, V - 1), trg)
    trg_input = torch.cat([torch.full((B, 1), V - 2), trg[:, :-1]], 1)
    return {"src": src, "src_length": src_len, "trg": trg, "trg_input": trg_input, "trg_length": trg_len}


def _oracle_step(model, inputs, cfg, fn=O.train_step_reference):
    W = {n: p.data.detach().clone().double() for n, p in model.store.params.items()}
    return fn(W, {k: (v.double() if v.is_floating_point() else v) for k, v in inputs.items()}, cfg, 0.1)


@pytest.mark.parametrize("case", ["toy", "small"])
def test_speech_transformer_host_schedule_matches_oracle(cpu_kernels, case):
    """logits, loss and every parameter gradient of the layer classes' explicit forward / backward == oracle autograd."""
    from neurst_amd.criterions import build_criterion
    model, cfg, shape = _speech_model(case)
    inputs = _speech_inputs(shape)
    loss_ref, logits_ref, grads_ref = _oracle_step(model, inputs, cfg)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(inputs, is_training=True)
    loss = crit.reduce_loss(inputs, logits)
    model.backward(crit.backward())
    assert rel_err(logits, logits_ref) < 1e-5
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    for n, p in model.store.params.items():
        assert rel_err(p.grad, grads_ref[n]) < 2e-5, n
    # the padding elements between parameters of the flat buffer stay zero (the reducer sends them too)
    used = torch.zeros(model.store.total, dtype=torch.bool)
    for p in model.store.params.values():
        used[p.offset:p.offset + p.numel] = True
    assert float(model.store.grad[~used].abs().sum()) == 0.0


def test_grouped_cross_attention_projection_equals_per_layer_and_leaves_nothing_behind(cpu_kernels, monkeypatch):
    """TransformerDecoder's grouped k|v projection (one GEMM over the packed kv_transform kernels, transformer_decoder.KV_GROUP) against the
    per-layer projections: same logits and gradients; and the per-layer hand-over attributes (_kv_pre / _dkv_out) never survive
    a call -- a training forward that dies inside the decoder must not feed its k|v to the next (teacher-forced, not grouped)
    evaluation forward of a DIFFERENT batch."""
    from neurst_amd.criterions import build_criterion
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    outs = []
    for grouped in ("1", "0"):
        monkeypatch.setattr("neurst_amd.layers.decoders.transformer_decoder.KV_GROUP", grouped == "1")
        model, cfg, shape = _speech_model("small")
        assert (model._decoder._kv_group is not None) == (grouped == "1")
        inputs = _speech_inputs(shape)
        logits = model(inputs, is_training=True)
        crit.reduce_loss(inputs, logits)
        model.backward(crit.backward())
        outs.append((logits.double(), model.store.grad.clone().double()))
        assert all(getattr(a, "_kv_pre", None) is None and getattr(a, "_dkv_out", None) is None for a in model._decoder._kv_atts)
    assert rel_err(outs[0][0], outs[1][0]) < 1e-5 and rel_err(outs[0][1], outs[1][1]) < 1e-5
    monkeypatch.setattr("neurst_amd.layers.decoders.transformer_decoder.KV_GROUP", True)
    model, cfg, shape = _speech_model("small")
    inputs, other = _speech_inputs(shape), _speech_inputs(shape, seed=99)
    want = model(other, is_training=False).double()
    last = model._decoder._stacking_layers[-1]
    real = last.forward
    monkeypatch.setattr(last, "forward", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("injected")))
    with pytest.raises(RuntimeError, match="injected"):
        model(inputs, is_training=True)
    monkeypatch.setattr(last, "forward", real)
    assert all(getattr(a, "_kv_pre", None) is None for a in model._decoder._kv_atts)
    assert rel_err(model(other, is_training=False).double(), want) < 1e-6


@pytest.mark.parametrize("dec_side", [False, True])
@pytest.mark.parametrize("at", ["end", "encoder"])
def test_grouped_weight_gradients_equal_the_per_product_schedule_and_reports_follow_the_launch(cpu_kernels, monkeypatch, at, dec_side):
    """Runtime.wgrad_group / launch_wgrad_group: with the group on, Dense.backward_params only queues its product; the model
    launches the queue once (behind the encoder stack or at the end; dec_side: the decoder stack's products in a launch of their
    own right behind the decoder's backward -- Runtime.launch_wgrad_group(side=True), the speech models' placement).  Same gradients as the per-product schedule (bit-identical over
    the emulated kernels), nothing left queued after backward(), and a data-parallel report for a layer is delivered only
    AFTER the launch that writes that layer's weight gradients -- in the original order, each exactly once."""
    from neurst_amd import kernels as K
    from neurst_amd.criterions import build_criterion
    monkeypatch.setattr(K.WgradGroup, "MIN_OUTPUTS", 1)
    monkeypatch.setattr("neurst_amd.models.encoder_decoder_model._WGRAD_GROUP_AT", at)
    monkeypatch.setattr("neurst_amd.models.encoder_decoder_model._WGRAD_DECODER_SIDE", dec_side)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    grads, reports = [], []
    for grouped in (False, True):
        model, cfg, shape = _speech_model("small", dtype="bfloat16")
        model.rt._wgrad_group_on_cpu = grouped
        inputs = _speech_inputs(shape)
        seen = []
        launches = []
        if grouped:
            real = K.gemm_wgrad_group
            monkeypatch.setattr(K, "gemm_wgrad_group", lambda items, table=None: (launches.append(len(items)), real(items, table))[1])
        model.grad_ready_hook = lambda prefixes: seen.append((tuple(prefixes), len(model.rt.wgrad_group() or ())))
        logits = model(inputs, is_training=True)
        crit.reduce_loss(inputs, logits)
        model.backward(crit.backward())
        grads.append(model.store.grad.clone())
        reports.append([p for p, _ in seen])
        if grouped:
            assert len(model.rt.wgrad_group()) == 0
            assert all(pending == 0 for _, pending in seen), "a report ran while its weight gradients were still queued"
            # 2 encoder layers x (qkv, out, ffn1, ffn2) + 2 decoder layers x (qkv, out, q, out, ffn1, ffn2); the cross-attention
            # k|v projections and the front dense layer (long, few tiles) stay on the per-product path
            assert launches == ([2 * 6, 2 * 4] if dec_side else [2 * 4 + 2 * 6]), launches
    assert torch.equal(grads[0], grads[1])
    assert reports[0] == reports[1]


def test_aborted_backward_leaves_no_weight_gradient_products_or_reports_for_the_next_step(cpu_kernels, monkeypatch):
    """A backward pass that dies between Dense.backward_params (queued for the grouped launch) and launch_wgrad_group must not
    leave its products or its deferred reducer reports behind: the next step's launch would run them on a dead batch's
    activations with stale accumulate flags, and the stale reports would make this rank issue extra collectives."""
    from neurst_amd import kernels as K
    from neurst_amd.criterions import build_criterion
    monkeypatch.setattr(K.WgradGroup, "MIN_OUTPUTS", 1)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    model, cfg, shape = _speech_model("small", dtype="bfloat16")
    model.rt._wgrad_group_on_cpu = True
    inputs = _speech_inputs(shape)
    seen = []
    model.grad_ready_hook = lambda prefixes: seen.append(tuple(prefixes))
    logits = model(inputs, is_training=True)
    crit.reduce_loss(inputs, logits)
    dlogits = crit.backward()
    model.backward(dlogits)
    want, want_reports = model.store.grad.clone(), list(seen)
    # the same step again, but the encoder's backward dies after the decoder queued its products and deferred its reports
    logits = model(inputs, is_training=True)
    crit.reduce_loss(inputs, logits)
    real = model._encoder.backward
    monkeypatch.setattr(model._encoder, "backward", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("injected")))
    del seen[:]
    with pytest.raises(RuntimeError, match="injected"):
        model.backward(crit.backward())
    assert len(model.rt.wgrad_group()) == 0 and not model.rt._deferred_reports
    monkeypatch.setattr(model._encoder, "backward", real)
    # a survivor planted by hand (as if the clean-up of the failing pass itself had been skipped) is dropped with a warning
    model.rt.wgrad_group().add(torch.ones(8, 8, dtype=torch.bfloat16), torch.ones(8, 8, dtype=torch.bfloat16),
                               torch.zeros(8, 8), False)
    model.rt._deferred_reports.append(lambda: seen.append("stale"))
    del seen[:]
    logits = model(inputs, is_training=True)
    crit.reduce_loss(inputs, logits)
    with pytest.warns(UserWarning, match="aborted backward"):
        model.backward(crit.backward())
    assert torch.equal(model.store.grad, want) and seen == want_reports


@pytest.mark.parametrize("variant", ["post_norm", "post_norm_encoder_only", "untied_softmax", "post_norm_untied"])
def test_post_norm_and_untied_softmax_host_schedule(cpu_kernels, variant):
    """post_normalize (common_layers.py:86-92; no output_ln, transformer_encoder.py:97-100) and the separate
    softmax_linear projection (encoder_decoder_model.py:63-67, 180-185) against the oracle's autograd."""
    from neurst_amd.criterions import build_criterion
    extra, cfg_extra = {}, {}
    if variant.startswith("post_norm"):
        extra["encoder.post_normalize"] = True
        cfg_extra["encoder_post_normalize"] = True
        if variant != "post_norm_encoder_only":
            extra["decoder.post_normalize"] = True
            cfg_extra["decoder_post_normalize"] = True
    if "untied" in variant:
        extra["modality.share_embedding_and_softmax_weights"] = False
    model, cfg, shape = _speech_model("small", **extra)
    cfg.update(cfg_extra)
    names = set(model.store.params)
    assert ("TransformerEncoder/output_ln/gamma" in names) == ("encoder_post_normalize" not in cfg_extra)
    assert ("TransformerDecoder/output_ln/gamma" in names) == ("decoder_post_normalize" not in cfg_extra)
    if "untied" in variant:
        assert {"softmax_linear/kernel", "softmax_linear/bias", "target_symbol_modality/emb/weights"} <= names
        assert not any(n.startswith("target_symbol_modality/shared/") for n in names)
        assert tuple(model.store.params["softmax_linear/kernel"].shape) == (32, 50)
    inputs = _speech_inputs(shape)
    loss_ref, logits_ref, grads_ref = _oracle_step(model, inputs, cfg)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(inputs, is_training=True)
    loss = crit.reduce_loss(inputs, logits)
    fired = []
    model.grad_ready_hook = fired.append
    model.backward(crit.backward())
    assert rel_err(logits, logits_ref) < 1e-5 and abs(float(loss) - float(loss_ref)) < 1e-5
    for n, p in model.store.params.items():
        assert rel_err(p.grad, grads_ref[n]) < 5e-5, n
    if "untied" in variant:  # decoder + softmax_linear are reported together and form one contiguous slice
        from neurst_amd.training.distributed import GradientReducer
        comp = ["TransformerDecoder/", "softmax_linear/"]
        n_dec = len(model._decoder._stacking_layers)    # per-layer reports first (last layer first), then the component
        # (the top layer's report carries output_ln, which is registered right behind it)
        want = [[f"TransformerDecoder/layer_{i}/"] + (["TransformerDecoder/output_ln/"] if i == n_dec - 1 and model._decoder._output_norm_layer is not None else [])
                for i in range(n_dec - 1, -1, -1)]
        assert fired[:n_dec] == want and fired[n_dec] == comp
        s, e = GradientReducer(model.store).range_of(comp)
        inside = [p for p in model.store.params.values() if s <= p.offset < e]
        assert all(p.name.startswith(("TransformerDecoder/", "softmax_linear/")) for p in inside)
        assert e == model.store.total
    # inference path agrees with the training path without dropout
    assert rel_err(model(inputs, is_training=False), logits_ref) < 1e-5


@pytest.mark.parametrize("tag", ["neurst_pt_st_1x1", "neurst_pt_st_2x2_ragged", "neurst_pt_st_2x2_postnorm_untied"])
def test_host_path_matches_the_reference_neurst_pt_speech_transformer(cpu_kernels, tag):
    """Logits, loss and every gradient of the layer classes (over the emulated kernels) against the reference's own
    PyTorch SpeechTransformer + autograd (tests/golden/make_golden.py: gen_neurst_pt_speech_transformer), including its
    post-norm stacks with untied logits."""
    from conftest import build_speech_model_for_reference_case, load_reference_pt_case
    from neurst_amd.criterions import build_criterion
    inputs, W, cfg, logits_ref, loss_ref, grads_ref = load_reference_pt_case(tag)
    model = build_speech_model_for_reference_case(W, cfg, logits_ref, "cpu")
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(inputs, is_training=True)
    loss = crit.reduce_loss(inputs, logits)
    model.backward(crit.backward())
    assert float((logits - logits_ref).abs().max()) < 5e-6 and abs(float(loss) - loss_ref) < 1e-6
    for n, g in grads_ref.items():
        assert rel_err(model.store.params[n].grad, g) < 2e-5, n
    # incremental decoding against the reference's own cached decoding of the same target prefix
    from conftest import load_golden
    steps_ref = torch.from_numpy(load_golden(tag)[0]["expected_step_logits"])
    fn, init, _ = model.get_symbols_to_logits_fn({k: v for k, v in inputs.items() if k.startswith("src")}, beam_size=1,
                                                 decode_padded_length=8)
    for t in range(steps_ref.shape[1]):
        got = fn(inputs["trg_input"][:, t], init["decoder_internal_cache"], t)
        assert float((got - steps_ref[:, t]).abs().max()) < 5e-6, t


@pytest.mark.parametrize("tag", ["neurst_pt_tr_2x2", "neurst_pt_tr_2x2_shared"])
def test_host_path_matches_the_reference_neurst_pt_text_transformer(cpu_kernels, tag):
    from conftest import build_text_model_for_reference_case, load_reference_pt_text_case
    from neurst_amd.criterions import build_criterion
    inputs, W, cfg, logits_ref, loss_ref, grads_ref, share = load_reference_pt_text_case(tag)
    model = build_text_model_for_reference_case(W, logits_ref, share, "cpu")
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(inputs, is_training=True)
    loss = crit.reduce_loss(inputs, logits)
    model.backward(crit.backward())
    assert float((logits - logits_ref).abs().max()) < 5e-6 and abs(float(loss) - loss_ref) < 1e-6
    for n, g in grads_ref.items():
        assert rel_err(model.store.params[n].grad, g) < 2e-5, n


@pytest.mark.parametrize("variant", ["pre_norm", "post_norm"])
def test_training_step_with_dropout_regenerates_the_same_masks_in_backward(cpu_kernels, variant):
    """Dropout 0.3 at every site (attention probabilities, FFN hidden, wrapper outputs, encoder / decoder inputs): the
    backward pass regenerates each site's mask from (step seed, site id) -- partly inside the LayerNorm backward kernel
    that emits the masked gradient for the next sublayer.  The oracle runs with exactly those masks (oracle/philox.py
    restates the kernels' generator), so logits, loss and every gradient must agree as tightly as without dropout."""
    from neurst_amd.criterions import build_criterion
    from oracle import philox
    extra = {"encoder.post_normalize": True, "decoder.post_normalize": True} if variant == "post_norm" else {}
    model, cfg, shape = _speech_model("small", dropout=0.3, **extra)
    if variant == "post_norm":
        cfg.update({"encoder_post_normalize": True, "decoder_post_normalize": True})
    cfg["dropout"] = 0.3
    inputs = _speech_inputs(shape)
    model.rt.step = 5
    masks = philox.SiteMasks(model.rt.step_seed, philox.model_dropout_sites(model))
    W = {n: p.data.detach().clone().double() for n, p in model.store.params.items()}
    loss_ref, logits_ref, grads_ref = O.train_step_reference(
        W, {k: (v.double() if v.is_floating_point() else v) for k, v in inputs.items()}, cfg, 0.1, is_training=True,
        generator=masks)
    assert len(set(masks.used)) == len(masks.sites) == 2 + 2 * (2 + 2) + 2 * (3 + 3)   # every site was visited once
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(inputs, is_training=True)
    loss = crit.reduce_loss(inputs, logits)
    model.backward(crit.backward())
    assert rel_err(logits, logits_ref) < 1e-5 and abs(float(loss) - float(loss_ref)) < 1e-5
    for n, p in model.store.params.items():
        assert rel_err(p.grad, grads_ref[n]) < 5e-5, n
    # and the masks matter: the eval-mode logits differ
    assert rel_err(model(inputs, is_training=False), logits_ref) > 1e-2


@pytest.mark.parametrize("variant", ["length", "padding_mask"])
@pytest.mark.parametrize("ls", [0.0, 0.1, 0.35])
def test_criterion_class_matches_the_reference_code(cpu_kernels, ls, variant):
    """The criterion class (weights from trg_length, or from trg_padding * mask; metrics; backward with the device-side
    1 / sum(tokens)) against the reference's own criterion code (tests/golden/criterion_reference.npz)."""
    from conftest import load_golden
    from neurst_amd.criterions import build_criterion
    r, _ = load_golden("criterion_reference")
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": ls}})
    inp = {"trg": torch.from_numpy(r["trg"])}
    if variant == "length":
        inp["trg_length"] = torch.from_numpy(r["trg_length"])
    else:
        inp.update({"trg_padding": torch.from_numpy(r["trg_padding"]), "mask": torch.from_numpy(r["mask"])})
    logits = torch.from_numpy(r["logits"])
    key = f"ls{ls}_{variant}"
    nll_sum, n_samples, n_tokens = crit(inp, logits)
    assert torch.allclose(nll_sum, torch.from_numpy(r[key + ":nll_sum"]), rtol=2e-6, atol=2e-6)
    assert n_samples.tolist() == r[key + ":n_samples"].tolist() and n_tokens.tolist() == r[key + ":n_tokens"].tolist()
    m = crit.reduce_metrics([(nll_sum, n_samples, n_tokens)])
    assert abs(m["NLL"] - r[key + ":metrics"][0]) < 1e-5 and abs(m["PPL"] / r[key + ":metrics"][1] - 1) < 1e-5
    loss = crit.reduce_loss(inp, logits)
    assert abs(float(loss) - float(r[key + ":loss"])) < 2e-6
    assert float((crit.backward() - torch.from_numpy(r[key + ":dlogits"])).abs().max()) < 2e-7


def test_gradient_accumulation_and_clipping_on_cpu(cpu_kernels):
    """TrainStep with update_cycle = 2 averages the micro-batch gradients (gradaccum_keras_model.py:62-109), clips the
    averaged gradients per tensor (:228-233) and applies Keras Adam -- against the oracle's functions."""
    from neurst_amd.criterions import build_criterion
    from neurst_amd.optimizers.adam import Adam
    from neurst_amd.training.train_step import TrainStep
    model, cfg, shape = _speech_model("toy")
    b1, b2 = _speech_inputs(shape, 5), _speech_inputs(shape, 6)
    W0 = {n: p.data.detach().clone() for n, p in model.store.params.items()}
    _, _, g1 = _oracle_step(model, b1, cfg)
    _, _, g2 = _oracle_step(model, b2, cfg)
    mean = {n: (g1[n] + g2[n]) / 2 for n in g1}
    clipped = O.clip_gradients(mean, clip_norm=0.05)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    opt = Adam(model.store, learning_rate=1e-2, beta_1=0.9, beta_2=0.98, epsilon=1e-9)
    step = TrainStep(model, crit, opt, None, update_cycle=2, clip_norm=0.05)
    step([b1, b2])
    for n, p in model.store.params.items():
        assert rel_err(p.grad, clipped[n]) < 2e-5, n
        # Adam is checked on the gradient the path itself produced: the first Keras-Adam step is ~lr * sign(g), which
        # turns fp32 rounding of near-zero components into O(lr) differences if the oracle's gradient were used
        w, _, _ = O.keras_adam_step(W0[n].double(), p.grad.double(), torch.zeros_like(W0[n]).double(),
                                    torch.zeros_like(W0[n]).double(), 1, 1e-2)
        assert float((p.data.double() - w).abs().max()) < 1e-6, n
    assert model.rt.step == 1 and opt.iterations == 1


def test_text_transformer_and_waitk_host_schedule(cpu_kernels):
    from neurst_amd.criterions import build_criterion
    from neurst_amd.models.transformer import Transformer
    from neurst_amd.tasks import build_task
    from neurst_amd.utils import compat
    d, H, ne, nd, ffn, B, S, L, Vs, Vt = 16, 2, 2, 2, 32, 3, 9, 7, 23, 19
    for wait_k in (None, 2):
        p = dict(Transformer.build_model_args_by_name("transformer_toy")["model.params"])
        p.update({"modality.dim": d, "encoder.num_layers": ne, "decoder.num_layers": nd, "encoder.hidden_size": d,
                  "decoder.hidden_size": d, "encoder.num_attention_heads": H, "decoder.num_attention_heads": H,
                  "encoder.filter_size": ffn, "decoder.filter_size": ffn})
        for k in list(p):
            if k.endswith("dropout_rate"):
                p[k] = 0.0
        task = build_task({"task.class": "translation", "task.params": {"src_vocab_size": Vs, "trg_vocab_size": Vt}})
        if wait_k is None:
            model = task.build_model({"model.class": "Transformer", "model.params": p}, device="cpu", dtype="float32", init_seed=5)
        else:
            model = task.build_model({"model.class": "WaitkTransformer", "model.params": dict(p, wait_k=wait_k)},
                                     device="cpu", dtype="float32", init_seed=5)
        g = torch.Generator().manual_seed(21)

        def side(Lx, V, lens):
            ids = torch.randint(0, V - 3, (B, Lx), generator=g)
            return torch.where(torch.arange(Lx)[None] >= (lens[:, None] - 1), torch.full_like(ids, V - 1), ids)
        src_len = torch.tensor([S - i for i in range(B)])
        trg_len = torch.tensor([L - i for i in range(B)])
        inputs = task.example_to_input({"feature": side(S, Vs, src_len), "label": side(L, Vt, trg_len)}, compat.ModeKeys.TRAIN)
        cfg = {"num_enc": ne, "num_dec": nd, "num_heads": H}
        if wait_k is not None:
            cfg.update({"attention_monotonic": True, "wait_k": wait_k})
        loss_ref, logits_ref, grads_ref = _oracle_step(model, inputs, cfg, O.text_train_step_reference)
        crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
        logits = model(inputs, is_training=True)
        loss = crit.reduce_loss(inputs, logits)
        model.backward(crit.backward())
        assert rel_err(logits, logits_ref) < 1e-5 and abs(float(loss) - float(loss_ref)) < 1e-5
        for n, prm in model.store.params.items():
            assert rel_err(prm.grad, grads_ref[n]) < 2e-5, (wait_k, n)


def test_reference_golden_logits_through_the_host_path(cpu_kernels):
    """The reference's own full-model golden logits (tests/neurst/models/transformer_test.py:23-666) through the layer
    classes + emulated kernels: pins the host wiring (variable names, layouts, call order) on CPU."""
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    r, W = load_golden("transformer_toy_logits")
    model = build_model(get_hyper_parameters("transformer_toy"), dict(vocab_size=8, eos_id=7, bos_id=6, unk_id=5),
                        dict(vocab_size=5, eos_id=4, bos_id=3, unk_id=2), device="cpu", dtype="float32")
    zero = {n: torch.zeros(p.shape) for n, p in model.store.params.items() if n.endswith("/bias")}
    model.store.load_state_dict(zero, strict=False)
    model.store.load_state_dict({k: v for k, v in W.items() if k in model.store.params}, strict=False)
    inputs = {"src": torch.from_numpy(r["src"]), "src_padding": torch.from_numpy(r["src_padding"]),
              "trg_input": torch.from_numpy(r["trg_input"])}
    logits = model(inputs, is_training=False)
    assert float(((logits.double() - torch.from_numpy(r["expected"]).double()) ** 2).sum()) < 1e-9


def test_incremental_decoding_equals_teacher_forcing_on_cpu(cpu_kernels):
    """The per-layer K/V caches of the decoder (multi_head_attention.py:254-290): step t of the incremental path
    reproduces position t of the full teacher-forced forward."""
    model, cfg, shape = _speech_model("toy")
    inputs = _speech_inputs(shape)
    full = model(inputs, is_training=False)
    enc_inputs = {k: v for k, v in inputs.items() if k.startswith("src")}
    step_fn, init, _ = model.get_symbols_to_logits_fn(enc_inputs, beam_size=1, decode_padded_length=8)
    assert init["decoder_input"].tolist() == [shape[4] - 2] * shape[0]   # BOS
    cache = init["decoder_internal_cache"]
    for t in range(inputs["trg_input"].shape[1]):
        logits_t = step_fn(inputs["trg_input"][:, t], cache, t)
        assert rel_err(logits_t, full[:, t]) < 1e-5, t


# ------------------------------------------------------------------------------------------------ data parallel, gloo, world 2
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from neurst_amd import kernels as K
    for n in E._NAMES:
        setattr(K, n, getattr(E, n))
    from neurst_amd.criterions import build_criterion
    from neurst_amd.optimizers.adam import Adam
    from neurst_amd.training.distributed import GradientReducer, init_distributed
    from neurst_amd.training.train_step import TrainStep
    init_distributed(backend="gloo")
    model, cfg, shape = _speech_model("toy")
    red = GradientReducer(model.store, bucket_bytes=2048)
    red.broadcast_parameters(0)
    fired = []
    orig = red.component_ready
    red.component_ready = lambda prefixes: (fired.append(list(prefixes)), orig(prefixes))[1]
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    opt = Adam(model.store, learning_rate=1e-2, beta_1=0.9, beta_2=0.98, epsilon=1e-9)
    step = TrainStep(model, crit, opt, red)
    losses = [float(step(_dp_inputs(shape, s, rank, world))) for s in range(2)]
    q.put((rank, model.store.master.numpy().copy(), losses, fired))   # numpy: no fd passing after exit
    dist.destroy_process_group()


def _dp_inputs(shape, step, rank, world):
    """The batch of one rank.  With more than two ranks the ranks hold UNEQUAL numbers of target tokens (rank r keeps
    max(1, L - r) tokens per utterance): the exchanged quantity is then the documented mean over ranks of per-rank
    token-mean gradients (hvd.Average of each rank's own mean loss, hvd_utils.py:46-62 + label_smoothed_cross_entropy.py:
    94-157), NOT the token mean over the global batch."""
    inp = _speech_inputs(shape, 100 + 10 * step + rank)
    if world > 2:
        B, T, F, L, V, _ = shape
        trg_len = torch.full((B,), max(1, L - rank))
        trg = torch.where(torch.arange(L)[None] >= (trg_len[:, None] - 1), torch.full_like(inp["trg"], V - 1), inp["trg"])
        inp.update(trg=trg, trg_length=trg_len, trg_input=torch.cat([torch.full((B, 1), V - 2), trg[:, :-1]], 1))
    return inp


@pytest.mark.parametrize("world", [2, 4, 8])
def test_data_parallel_train_step_gloo_matches_oracle_average(world):
    """N ranks, different batches (N = 4: unequal token counts per rank, _dp_inputs): after two steps every rank holds the
    weights the oracle gets from the MEAN of the per-rank gradients (hvd.Average, hvd_utils.py:46-62) under Keras Adam; the
    component hooks fire in backward order."""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=360) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [(r, torch.from_numpy(w), l, f) for r, w, l, f in res]
    assert all(torch.equal(res[0][1], r[1]) for r in res[1:]), "ranks diverged"
    if world > 2:    # the ranks really hold different token counts (the toy case: L = 3 -> 3, 2, 1, 1 tokens per utterance)
        assert len({int(_dp_inputs(_speech_model("toy")[2], 0, r, world)["trg_length"].sum()) for r in range(world)}) >= 3
    # per-layer reports (last layer first) followed by the component report that sweeps up output_ln
    comps = [f[0] for f in res[0][3] if "/layer_" not in f[0]]
    assert all(res[0][3] == r[3] for r in res[1:]) and comps[:4] == [
        "TransformerDecoder/", "target_symbol_modality/", "TransformerEncoder/", "input_audio_modality/"]
    first = [f[0] for f in res[0][3]]
    assert first[0].startswith("TransformerDecoder/layer_") and first.index("TransformerEncoder/") > first.index("TransformerEncoder/layer_0/")

    # single-process oracle of the same two steps
    model, cfg, shape = _speech_model("toy")   # same init seed as the workers (rank 0's weights are broadcast)
    W = {n: p.data.detach().clone().double() for n, p in model.store.params.items()}
    m = {n: torch.zeros_like(w) for n, w in W.items()}
    v = {n: torch.zeros_like(w) for n, w in W.items()}
    # Adam divides by sqrt(v): where the true gradient is zero (e.g. the key bias, softmax is shift invariant) the update
    # is lr * sign(rounding noise); those elements are excluded from the weight comparison
    solid = {n: torch.ones_like(w, dtype=torch.bool) for n, w in W.items()}
    for s in range(2):
        per_rank = []
        for rank in range(world):
            inp = _dp_inputs(shape, s, rank, world)
            loss, _, g = O.train_step_reference(W, {k: (t.double() if t.is_floating_point() else t) for k, t in inp.items()}, cfg, 0.1)
            per_rank.append(g)
            assert abs(float(loss) - res[rank][2][s]) < 1e-5
        names = sorted(W)
        avg = dict(zip(names, O.average_gradients([[g[n] for n in names] for g in per_rank])))
        gmax = max(float(g.abs().max()) for g in avg.values())
        for n in W:
            solid[n] &= avg[n].abs() > 1e-4 * gmax
            W[n], m[n], v[n] = O.keras_adam_step(W[n], avg[n].double(), m[n], v[n], s + 1, 1e-2)
    assert sum(int(x.sum()) for x in solid.values()) > 0.5 * sum(x.numel() for x in solid.values())
    for n, p in model.store.params.items():
        got = res[0][1][p.offset:p.offset + p.numel].view(p.shape).double()
        assert float(((got - W[n]).abs() * solid[n]).max()) < 2e-5, n


# ------------------------------------------------------------------------------------------------ streaming wait-k
def _waitk_model(wait_k=2, d=16, H=2, post_norm=False):
    from neurst_amd.models.transformer import Transformer
    from neurst_amd.tasks import build_task
    p = dict(Transformer.build_model_args_by_name("transformer_toy")["model.params"])
    p.update({"modality.dim": d, "encoder.hidden_size": d, "decoder.hidden_size": d, "encoder.num_attention_heads": H,
              "decoder.num_attention_heads": H, "encoder.filter_size": 2 * d, "decoder.filter_size": 2 * d,
              "encoder.post_normalize": post_norm, "decoder.post_normalize": post_norm})
    for k in list(p):
        if k.endswith("dropout_rate"):
            p[k] = 0.0
    task = build_task({"task.class": "translation", "task.params": {"src_vocab_size": 23, "trg_vocab_size": 19}})
    model = task.build_model({"model.class": "WaitkTransformer", "model.params": dict(p, wait_k=wait_k)}, device="cpu",
                             dtype="float32", init_seed=5)
    g = torch.Generator().manual_seed(3)
    sd = {n: torch.randn(prm.shape, generator=g) * 0.1 for n, prm in model.store.params.items()
          if n.endswith("/bias") or n.endswith("/beta")}
    model.store.load_state_dict(sd, strict=False)
    return model


@pytest.mark.parametrize("post_norm", [False, True])
def test_streaming_encoder_chunks_equal_the_monotonic_full_forward(cpu_kernels, post_norm):
    """TransformerEncoder.incremental_encode (transformer_encoder.py:138-175): chunks of 3, 1, 1 and 4 positions over the
    cached keys / values reproduce the rows of the full attention_monotonic forward."""
    model = _waitk_model(post_norm=post_norm)
    enc = model._encoder
    g = torch.Generator().manual_seed(9)
    B, S, d = 2, 9, 16
    x = torch.randn(B, S, d, generator=g)
    full = enc.forward(x, torch.zeros(B, S), is_training=False)
    cache, t, outs = {}, 0, []
    for n in (3, 1, 1, 4):
        chunk = x[:, t] if n == 1 and t == 3 else x[:, t:t + n].contiguous()   # also the 2-d single-position form
        out, cache = enc.incremental_encode(chunk, cache, time=t, max_length=16)
        outs.append(out)
        t += n
    assert rel_err(torch.cat(outs, 1), full) < 1e-5
    with pytest.raises(ValueError):
        enc.incremental_encode(x[:, :1].contiguous(), cache, time=3)


@pytest.mark.parametrize("wait_k", [1, 3])
def test_streaming_waitk_decoding_equals_offline_waitk_decoding(cpu_kernels, wait_k):
    """WaitkTransformer.incremental_encode / incremental_decode (waitk_transformer.py:117-139) driven by the wait-k
    read / write schedule: step t reads until k + t source tokens are encoded, then decodes one position.  With a
    monotonic encoder this equals the offline path, whose step t masks the fully encoded source beyond k + t."""
    model = _waitk_model(wait_k=wait_k)
    g = torch.Generator().manual_seed(4)
    S, L = 7, 9
    src = torch.randint(0, 20, (1, S), generator=g)
    trg_in = torch.cat([torch.tensor([[17]]), torch.randint(0, 16, (1, L - 1), generator=g)], 1)
    # offline: full encoder pass, incremental decoder with lagging k + t
    step_fn, init, _ = model.get_symbols_to_logits_fn({"src": src, "src_length": torch.tensor([S])}, beam_size=1,
                                                      decode_padded_length=16)
    offline = [step_fn(trg_in[:, t], init["decoder_internal_cache"], t) for t in range(L)]
    # streaming
    enc_cache, dec_cache, read = {}, {}, 0
    for t in range(L):
        want = min(S, wait_k + t)
        if want > read:
            enc_cache, dec_cache = model.incremental_encode(
                {"src": src[:, read:want], "src_length": [want - read]}, enc_cache, dec_cache, time=read,
                max_source_length=8, decode_padded_length=16)
            read = want
        logits, dec_cache = model.incremental_decode(trg_in[:, t], dec_cache, time=t)
        assert dec_cache["memory"].shape[1] == read
        assert rel_err(logits, offline[t]) < 1e-5, t
    assert read == S
    with pytest.raises(RuntimeError):   # the preallocated memory is full
        model.incremental_encode({"src": src[:, :2], "src_length": [2]}, enc_cache, dec_cache, time=read)


class _CharPipeline(object):
    """Stand-in text pipeline for the agent test: a word becomes the word-initial unit '▁<c>' plus one unit per further
    character (the SentencePiece convention the agent's units_to_segment relies on)."""

    def __init__(self, chars):
        self.tokens = ["▁" + c for c in chars] + list(chars) + ["<UNK>", "<SEQ_BEG>", "<SEQ_END>"]
        n = len(self.tokens)
        self.meta = {"vocab_size": n, "unk_id": n - 3, "bos_id": n - 2, "eos_id": n - 1, "pad_id": n - 1, "language": "en"}

    def encode(self, word):
        ids = [self.tokens.index("▁" + word[0])] + [self.tokens.index(c, len(self.tokens) // 2 - 1) for c in word[1:]]
        return ids + [self.meta["eos_id"]]


def test_simul_trans_text_agent_wait_k_policy_and_streaming_predictions(cpu_kernels):
    """End to end on CPU: a toy WaitkTransformer is TRAINED through TrainStep (character-mapping task, 300 steps) and then
    driven by SimulTransTextAgent (simul_trans_text_agent.py:45-245) through the local stand-in of SimulEval's client
    loop.  The policy waits at the WORD level, multi-unit words are drained before the next decision, and every written
    unit is the argmax of the oracle's logits for that position given exactly the source units that had been read (and
    were inside the wait-k window) when it was written."""
    import argparse
    import random
    from neurst_amd.criterions import build_criterion
    from neurst_amd.models.transformer import Transformer
    from neurst_amd.optimizers.adam import Adam
    from neurst_amd.tasks import build_task
    from neurst_amd.training.train_step import TrainStep
    from neurst_amd.utils import compat
    from neurst_amd.utils.simuleval_agents import AGENTS
    from neurst_amd.utils.simuleval_agents import simul_trans_text_agent as A
    src_pipe, trg_pipe = _CharPipeline("abcde"), _CharPipeline("xyz")
    mapping = dict(zip("abcde", "xyzxy"))
    k = 2
    task = build_task({"task.class": "WaitkTranslation",
                       "task.params": {"src_vocab_size": src_pipe.meta["vocab_size"],
                                       "trg_vocab_size": trg_pipe.meta["vocab_size"], "wait_k": k}})
    assert task.get_config()["wait_k"] == k
    task._src_data_pipeline, task._trg_data_pipeline = src_pipe, trg_pipe
    p = dict(Transformer.build_model_args_by_name("transformer_toy")["model.params"])
    p.update({"modality.dim": 16, "encoder.hidden_size": 16, "decoder.hidden_size": 16, "encoder.filter_size": 32,
              "decoder.filter_size": 32})
    for name in list(p):
        if name.endswith("dropout_rate"):
            p[name] = 0.0
    model = task.build_model({"model.class": "WaitkTransformer", "model.params": p}, device="cpu", dtype="float32", init_seed=1)
    assert model.wait_k == k
    rng = random.Random(0)

    def units(pipe, words):
        return [u for w in words for u in pipe.encode(w)[:-1]] + [pipe.meta["eos_id"]]

    def batch(n=16):
        sents = [["".join(rng.choice("abcde") for _ in range(rng.randint(1, 3))) for _ in range(rng.randint(2, 5))]
                 for _ in range(n)]
        src = [units(src_pipe, ws) for ws in sents]
        trg = [units(trg_pipe, ["".join(mapping[c] for c in w) for w in ws]) for ws in sents]

        def pad(rows, eos):
            width = max(map(len, rows))
            return torch.tensor([r + [eos] * (width - len(r)) for r in rows])
        return task.example_to_input({"feature": pad(src, src_pipe.meta["eos_id"]), "label": pad(trg, trg_pipe.meta["eos_id"])},
                                     compat.ModeKeys.TRAIN)
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    step = TrainStep(model, crit, Adam(model.store, learning_rate=5e-3, beta_1=0.9, beta_2=0.98, epsilon=1e-9), None)
    losses = [float(step(batch())) for _ in range(300)]
    assert losses[0] > 2.5 and sum(losses[-20:]) / 20 < 0.6, (losses[0], losses[-1])

    eos = trg_pipe.meta["eos_id"]
    args = argparse.Namespace(wait_k=k, model_dir=None, force_segment=False, max_len=30)
    agent = A.SimulTransTextAgent(args, task=task, models=[model])
    assert AGENTS["simul_trans_text_agent"] is A.SimulTransTextAgent
    reads = []
    orig_predict = agent.predict

    def spy(states):
        out = orig_predict(states)
        reads.append(states.encoding_time)   # source units encoded when this position was decoded
        return out
    agent.predict = spy
    words = ["ab", "cde", "a", "eb", "d"]
    res = A.run_agent_on_sentence(agent, words)
    units_src = units(src_pipe, words)
    assert res["source_units"] == units_src
    y = res["target_units"]
    assert y[-1] == eos and len(reads) == len(y)
    # the trained model mostly solves the mapping even under the wait-2 window
    ref = ["".join(mapping[c] for c in w) for w in words]
    assert len(res["hypothesis"]) == len(ref) and sum(h == r for h, r in zip(res["hypothesis"], ref)) >= 3
    # wait-k at word level: target word t is written when exactly min(k + t, all) source words have been read
    assert res["delays"] == [min(len(words), k + t) for t in range(len(res["delays"]))]
    assert abs(res["average_lagging"] - k) < 1e-9
    assert res["actions"][:k] == [A.READ_ACTION] * k and A.WRITE_ACTION in res["actions"]
    # oracle replay: position i saw min(reads[i], k + i) source units
    W = {n: prm.data.detach().clone().double() for n, prm in model.store.params.items()}
    S = max(reads)
    inputs = {"src": torch.tensor([units_src[:S]]), "src_padding": torch.zeros(1, S, dtype=torch.float64),
              "trg_input": torch.tensor([[trg_pipe.meta["bos_id"]] + y[:-1]])}
    cfg = {"num_enc": 2, "num_dec": 2, "num_heads": 2, "attention_monotonic": True,
           "wait_k": [min(r, k + i) for i, r in enumerate(reads)]}
    logits = O.transformer_logits(inputs, W, cfg)
    assert logits.argmax(-1)[0].tolist() == y
    # the hypothesis is the de-tokenised unit sequence
    text = "".join(trg_pipe.tokens[u] for u in y if u != eos).replace("▁", " ").split()
    assert res["hypothesis"] == text


def test_simuleval_cli_from_a_model_dir(cpu_kernels, tmp_path, capsys):
    """The whole inference flow of examples/simultaneous_translation: a model_dir (model_configs.yml with the task's text
    pipelines + a TensorFlow-format checkpoint) -> simuleval_cli -> agent -> hypotheses + Average Lagging.  Also the
    checkpoint-averaging CLI in front of it (the recipe averages the last checkpoints before evaluating)."""
    import json
    from neurst_amd.cli import simuleval_cli
    from neurst_amd.cli.avg_checkpoint import average_checkpoints
    from neurst_amd.models.transformer import Transformer
    from neurst_amd.tasks import build_task
    from neurst_amd.utils.checkpoints import NameBasedCheckpointManager
    from neurst_amd.utils.configurable import ModelConfigs
    src_vocab = ["ich", "bin", "ein", "haus", "und", "du"]
    trg_vocab = ["▁i", "▁am", "▁a", "▁house", "▁and", "▁you", "s"]
    task_params = {"src_data_pipeline.params": {"vocab_path": src_vocab, "language": "de"},
                   "trg_data_pipeline.params": {"vocab_path": trg_vocab, "language": "en"}, "wait_k": 2}
    task = build_task({"task.class": "WaitkTranslation", "task.params": task_params})
    p = dict(Transformer.build_model_args_by_name("transformer_toy")["model.params"])
    model = task.build_model({"model.class": "WaitkTransformer", "model.params": p}, device="cpu", dtype="float32", init_seed=2)
    run = tmp_path / "run"
    mgr = NameBasedCheckpointManager(model, str(run), max_to_keep=3)
    w0 = model.store.master.clone()
    mgr.save(10)
    model.store.master.mul_(3.0)
    mgr.save(20)
    ModelConfigs.dump({"model.class": "WaitkTransformer", "model.params": model.args, "task.class": "WaitkTranslation",
                       "task.params": task.get_config()}, str(run))
    avg = tmp_path / "avg"
    average_checkpoints(str(run), str(avg))
    (tmp_path / "src.txt").write_text("ich bin ein haus\ndu und ich\n")
    res = simuleval_cli.main(["--source", str(tmp_path / "src.txt"), "--model-dir", str(avg), "--wait-k", "2", "--max-len", "6",
                              "--device", "cpu", "--output", str(tmp_path / "out")])
    report = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert report["sentences"] == 2 and report["AL"] >= 2.0 - 1e-9
    full = [[0, 1, 2, 3, len(src_vocab) + 2], [5, 4, 0, len(src_vocab) + 2]]             # ids + EOS closing the source
    assert len(res) == 2 and all(r["source_units"] == f[:len(r["source_units"])] for r, f in zip(res, full))
    assert res[0]["actions"][:2] == ["read", "read"]
    lines = (tmp_path / "out" / "instances.log").read_text().strip().splitlines()
    assert len(lines) == 2 and json.loads(lines[0])["delays"] == res[0]["delays"]
    # the agent's model holds the AVERAGE of the two checkpoints (w0 and 3 w0 -> 2 w0)
    from neurst_amd.utils.simuleval_agents import simul_trans_text_agent as A
    import argparse
    agent = A.SimulTransTextAgent(argparse.Namespace(wait_k=2, model_dir=str(avg), device="cpu", max_len=6))
    assert torch.allclose(agent.models[0].store.master, 2.0 * w0, rtol=1e-6, atol=1e-7)
    assert agent.models[0].wait_k == 2 and agent.src_pipeline.meta["language"] == "de"


def test_seq_generation_validator_keeps_and_averages_the_best_checkpoints(cpu_kernels, tmp_path):
    """SeqGenerationValidator (seq_generation_validator.py:30-290): decode the validation set with beam search, score with
    the registered metric, keep the best checkpoints under <model_dir>_best, refresh their average, stop after
    `eval_estop_patience` validations without improvement."""
    from neurst_amd.models.transformer import Transformer
    from neurst_amd.tasks import build_task
    from neurst_amd.training import seq_generation_validator  # noqa: F401  (registers the class)
    from neurst_amd.training.criterion_validator import build_validator
    from neurst_amd.utils.checkpoints import latest_checkpoint
    words = ["a", "b", "c", "d", "e"]
    task = build_task({"task.class": "Seq2Seq", "task.params": {"src_data_pipeline.params": {"vocab_path": words},
                                                                 "trg_data_pipeline.params": {"vocab_path": words}}})
    p = dict(Transformer.build_model_args_by_name("transformer_toy")["model.params"])
    model = task.build_model({"model.class": "Transformer", "model.params": p}, device="cpu", dtype="float32", init_seed=4)
    (tmp_path / "src.txt").write_text("a b c\nd e\nc c a b\n")
    (tmp_path / "trg.txt").write_text("a b c\nd e\nc c a b\n")
    model_dir = str(tmp_path / "run")
    v = build_validator({"validator.class": "SeqGenerationValidator", "validator.params": {
        "eval_steps": 10, "eval_dataset.class": "ParallelTextDataset",
        "eval_dataset.params": {"src_file": str(tmp_path / "src.txt"), "trg_file": str(tmp_path / "trg.txt"), "data_is_processed": True},
        "eval_metric.class": "tok_bleu", "eval_search_method.class": "beam_search",
        "eval_search_method.params": {"beam_size": 2, "maximum_decode_length": 6, "extra_decode_length": 2},
        "eval_top_checkpoints_to_keep": 2, "eval_estop_patience": 2, "eval_batch_size": 2}})
    assert v.due(10) and not v.due(15)
    v.build(task, model, model_dir)
    hyps = v.generate()
    assert len(hyps) == 3 and all(isinstance(h, str) for h in hyps)

    scores = iter([5.0, 7.0, 6.0, 6.5])     # the metric's verdicts for steps 10..40 (the toy model itself is random)

    class _Scripted(type(v._gen_metric)):
        def call(self, hypothesis, groundtruth=None):
            assert len(hypothesis) == 3
            return {"tok_bleu": next(scores)}
    v._gen_metric.__class__ = _Scripted
    w0 = model.store.master.clone()
    res = v.validate(10)
    assert res["tok_bleu"] == 5.0 and "NLL" in res and v.gen_best["tok_bleu"] == 5.0
    model.store.master.mul_(2.0)
    v.validate(20)                            # better: second checkpoint, average of both
    assert v.gen_best["tok_bleu"] == 7.0 and latest_checkpoint(model_dir + "_best").endswith("ckpt-20-7.00")
    from neurst_amd.utils import tensor_bundle as tb
    avg = tb.read_bundle(latest_checkpoint(model_dir + "_best_avg"))
    name, prm = next(iter(model.store.params.items()))
    key = tb.checkpoint_key(f"{model.name or model.__class__.__name__}/{name}")
    want = 1.5 * w0[prm.offset:prm.offset + prm.numel].view(prm.shape)
    assert torch.allclose(torch.from_numpy(avg[key]), want, rtol=1e-6, atol=1e-7)
    v.validate(30)
    assert not v.should_stop and v.gen_best["tok_bleu"] == 7.0
    v.validate(40)
    assert v.should_stop and [s for s, _ in v.gen_history] == [10, 20, 30, 40]
    # keep-best rule of the reference's KeepBestCheckpointSaver (checkpoints.py:186-237): a checkpoint is written when fewer
    # than K are kept or its score is >= the WORST kept one -- 6.0 replaced 5.0, 6.5 replaced 6.0; names carry the score
    assert sorted(f for f in os.listdir(model_dir + "_best") if f.endswith(".index")) == ["ckpt-20-7.00.index", "ckpt-40-6.50.index"]
    assert latest_checkpoint(model_dir + "_best").endswith("ckpt-20-7.00")          # model_checkpoint_path = the best kept
    assert sorted(f for f in os.listdir(model_dir + "_best_avg") if f.endswith(".index")) == ["ckpt-20-7.00.index", "ckpt-40-6.50.index"]


def test_cli_flow_train_validate_resume_predict_on_cpu(cpu_kernels, tmp_path):
    """The `neurst-run` flow end to end over the emulated kernels: flag / yaml parsing (run_exp.py), Trainer with a
    SeqGenerationValidator, TensorFlow-format checkpoints + model_configs.yml, resume, then the `predict` entry with a
    metric.  (tests/test_gpu_model.py::test_cli_training_from_tfrecord_shards is the same flow on the MI355X.)"""
    import random
    import yaml
    import neurst_amd.utils.flags_core as flags_core
    from neurst_amd.cli import run_exp as R
    from neurst_amd.data.datasets import build_dataset
    from neurst_amd.exps import build_exp
    from neurst_amd.tasks import build_task
    from neurst_amd.utils.checkpoints import latest_checkpoint
    words = [chr(ord("a") + i) for i in range(8)]
    rng = random.Random(1)
    lines = [" ".join(rng.choice(words) for _ in range(rng.randint(2, 5))) for _ in range(64)]
    for name, rows in (("train", lines), ("dev", lines[:6])):
        (tmp_path / f"{name}.src").write_text("\n".join(rows) + "\n")
        (tmp_path / f"{name}.trg").write_text("\n".join(rows) + "\n")        # copy task
    (tmp_path / "vocab.txt").write_text("\n".join(words) + "\n")
    dp = {"vocab_path": str(tmp_path / "vocab.txt")}
    cfg = {"task.class": "translation",
           "task.params": {"src_data_pipeline.params": dp, "trg_data_pipeline.params": dp, "batch_size": 160, "max_src_len": 12,
                           "max_trg_len": 12, "batch_by_tokens": True},
           "dataset.class": "ParallelTextDataset",
           "dataset.params": {"src_file": str(tmp_path / "train.src"), "trg_file": str(tmp_path / "train.trg"), "data_is_processed": True},
           "entry.class": "trainer",
           "entry.params": {"train_steps": 40, "summary_steps": 20, "save_checkpoint_steps": 20, "optimizer.class": "Adam",
                            "optimizer.params": {"learning_rate": 0.01, "beta_1": 0.9, "beta_2": 0.98, "epsilon": 1e-9},
                            "lr_schedule.class": "piecewise",
                            "lr_schedule.params": {"schedule_steps": [5], "schedule_lrs": [0.01, 0.01]},
                            "validator.class": "SeqGenerationValidator",
                            "validator.params": {"eval_steps": 20, "eval_dataset.class": "ParallelTextDataset",
                                                 "eval_dataset.params": {"src_file": str(tmp_path / "dev.src"),
                                                                         "trg_file": str(tmp_path / "dev.trg"), "data_is_processed": True},
                                                 "eval_metric.class": "tok_bleu", "eval_search_method.class": "beam_search",
                                                 "eval_search_method.params": {"beam_size": 2, "maximum_decode_length": 8},
                                                 "eval_top_checkpoints_to_keep": 2}}}
    (tmp_path / "cfg.yml").write_text(yaml.dump(cfg))
    model_dir = str(tmp_path / "out")

    def launch(extra):
        argv = ["--config_paths", str(tmp_path / "cfg.yml"), "--hparams_set", "transformer_toy", "--model_dir", model_dir,
                "--dtype", "float32", "--distribution_strategy", "none"] + extra
        parser = flags_core.define_flags(R.FLAG_LIST, argv=argv)
        args, _ = flags_core.intelligent_parse_flags(R.FLAG_LIST, parser, R._pre_load_args, argv=argv)
        task = build_task(args)
        ds = build_dataset(args)
        model = task.build_model(args, device="cpu", dtype="float32", seed=args["seed"])   # run_experiment() with device=cpu
        for k in list(model.args):
            assert not k.endswith("dropout_rate") or model.args[k] == 0.1
        entry = build_exp(args, strategy="none", model=model, task=task, model_dir=args["model_dir"], custom_dataset=ds)
        return entry, entry.run()
    trainer, loss = launch([])
    assert float(loss) < 2.2                                   # from ~2.4 (ln 11 + smoothing) on the copy task
    assert latest_checkpoint(model_dir).endswith("ckpt-40") and os.path.exists(os.path.join(model_dir, "model_configs.yml"))
    v = trainer._validator
    assert [s for s, _ in v.gen_history] == [20, 40] and [s for s, _ in v.history] == [20, 40]
    assert latest_checkpoint(model_dir + "_best") is not None and latest_checkpoint(model_dir + "_best_avg") is not None
    # resume: the second invocation starts at step 41 (optimizer state restored) and stops at train_steps
    trainer2, _ = launch(["--train_steps", "45"])
    assert latest_checkpoint(model_dir).endswith("ckpt-40") and trainer2.model.rt.step == 45   # dropout-mask step count continues across the restart
    # predict entry with a metric, from the same model_dir
    out = tmp_path / "hyp.txt"
    gen, hyps = launch(["--entry", "predict", "--dataset.params", yaml.dump(cfg["entry.params"]["validator.params"]["eval_dataset.params"]),
                        "--output_file", str(out), "--metric", "tok_bleu", "--search_method.params", "{'beam_size': 2, 'maximum_decode_length': 8}"])
    assert len(hyps) == 6 and out.read_text().strip().splitlines() == hyps
    assert gen.metric_result is not None and 0.0 <= gen.metric_result["tok_bleu"] <= 100.0


def test_top_sampling_with_k1_equals_greedy_beam_search_on_the_model(cpu_kernels):
    """The registered `TopSampling` search drives the model's incremental decoder like the beam search does: with top_k = 1
    its hypotheses are the greedy ones (beam_size 1, no length penalty effect on a single beam)."""
    from neurst_amd.layers.search import build_search_layer
    model, cfg, shape = _speech_model("toy")
    inputs = {k: v for k, v in _speech_inputs(shape).items() if k.startswith("src")}
    greedy = build_search_layer({"search_method.class": "beam_search",
                                 "search_method.params": {"beam_size": 1, "maximum_decode_length": 6, "extra_decode_length": 2}})
    samp = build_search_layer({"search_method.class": "TopSampling",
                               "search_method.params": {"top_k": 1, "maximum_decode_length": 6, "extra_decode_length": 2, "seed": 7}})
    h_greedy, _ = greedy(model, inputs)
    h_samp, _ = samp(model, inputs)
    eos = shape[4] - 1
    for a, b in zip(h_greedy.tolist(), h_samp.tolist()):     # compare up to and including the first EOS
        cut = a.index(eos) + 1 if eos in a else len(a)
        assert a[:cut] == b[:cut]
    three = build_search_layer({"search_method.class": "TopSampling",
                                "search_method.params": {"sample_num": 3, "top_p": 0.8, "maximum_decode_length": 6, "seed": 1}})
    h3, _ = three(model, inputs)
    assert h3.shape == (shape[0] * 3, 6) and three.top_k == 3


def test_model_ensemble_beam_search(cpu_kernels):
    """BeamSearch over a list of models (the reference's ensemble decoding): two copies of one model decode exactly like the
    single model; two different models give a valid, different search (scores are those of the averaged distribution)."""
    from neurst_amd.layers.search import build_search_layer
    model, cfg, shape = _speech_model("toy")
    other, _, _ = _speech_model("toy")
    other.store.master.mul_(1.3)
    inputs = {k: v for k, v in _speech_inputs(shape).items() if k.startswith("src")}
    search = build_search_layer({"search_method.class": "beam_search",
                                 "search_method.params": {"beam_size": 3, "top_k": 2, "maximum_decode_length": 6, "extra_decode_length": 2}})
    h1, s1 = search(model, inputs)
    h2, s2 = search([model, model], inputs)
    assert torch.equal(h1, h2) and torch.allclose(s1, s2, rtol=1e-5, atol=1e-5)
    h3, s3 = search([model, other], inputs, ensemble_weights=[0.5, 0.5])
    assert h3.shape == h1.shape and bool(torch.isfinite(s3).all()) and not torch.allclose(s3, s1)


def test_cli_tfrecord_flow_train_resume_evaluate_predict_on_cpu(cpu_kernels, tmp_path):
    """The SAME body as tests/test_gpu_model.py::test_cli_training_from_tfrecord_shards, over the emulated kernels: trainer from
    AudioTFRecordDataset shards with PROJECTED transcripts, CriterionValidator, resume, `evaluation`, then `predict` (the entry
    that must not touch `dataset.targets` when the records hold ids: exps/sequence_generator.py:163-175)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cli_flows import cli_tfrecord_flow
    report = {}
    cli_tfrecord_flow(tmp_path, report, device="cpu")
    assert report["cli_tfrecord.last_loss"] < report["cli_tfrecord.first_loss"]


def test_predict_entry_with_metric_on_projected_audio_records_skips_the_references(cpu_kernels, tmp_path):
    """`--metric` configured + a dataset without text references: hypotheses are still written, no score, no assertion."""
    from neurst_amd.data import tfrecord
    from neurst_amd.data.datasets import build_dataset
    from neurst_amd.exps import build_exp
    from neurst_amd.tasks import build_task
    rng = np.random.RandomState(1)
    V, fdim = 23, 16
    recs = [tfrecord.encode_example({"audio": rng.randn(30 * fdim).astype(np.float32),
                                     "translation": np.array([1, 2, V - 1], np.int64)}) for _ in range(5)]
    tfrecord.write_records(str(tmp_path / "dev.tfrecords-00000-of-00001"), recs)
    args = {"task.class": "SpeechToText", "task.params": {"audio_feature_dim": fdim, "vocab_size": V},
            "dataset.class": "AudioTFRecordDataset",
            "dataset.params": {"data_path": str(tmp_path / "dev.tfrecords"), "feature_key": "audio", "transcript_key": "translation"},
            "entry.class": "predict", "hparams_set": "speech_transformer_toy"}
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    hp = get_hyper_parameters("speech_transformer_toy")
    args.update({k: hp[k] for k in ("model.class", "model.params")})
    task = build_task(args)
    ds = build_dataset(args)
    assert ds._transcript_is_projected
    model = task.build_model(args, device="cpu", dtype="float32", seed=3)
    entry = build_exp({"entry.class": "predict", "entry.params": {"metric.class": "tok_bleu", "batch_size": 4,
                                                                  "search_method.class": "beam_search",
                                                                  "search_method.params": {"beam_size": 2, "maximum_decode_length": 5}}},
                      strategy="none", model=model, task=task, model_dir=None, custom_dataset=ds)
    hyps = entry.run()
    assert len(hyps) == 5 and entry.metric_result is None


def test_fused_feed_forward_host_wiring_matches_the_two_gemm_schedule(cpu_kernels, monkeypatch):
    """TransformerFFN in bf16 at d_model 256 takes the one-launch path (nst_ffn_fwd / nst_ffn_bwd over transposed weight
    copies); over the emulated kernels it must give what the two-GEMM schedule gives -- same masks (both dropouts on),
    same gradients in the flat buffer -- for the pre-norm and the post-norm wrapper."""
    from neurst_amd.layers.common_layers import PrePostProcessingWrapper, TransformerFFN
    from neurst_amd.runtime import Runtime
    for pre_norm in (True, False):
        outs = []
        for fused in ("1", "0"):
            monkeypatch.setattr("neurst_amd.layers.common_layers._FFN_FUSED", fused == "1")
            monkeypatch.setattr("neurst_amd.layers.common_layers._FFN_FUSED_MIN_ROWS", 1)
            monkeypatch.setattr("neurst_amd.layers.common_layers._FFN_FUSED_BWD", True)
            rt = Runtime(device="cpu", dtype="bfloat16", seed=3)
            w = PrePostProcessingWrapper(rt, "w", TransformerFFN(rt, "w/ffn", 256, 384, 0.2, torch.Generator().manual_seed(0)),
                                         256, 0.1, 1e-6, pre_norm=pre_norm)
            assert w.layer.fused == (fused == "1")
            rt.store.finalize(rt.device, rt.dtype)
            if fused == "1":
                assert torch.equal(w.layer._w1t.t, w.layer.dense1.kernel.compute.t())
            g = torch.Generator().manual_seed(1)
            x = torch.randn(70, 256, generator=g).to(torch.bfloat16)
            y = w.forward(x, True)
            rt.store.begin_backward()
            dx = w.backward(torch.randn(70, 256, generator=g).to(torch.bfloat16))
            outs.append((y.double(), dx.double(), rt.store.grad.clone().double()))
        (y1, dx1, g1), (y0, dx0, g0) = outs
        assert float((y1 - y0).abs().max()) <= 2e-2 * float(y0.abs().max())
        assert float((dx1 - dx0).abs().max()) <= 2e-2 * float(dx0.abs().max())
        assert float((g1 - g0).norm() / g0.norm()) <= 1e-2


def test_whole_row_products_host_wiring_matches_the_unfused_pairs(cpu_kernels, monkeypatch):
    """bf16, d_model 256: the last product of a sub-layer waits for the next LayerNorm (DeferredDelta ->
    nst_gemm_add_layernorm_fwd) and the first input-gradient product of a sub-layer's backward carries that LayerNorm's backward
    (LnBackward -> nst_gemm_layernorm_bwd).  Over the emulated kernels -- where a fused entry IS the composition of the pair it
    replaces -- logits, loss and the flat gradient buffer must equal the unfused schedule's bit for bit, with every dropout on;
    and the fused entries must have taken every call they can take (all but the first sub-layer of each stack, whose stream is
    still the bf16 embedding output)."""
    from neurst_amd import kernels as K
    from neurst_amd.criterions import build_criterion
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    outs, calls = [], {}
    for fused in (True, False):
        monkeypatch.setattr("neurst_amd.layers.common_layers._ROW_FUSION", fused)
        n = {"fwd": 0, "bwd": 0}
        real_f, real_b = K.gemm_add_layernorm_fwd, K.gemm_layernorm_bwd
        monkeypatch.setattr(K, "gemm_add_layernorm_fwd", lambda *a, _r=real_f, _n=n, **k: (_n.__setitem__("fwd", _n["fwd"] + 1), _r(*a, **k))[1])
        monkeypatch.setattr(K, "gemm_layernorm_bwd", lambda *a, _r=real_b, _n=n, **k: (_n.__setitem__("bwd", _n["bwd"] + 1), _r(*a, **k))[1])
        model, cfg, shape = _speech_model("row", dropout=0.1, dtype="bfloat16")
        inputs = _speech_inputs(shape)
        logits = model(inputs, is_training=True)
        loss = crit.reduce_loss(inputs, logits)
        model.backward(crit.backward())
        outs.append((logits.double().clone(), float(loss), model.store.grad.clone()))
        calls[fused] = dict(n)
        monkeypatch.setattr(K, "gemm_add_layernorm_fwd", real_f)
        monkeypatch.setattr(K, "gemm_layernorm_bwd", real_b)
        # evaluation forward (no dropout, nothing saved) goes the same way
        ev = model(inputs, is_training=False)
        outs[-1] += (ev.double().clone(),)
    # 2 encoder layers x 2 sub-layers + 2 decoder layers x 3 sub-layers, minus the first sub-layer of each stack
    assert calls[True] == {"fwd": 8, "bwd": 8} and calls[False] == {"fwd": 0, "bwd": 0}, calls
    (l1, s1, g1, e1), (l0, s0, g0, e0) = outs
    assert torch.equal(l1, l0) and s1 == s0 and torch.equal(e1, e0)
    assert torch.equal(g1, g0)


def test_feed_forward_pair_with_row_stages_host_wiring(cpu_kernels, monkeypatch):
    """The one-launch feed-forward pair carrying the wrapper's row stages (DeferredFfn -> nst_ffn_add_layernorm_fwd,
    LnBackward.run_ffn -> nst_ffn_layernorm_bwd) against the unfused schedule over the emulated kernels: the next LayerNorm's
    output, the float32 stream, the gradient handed upstream (and its dropped copy) and the flat gradient buffer are identical;
    the fused entries were taken exactly once each."""
    from neurst_amd import kernels as K
    from neurst_amd.layers.common_layers import LayerNorm, PrePostProcessingWrapper, ResidualStream, TransformerFFN
    from neurst_amd.runtime import Runtime
    monkeypatch.setattr("neurst_amd.layers.common_layers._FFN_FUSED_MIN_ROWS", 1)
    monkeypatch.setattr(K, "ffn_ln_supported", lambda rows, d, f: True)
    outs, calls = [], {}
    for fused in (True, False):
        monkeypatch.setattr("neurst_amd.layers.common_layers._ROW_FUSION", fused)
        n = {"fwd": 0, "bwd": 0}
        real_f, real_b = K.ffn_add_layernorm_fwd, K.ffn_layernorm_bwd
        monkeypatch.setattr(K, "ffn_add_layernorm_fwd", lambda *a, _r=real_f, _n=n, **k: (_n.__setitem__("fwd", _n["fwd"] + 1), _r(*a, **k))[1])
        monkeypatch.setattr(K, "ffn_layernorm_bwd", lambda *a, _r=real_b, _n=n, **k: (_n.__setitem__("bwd", _n["bwd"] + 1), _r(*a, **k))[1])
        rt = Runtime(device="cpu", dtype="bfloat16", seed=3)
        prev = PrePostProcessingWrapper(rt, "p", TransformerFFN(rt, "p/ffn", 256, 128, 0.0, torch.Generator().manual_seed(2)), 256, 0.1, 1e-6)
        w = PrePostProcessingWrapper(rt, "w", TransformerFFN(rt, "w/ffn", 256, 384, 0.2, torch.Generator().manual_seed(0)), 256, 0.1, 1e-6)
        nxt = LayerNorm(rt, "n", 256, 1e-6)
        rt.store.finalize(rt.device, rt.dtype)
        prev._p = 0.1                                           # (the upstream wrapper only lends its dropout site to the backward)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(70, 256, generator=g)                   # the float32 stream in front of the wrapper
        delta0 = torch.randn(70, 256, generator=g).to(torch.bfloat16)
        stream = w.forward(ResidualStream(x, delta0), True)     # the wrapper's own LayerNorm adds delta0 (add_layernorm_fwd)
        y, xs = nxt.forward_stream(stream, save=True)
        rt.store.begin_backward()
        dy = torch.randn(70, 256, generator=g).to(torch.bfloat16)
        d_mid = nxt.backward(dy, consumer=w)
        d_in = w.backward(d_mid, consumer=prev)
        outs.append((y.double(), xs.double(), d_in.double(), d_in._nst_dropped[1].double(), rt.store.grad.clone()))
        calls[fused] = dict(n)
        monkeypatch.setattr(K, "ffn_add_layernorm_fwd", real_f)
        monkeypatch.setattr(K, "ffn_layernorm_bwd", real_b)
    assert calls[True] == {"fwd": 1, "bwd": 1} and calls[False] == {"fwd": 0, "bwd": 0}, calls
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_dynamic_loss_scale_skips_overflow_steps_and_follows_the_reference_schedule(cpu_kernels):
    """RevisedDynamicLossScale (neurst/training/revised_dynamic_loss_scale.py:48-107) around the train step: gradients carry the
    scale, a finite step applies them unscaled (same weights as the unscaled step), every `growth_steps` good steps the scale
    doubles, an overflow halves it (floor 1), clears the counter and leaves weights and moments untouched."""
    from neurst_amd.criterions import build_criterion
    from neurst_amd.optimizers.adam import Adam
    from neurst_amd.training.train_step import TrainStep
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    model, cfg, shape = _speech_model("toy")
    ref_model, _, _ = _speech_model("toy")
    opt, ref_opt = Adam(model.store, learning_rate=1e-2), Adam(ref_model.store, learning_rate=1e-2)
    step = TrainStep(model, crit, opt, loss_scale={"initial_loss_scale": 1024.0, "growth_steps": 2, "multiplier": 2.0})
    ref_step = TrainStep(ref_model, build_criterion({"criterion.class": "label_smoothed_cross_entropy",
                                                     "criterion.params": {"label_smoothing": 0.1}}), ref_opt)
    scales = []
    for i in range(3):
        b = _speech_inputs(shape, 50 + i)
        step(b), ref_step(b)
        scales.append(float(step._ls_state[0]))
    assert scales == [1024.0, 2048.0, 2048.0]                       # good steps 1, 2 (-> x2, counter 0), 1
    assert rel_err(model.store.master, ref_model.store.master) < 1e-4   # scaled-then-unscaled == unscaled
    # an overflow: poison one gradient element through a hook that runs after the backward pass
    before = model.store.master.clone()
    m_before = opt.m.clone()
    orig = model.backward

    def poisoned(dlogits, accumulate=False):
        orig(dlogits, accumulate=accumulate)
        model.store.grad[5] = float("inf")
    model.backward = poisoned
    step(_speech_inputs(shape, 99))
    model.backward = orig
    assert float(step._ls_state[0]) == 1024.0 and float(step._ls_state[1]) == 0.0 and float(step._ls_state[2]) == 0.0
    assert torch.equal(model.store.master, before) and torch.equal(opt.m, m_before)      # the step was skipped
    step(_speech_inputs(shape, 100))
    assert float(step._ls_state[2]) == 1.0 and not torch.equal(model.store.master, before)


@pytest.mark.parametrize("clip", [{"clip_value": 2e-3}, {"clip_norm": 5e-2}])
def test_dynamic_loss_scale_together_with_clipping(cpu_kernels, clip):
    """gradaccum_keras_model.py:224-233 with a LossScaleOptimizer: aggregate -> get_unscaled_gradients -> clip -> apply.  The
    scaled-and-clipped step must land on the weights of the unscaled clipped step (the clip acts on UNSCALED gradients: with
    a scale of 1024 a clip applied first would cut everything), and an overflow step is still skipped."""
    from neurst_amd.criterions import build_criterion
    from neurst_amd.optimizers.adam import Adam
    from neurst_amd.training.train_step import TrainStep
    mk = lambda: build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    model, cfg, shape = _speech_model("toy")
    ref_model, _, _ = _speech_model("toy")
    free_model, _, _ = _speech_model("toy")
    opt, ref_opt, free_opt = (Adam(m.store, learning_rate=1e-2) for m in (model, ref_model, free_model))
    step = TrainStep(model, mk(), opt, loss_scale={"initial_loss_scale": 1024.0, "growth_steps": 2, "multiplier": 2.0}, **clip)
    ref_step = TrainStep(ref_model, mk(), ref_opt, **clip)
    free_step = TrainStep(free_model, mk(), free_opt)
    for i in range(3):
        b = _speech_inputs(shape, 70 + i)
        step(b), ref_step(b), free_step(b)
    assert [float(step._ls_state[0]), float(step._ls_state[2])] == [2048.0, 1.0]
    assert rel_err(model.store.master, ref_model.store.master) < 1e-4
    # the clip bites at these thresholds (otherwise the comparison above would not tell the order of the two apart)
    assert rel_err(free_model.store.master, ref_model.store.master) > 1e-3
    before = model.store.master.clone()
    orig = model.backward

    def poisoned(dlogits, accumulate=False):
        orig(dlogits, accumulate=accumulate)
        model.store.grad[7] = float("nan")
    model.backward = poisoned
    step(_speech_inputs(shape, 99))
    model.backward = orig
    assert float(step._ls_state[2]) == 0.0 and torch.equal(model.store.master, before)


def test_decoder_group_goes_to_the_weight_gradient_stream_only_when_its_products_are_short(monkeypatch):
    """encoder_decoder_model._decoder_group_on_side: the rule is rows(decoder) <= rows(encoder) / 2 (speech: 75 target positions
    against 225 encoder frames per utterance -> yes; text models with equal lengths -> no); tests may pin it."""
    from neurst_amd.models import encoder_decoder_model as M
    monkeypatch.setattr(M, "_WGRAD_DECODER_SIDE", None)
    assert M._decoder_group_on_side(torch.empty(128, 75, 256), torch.empty(128, 225, 256))
    assert M._decoder_group_on_side(torch.empty(4, 10, 8), torch.empty(4, 20, 8))
    assert not M._decoder_group_on_side(torch.empty(256, 64, 512), torch.empty(256, 64, 512))
    assert not M._decoder_group_on_side(torch.empty(4, 11, 8), torch.empty(4, 20, 8))
    monkeypatch.setattr(M, "_WGRAD_DECODER_SIDE", True)
    assert M._decoder_group_on_side(torch.empty(256, 64, 512), torch.empty(256, 64, 512))
