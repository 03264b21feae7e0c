"""Parity of the HIP path at layer / model / training-step level against the CPU oracle and against the
reference's golden vectors (tests/golden/*.npz).  Everything goes through libneurst_hip.so.
Tolerances: 1e-3 fp32, 1e-2 bf16 (north star), relative to the magnitude of the reference tensor."""
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import neurst_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = {"float32": 1e-3, "bfloat16": 1e-2}
REPORT = {}


@pytest.fixture(scope="module", autouse=True)
def _dump_report():
    yield
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "model_report.json")
    merged = {}
    if os.path.exists(path):
        try:
            merged = json.load(open(path))
        except Exception:
            merged = {}
    merged.update(REPORT)
    with open(path, "w") as fp:
        json.dump(merged, fp, indent=1, sort_keys=True)


def rel_err(got, ref):
    got, ref = got.detach().float().cpu().double(), ref.detach().double()
    assert got.shape == ref.shape, f"{tuple(got.shape)} vs {tuple(ref.shape)}"
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)


def check(name, got, ref, tol):
    e = rel_err(got, ref)
    REPORT[name] = e
    assert math.isfinite(e) and e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"


def make_runtime(dtype="float32"):
    from neurst_amd.runtime import Runtime
    return Runtime(device=DEV, dtype=dtype, seed=7)


def load_weights(store, W):
    sd = {k: v for k, v in W.items() if k in store.params}
    store.load_state_dict(sd, strict=False)


# ------------------------------------------------------------------------------------------------ golden vectors via HIP
def test_golden_mha_cross_hip():
    from neurst_amd.layers.attentions.multi_head_attention import MultiHeadAttention
    r, W = load_golden("mha_cross")
    rt = make_runtime()
    att = MultiHeadAttention(rt, "a", 2, 4, 0.0, torch.Generator().manual_seed(0), output_depth=3, input_depth=1,
                             memory_depth=1)
    rt.store.finalize(DEV, torch.float32)
    load_weights(rt.store, {"a/" + k: v for k, v in W.items()})
    q, m = torch.from_numpy(r["query"]).to(DEV), torch.from_numpy(r["memory"]).to(DEV)
    out = att.forward(q.reshape(2, 1), m.reshape(2, 1), 1, 2, 2, is_training=False)
    ssd = float(((out.cpu().double().reshape(1, 2, 3) - torch.from_numpy(r["expected"]).double()) ** 2).sum())
    REPORT["golden.mha_cross.ssd"] = ssd
    assert ssd < 1e-9


def test_golden_mha_self_hip():
    from neurst_amd.layers.attentions.multi_head_attention import MultiHeadSelfAttention
    r, W = load_golden("mha_self")
    rt = make_runtime()
    att = MultiHeadSelfAttention(rt, "a", 2, 4, 0.0, torch.Generator().manual_seed(0), output_depth=3, input_depth=2)
    rt.store.finalize(DEV, torch.float32)
    load_weights(rt.store, {"a/" + k: v for k, v in W.items()})
    q = torch.from_numpy(r["query"]).to(DEV)
    out = att.forward(q.reshape(2, 2), 1, 2, bias=torch.from_numpy(r["bias"]).to(DEV), is_training=False)
    ssd = float(((out.cpu().double().reshape(1, 2, 3) - torch.from_numpy(r["expected"]).double()) ** 2).sum())
    REPORT["golden.mha_self.ssd"] = ssd
    assert ssd < 1e-9


def test_golden_encoder_hip():
    from neurst_amd.layers.encoders import build_encoder
    r, W = load_golden("transformer_encoder")
    rt = make_runtime()
    enc = build_encoder({"encoder.class": "TransformerEncoder",
                         "encoder.params": dict(num_layers=1, hidden_size=4, num_attention_heads=2, filter_size=16,
                                                attention_dropout_rate=0.1, ffn_dropout_rate=0.1,
                                                layer_postprocess_dropout_rate=0.1)}).build(rt, torch.Generator().manual_seed(0))
    rt.store.finalize(DEV, torch.float32)
    W = O.fill_default_biases(W)
    load_weights(rt.store, W)
    out = enc(torch.from_numpy(r["inputs"]).to(DEV), torch.from_numpy(r["input_padding"]).to(DEV), is_training=False)
    ssd = float(((out.cpu().double() - torch.from_numpy(r["expected"]).double()) ** 2).sum())
    REPORT["golden.encoder.ssd"] = ssd
    assert ssd < 1e-9


def test_golden_decoder_hip():
    from neurst_amd.layers.decoders import build_decoder
    r, W = load_golden("transformer_decoder")
    rt = make_runtime()
    dec = build_decoder({"decoder.class": "TransformerDecoder",
                         "decoder.params": dict(num_layers=1, hidden_size=4, num_attention_heads=2, filter_size=16,
                                                attention_dropout_rate=0.1, ffn_dropout_rate=0.1,
                                                layer_postprocess_dropout_rate=0.1)}).build(rt, torch.Generator().manual_seed(0))
    rt.store.finalize(DEV, torch.float32)
    load_weights(rt.store, O.fill_default_biases(W))
    cache = dec.create_decoding_internal_cache(torch.from_numpy(r["encoder_outputs"]).to(DEV),
                                               torch.from_numpy(r["encoder_inputs_padding"]).to(DEV), is_inference=False)
    out = dec(torch.from_numpy(r["decoder_inputs"]).to(DEV), cache, is_training=False)
    ssd = float(((out.cpu().double() - torch.from_numpy(r["expected"]).double()) ** 2).sum())
    REPORT["golden.decoder.ssd"] = ssd
    assert ssd < 1e-9


def test_golden_full_transformer_logits_hip():
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    r, W = load_golden("transformer_toy_logits")
    hp = get_hyper_parameters("transformer_toy")
    model = build_model(hp, dict(vocab_size=8, eos_id=7, bos_id=6, unk_id=5), dict(vocab_size=5, eos_id=4, bos_id=3, unk_id=2),
                        device=DEV, dtype="float32")
    zero = {n: torch.zeros(p.shape) for n, p in model.store.params.items() if n.endswith("/bias")}
    model.store.load_state_dict(zero, strict=False)   # golden test pins kernels only; biases are zero
    load_weights(model.store, W)
    inputs = {"src": torch.from_numpy(r["src"]).to(DEV), "src_padding": torch.from_numpy(r["src_padding"]).to(DEV),
              "trg_input": torch.from_numpy(r["trg_input"]).to(DEV)}
    logits = model(inputs, is_training=False)
    ssd = float(((logits.cpu().double() - torch.from_numpy(r["expected"]).double()) ** 2).sum())
    REPORT["golden.transformer_toy_logits.ssd"] = ssd
    assert ssd < 1e-9


@pytest.mark.parametrize("tag", ["frontend_ln", "frontend_noln", "frontend_ragged"])
def test_frontend_matches_reference_neurst_pt_hip(tag):
    from neurst_amd.layers.modalities.audio_modalities import AudioConv2dSubsamplingLayer
    r, W = load_golden("neurst_pt_" + tag)
    rt = make_runtime()
    C = W["input_audio_modality/conv1/kernel"].shape[-1]
    d = W["input_audio_modality/output_dense/kernel"].shape[-1]
    layer = AudioConv2dSubsamplingLayer(rt, "input_audio_modality", d, r["src"].shape[2], torch.Generator().manual_seed(0),
                                        channels=C, layer_norm=bool(int(r["layer_norm"])))
    rt.store.finalize(DEV, torch.float32)
    load_weights(rt.store, W)
    out = layer.forward(torch.from_numpy(r["src"]).to(DEV), is_training=False)
    np.testing.assert_allclose(out.cpu().numpy(), r["expected"], atol=5e-5, rtol=0)


# ------------------------------------------------------------------------------------------------ model fwd/bwd vs oracle
def _speech_case(name, dtype, device=None, batch=None, ragged_batch=None, **extra):
    cases = {
        # d, H, enc, dec, ffn, C, B, T, F, L, V, ragged
        "toy": (8, 2, 2, 2, 10, 5, 2, 11, 80, 3, 5, False),
        "small": (64, 2, 2, 2, 128, 32, 3, 70, 16, 9, 50, True),
        "mid": (256, 4, 2, 1, 512, 64, 2, 120, 80, 12, 300, True),
        # the benchmark architecture itself (speech_transformer_s, neurst/models/speech_transformer.py:209-216) at its real
        # sequence shape: 12 + 6 layers, C = 256 (the conv2 patch kernels), T = 900 ragged, L = 75, V = 8008
        "s_real": (256, 4, 12, 6, 2048, 256, 3, 900, 80, 75, 8008, True),
    }
    d, H, ne, nd, ffn, C, B, T, F, L, V, ragged = cases[name]
    if batch is not None:
        B, ragged = batch, False
    if ragged_batch is not None:      # the case's architecture at another batch size, lengths ragged as in the case itself
        B, ragged = ragged_batch, True
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    hp = get_hyper_parameters("speech_transformer_toy")
    p = dict(hp["model.params"])
    p.update({"modality.dim": d, "modality.source.channels": C, "encoder.num_layers": ne, "decoder.num_layers": nd,
              "encoder.hidden_size": d, "decoder.hidden_size": d, "encoder.num_attention_heads": H,
              "decoder.num_attention_heads": H, "encoder.filter_size": ffn, "decoder.filter_size": ffn})
    for k in list(p):
        if k.endswith("dropout_rate"):
            p[k] = 0.0
    p.update(extra)
    model = build_model({"model.class": "SpeechTransformer", "model.params": p},
                        {"audio_feature_dim": F, "audio_feature_channels": 1},
                        {"vocab_size": V, "eos_id": V - 1, "bos_id": V - 2, "unk_id": V - 3}, device=device or DEV, dtype=dtype,
                        init_seed=3)
    g = torch.Generator().manual_seed(11)
    # non-trivial biases / LN affine so every gradient path is exercised
    sd = {}
    for n, prm in model.store.params.items():
        if n.endswith("/bias") or n.endswith("/beta"):
            sd[n] = torch.randn(prm.shape, generator=g) * 0.05
        elif n.endswith("/gamma"):
            sd[n] = 1.0 + torch.randn(prm.shape, generator=g) * 0.1
    model.store.load_state_dict(sd, strict=False)
    src = torch.randn(B, T, F, 1, generator=g)
    if ragged:
        src_len = torch.tensor([T - (i * T) // (2 * B) for i in range(B)])
        trg_len = torch.tensor([L - i for i in range(B)]).clamp(min=1)
    else:
        src_len, trg_len = torch.full((B,), T), torch.full((B,), L)
    trg = torch.randint(0, V - 3, (B, L), generator=g)
    trg = torch.where(torch.arange(L)[None] >= (trg_len[:, None] - 1), torch.full_like(trg, V - 1), trg)
    trg_input = torch.cat([torch.full((B, 1), V - 2), trg[:, :-1]], 1)
    inputs = {"src": src, "src_length": src_len, "trg": trg, "trg_input": trg_input, "trg_length": trg_len}
    cfg = {"num_enc": ne, "num_dec": nd, "num_heads": H, "layer_norm": True}
    return model, inputs, cfg


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("case", ["toy", "small", "mid"])
def test_speech_transformer_forward_backward(case, dtype):
    from neurst_amd.criterions import build_criterion
    model, inputs, cfg = _speech_case(case, dtype)
    W = {n: p.data.detach().cpu().clone() for n, p in model.store.params.items()}
    if dtype == "bfloat16":  # the oracle sees the same (bf16-rounded) GEMM weights the device path uses
        for n, p in model.store.params.items():
            if n.endswith("/kernel") and "conv1" not in n or n.endswith("shared/weights"):
                W[n] = p.compute.detach().float().cpu()
    loss_ref, logits_ref, grads_ref = O.train_step_reference({k: v.double() for k, v in W.items()},
                                                             {k: (v.double() if v.is_floating_point() else v)
                                                              for k, v in inputs.items()}, cfg, 0.1)
    dinp = {k: v.to(DEV) for k, v in inputs.items()}
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(dinp, is_training=True)
    loss = crit.reduce_loss(dinp, logits)
    model.backward(crit.backward())
    tol = TOL[dtype]
    tag = f"st[{case},{dtype}]"
    check(tag + ".logits", logits, logits_ref, tol * (3 if dtype == "bfloat16" else 1))
    REPORT[tag + ".loss_abs_err"] = abs(float(loss) - float(loss_ref))
    assert abs(float(loss) - float(loss_ref)) <= tol * max(1.0, abs(float(loss_ref)))
    # fp32: every gradient tensor within 2e-3 of the oracle (max-abs relative to the tensor's magnitude).
    # bf16: activations are rounded to bf16 between kernels, which flips a few ReLU gates of near-zero
    # pre-activations w.r.t. the fp32 oracle; per-element max error is then dominated by single flips when the
    # batch has few rows, so the bf16 criterion is the relative L2 error, per tensor and over the whole gradient.
    worst, bad = 0.0, []
    num = den = 0.0
    for n, p in model.store.params.items():
        g, r = p.grad.detach().float().cpu().double(), grads_ref[n].double()
        e_max = rel_err(p.grad, grads_ref[n])
        e_l2 = float((g - r).norm() / max(float(r.norm()), 1e-12))
        num += float(((g - r) ** 2).sum())
        den += float((r ** 2).sum())
        REPORT[f"{tag}.grad.{n}"] = e_max if dtype == "float32" else e_l2
        e = e_max if dtype == "float32" else e_l2
        worst = max(worst, e)
        # bf16: 1.5 x the worst per-tensor error measured on these small cases (8.3e-2, profiles/r01_model_parity_report.json;
        # few rows, so single ReLU-gate flips weigh most here)
        if not (e <= (2e-3 if dtype == "float32" else 0.125)):
            bad.append((n, e))
    glob = math.sqrt(num / max(den, 1e-30))
    REPORT[tag + ".grad_worst"] = worst
    REPORT[tag + ".grad_global_rel_l2"] = glob
    assert not bad, f"{tag}: gradients out of tolerance: {bad[:8]}"
    assert glob <= (1e-3 if dtype == "float32" else 3e-2), f"{tag}: global gradient rel-L2 error {glob:.3e}"


class _DeviceGates(object):
    """ReLU gates of the device path (sign of the saved post-ReLU activations), keyed like oracle.relu_gates expects."""

    def __init__(self, model):
        self.g = {}
        for stack in (model._encoder, model._decoder):
            for layer in stack._stacking_layers:
                ffn = layer._ffn_layer.layer
                self.g[ffn.name] = (ffn._saved[1] > 0).cpu()
        front = getattr(model._src_modality, "embedding_layer", model._src_modality)
        saved = front._saved
        self.g[front.name + "/conv1"] = (saved[1] > 0).cpu()
        self.g[front.name + "/conv2"] = (saved[5] > 0).cpu()

    def gate_for(self, tag, shape):
        g = self.g.get(tag)
        return None if g is None else g.reshape(shape)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_speech_transformer_s_real_configuration_parity(dtype):
    """Logits, loss and EVERY gradient of the real speech_transformer_s (12 + 6 layers, d = 256, ffn = 2048, C = 256, V = 8008)
    on a ragged batch of 900-frame utterances against the fp64 oracle.  bf16 is reported twice: against the oracle as is,
    and against the oracle run under the DEVICE's ReLU gates (oracle.relu_gates) -- the second number is the rounding error
    of the bf16 path proper, the difference between the two is what discrete gate flips of near-zero pre-activations add."""
    from neurst_amd.criterions import build_criterion
    model, inputs, cfg = _speech_case("s_real", dtype)
    W = {n: p.data.detach().cpu().clone() for n, p in model.store.params.items()}
    if dtype == "bfloat16":
        for n, p in model.store.params.items():
            if n.endswith("/kernel") and "conv1" not in n or n.endswith("shared/weights"):
                W[n] = p.compute.detach().float().cpu()
    W64 = {k: v.double() for k, v in W.items()}
    in64 = {k: (v.double() if v.is_floating_point() else v) for k, v in inputs.items()}
    dinp = {k: v.to(DEV) for k, v in inputs.items()}
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(dinp, is_training=True)
    loss = crit.reduce_loss(dinp, logits)
    gates = _DeviceGates(model)
    model.backward(crit.backward())
    tag = f"st[s_real,{dtype}]"

    def compare(suffix, ref):
        loss_ref, logits_ref, grads_ref = ref
        REPORT[f"{tag}{suffix}.logits"] = rel_err(logits, logits_ref)
        REPORT[f"{tag}{suffix}.loss_abs_err"] = abs(float(loss) - float(loss_ref))
        num = den = worst_l2 = worst_max = 0.0
        for n, p in model.store.params.items():
            g, r = p.grad.detach().float().cpu().double(), grads_ref[n].double()
            num += float(((g - r) ** 2).sum())
            den += float((r ** 2).sum())
            e_l2 = float((g - r).norm() / max(float(r.norm()), 1e-12))
            REPORT[f"{tag}{suffix}.grad.{n}"] = e_l2
            worst_l2, worst_max = max(worst_l2, e_l2), max(worst_max, rel_err(p.grad, grads_ref[n]))
        glob = math.sqrt(num / max(den, 1e-30))
        REPORT[f"{tag}{suffix}.grad_global_rel_l2"], REPORT[f"{tag}{suffix}.grad_worst_rel_l2"] = glob, worst_l2
        REPORT[f"{tag}{suffix}.grad_worst_max_abs_rel"] = worst_max
        return REPORT[f"{tag}{suffix}.logits"], REPORT[f"{tag}{suffix}.loss_abs_err"], glob, worst_l2, worst_max

    own = compare("", O.train_step_reference(W64, in64, cfg, 0.1))
    if dtype == "float32":
        assert own[0] <= 1e-3 and own[1] <= 1e-3 and own[4] <= 2e-3, own
        return
    with O.relu_gates(gates):
        gated = compare(".device_gates", O.train_step_reference(W64, in64, cfg, 0.1))
    # north star: 1e-2 (bf16) on loss, logits and gradients
    assert own[0] <= 3e-2 and own[1] <= 1e-2 * max(1.0, abs(float(loss))), own
    assert own[2] <= 3e-2, f"global gradient rel-L2 vs the oracle {own[2]:.3e}"
    # with the discrete gate flips taken out, the bf16 path meets the north-star tolerance on the whole gradient
    assert gated[2] <= 1e-2 and gated[0] <= 1e-2, (gated, own)


@pytest.mark.parametrize("batch", [32, 128])
def test_speech_transformer_s_bf16_step_against_the_oracle_fixture(batch):
    """The bf16 HIP step of the real speech_transformer_s at a batch of 32 / 128 (the benchmark's) ragged 900-frame utterances
    against the float64 ORACLE -- not against another HIP path.  The oracle at these batches runs on the build host
    (tests/golden/make_oracle_b32.py, minutes and tens of GB); what travels is, per gradient tensor and for the logits, the L2
    norm and 64 fixed +-1 projections (oracle/projections.py: the relative L2 distance of two projection sets estimates the
    relative L2 distance of the tensors), for the oracle and for the float64 emulation of the kernels with the same bf16
    rounding points.  Asserted: loss and logits at the north-star bf16 bar of 1e-2; gradients (a) no farther from the oracle
    than 1.3 x what exact kernels with bf16 rounding points are, globally, and within 3 x per tensor, (b) the measured level
    itself.  Round 4 on MI355X (profiles/r04_model_parity_report.json), residual stream rounded to bf16 after every sub-layer:
    global rel-L2 of the gradients against the ORACLE 1.87e-2 at 32 utterances and 1.03e-2 at 128 -- AT the north star's 1e-2, and
    exactly where exact kernels with the same rounding points sit.  Round 5 carries the residual stream in float32
    (layers/common_layers.py: ResidualStream; scripts/rounding_point_study.py found it to be the one rounding class that
    matters): 1.25e-2 at 32 utterances and 8.8e-3 at 128 (emulation 1.20e-2 / 7.6e-3 exact), logits 3.4e-3 -- inside the bar."""
    from neurst_amd.criterions import build_criterion
    from oracle.projections import sign_projections
    fx = np.load(os.path.join(GOLDEN, f"oracle_s_real_b{batch}.npz"))
    assert int(fx["batch"]) == batch
    model, inputs, cfg = _speech_case("s_real", "bfloat16", ragged_batch=batch)
    dinp = {k: v.to(DEV) for k, v in inputs.items()}
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(dinp, is_training=True)
    loss = float(crit.reduce_loss(dinp, logits))
    model.backward(crit.backward())
    torch.cuda.synchronize()
    names = [str(n) for n in fx["names"]]
    assert names == sorted(model.store.params)
    lp = sign_projections(logits.float().cpu(), "logits").numpy()
    gp = np.stack([sign_projections(model.store.params[n].grad.float().cpu(), n).numpy() for n in names])
    gn = np.array([float(model.store.params[n].grad.float().norm()) for n in names])
    ref, emu = fx["grad_proj_ref"], fx["grad_proj_emu"]
    tag = f"st[s_real,bfloat16,B{batch}].oracle_fixture"
    rep = {
        "loss_abs_err": abs(loss - float(fx["loss_ref"])),
        "logits_rel_l2": float(np.linalg.norm(lp - fx["logits_proj_ref"]) / np.linalg.norm(fx["logits_proj_ref"])),
        "logits_rel_l2_emulation": float(np.linalg.norm(fx["logits_proj_emu"] - fx["logits_proj_ref"]) / np.linalg.norm(fx["logits_proj_ref"])),
        "grad_global_rel_l2": float(np.linalg.norm(gp - ref) / np.linalg.norm(ref)),
        "grad_global_rel_l2_emulation": float(np.linalg.norm(emu - ref) / np.linalg.norm(ref)),
        "grad_global_rel_l2_emulation_exact": float(fx["grad_global_rel_l2_emu"]),
        "grad_norm_rel_err_worst": float(np.max(np.abs(gn - fx["grad_norm_ref"]) / np.maximum(fx["grad_norm_ref"], 1e-12))),
    }
    per_hip = np.linalg.norm(gp - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-12)
    per_emu = np.linalg.norm(emu - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-12)
    rep["grad_worst_tensor_rel_l2"], rep["grad_worst_tensor_rel_l2_emulation"] = float(per_hip.max()), float(per_emu.max())
    rep["tensors_above_1e-2"], rep["tensors_above_1e-2_emulation"] = int((per_hip > 1e-2).sum()), int((per_emu > 1e-2).sum())
    rep["ratio_median"] = float(np.median(per_hip / np.maximum(per_emu, 1e-12)))
    for k, v in rep.items():
        REPORT[f"{tag}.{k}"] = v
    assert rep["loss_abs_err"] <= 1e-2 and rep["logits_rel_l2"] <= 1e-2, rep
    assert rep["grad_global_rel_l2"] <= 1.3 * rep["grad_global_rel_l2_emulation"] + 1e-3, rep
    # per tensor: 64 projections estimate a tensor's error to ~ +-12 %, and the device and the emulation are two different
    # realisations of the same rounding noise: 3 x the emulation + a small absolute term (round 5, fp32 residual stream: worst
    # ratio 2.6 at 32 utterances -- the cross-attention q_transform kernel of decoder layer 4, one of the smallest gradients of
    # the model, 2.9e-2 against 1.1e-2 -- and 1.9 at 128; median 0.7 - 0.8)
    bad = [(n, float(h), float(e)) for n, h, e in zip(names, per_hip, per_emu) if h > 3.0 * e + 4e-3]
    assert not bad, f"{len(bad)} gradient tensors farther from the oracle than bf16 rounding explains: {bad[:6]}"
    # ... and the wide band is for the SMALL tensors only: a tensor that carries more than 1 % of the gradient's norm stays
    # within 2 x the emulation (a real regression in a tensor that matters cannot hide behind the q_transform's noise)
    share = fx["grad_norm_ref"] / max(float(np.linalg.norm(fx["grad_norm_ref"])), 1e-30)
    bad2 = [(n, float(h), float(e), float(w)) for n, h, e, w in zip(names, per_hip, per_emu, share) if w > 1e-2 and h > 2.0 * e + 2e-3]
    assert not bad2, f"{len(bad2)} large gradient tensors beyond 2 x the emulation's distance: {bad2[:6]}"
    assert rep["grad_norm_rel_err_worst"] <= 5e-2, rep
    # the north star's bf16 bar of 1e-2 at the benchmark batch: met since the residual stream is carried in float32
    # (round 5: 8.8e-3 measured by projections, 1.05e-2 before; at 32 utterances 1.25e-2, 1.87e-2 before)
    assert rep["grad_global_rel_l2"] <= (1.0e-2 if batch >= 128 else 1.6e-2), rep


def _grad_errors(grads, grads_ref):
    """-> ({name: rel-L2 error}, global rel-L2 error) of a dict of gradients against a reference dict."""
    per, num, den = {}, 0.0, 0.0
    for n, g in grads.items():
        g, r = g.detach().float().cpu().double(), grads_ref[n].detach().double().cpu()
        per[n] = float((g - r).norm() / max(float(r.norm()), 1e-12))
        num += float(((g - r) ** 2).sum())
        den += float((r ** 2).sum())
    return per, math.sqrt(num / max(den, 1e-30))


def s_real_bf16_noise_report(tag):
    """Runs the real speech_transformer_s configuration in bf16 through the HIP path, the float64 oracle, and the float64
    EMULATION of every C-ABI contract (oracle/kernel_emulation.py: the same layer classes, the same bf16 rounding points
    between kernels, exact arithmetic inside them), all on the same weights and batch.  The emulation's distance from the
    oracle is what rounding activations to bf16 at those points costs with perfect kernels -- any quantised path decorrelates
    from the exact one to that level within a few stages (two emulations that differ only in fp32 vs fp64 accumulation end
    2.2e-2 apart on the gradients) -- so the assertable statement about the KERNELS is that the HIP path is no farther from
    the oracle than the emulation is, tensor by tensor.  Returns the report dict (also merged into REPORT)."""
    import pytest as _pytest
    from neurst_amd import kernels as K
    from neurst_amd.criterions import build_criterion
    from oracle import kernel_emulation as E
    model, inputs, cfg = _speech_case("s_real", "bfloat16")
    W = {n: p.data.detach().cpu().clone() for n, p in model.store.params.items()}
    for n, p in model.store.params.items():
        if n.endswith("/kernel") and "conv1" not in n or n.endswith("shared/weights"):
            W[n] = p.compute.detach().float().cpu()
    state = {n: p.data.detach().cpu().clone() for n, p in model.store.params.items()}
    dinp = {k: v.to(DEV) for k, v in inputs.items()}
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(dinp, is_training=True)
    loss = float(crit.reduce_loss(dinp, logits))
    import neurst_amd.layers.common_layers as CL
    ffn0 = model._encoder._stacking_layers[0]._ffn_layer.layer
    rows0 = ffn0._saved[1].shape[0]
    fused = {"rows": rows0, "one_launch_pair": bool(ffn0.fused and rows0 >= CL._FFN_FUSED_MIN_ROWS and CL._FFN_FUSED_BWD),
             }
    model.backward(crit.backward())
    torch.cuda.synchronize()
    logits_hip = logits.float().cpu()
    grads_hip = {n: p.grad.detach().float().cpu().clone() for n, p in model.store.params.items()}
    del model, logits
    loss_ref, logits_ref, grads_ref = O.train_step_reference({k: v.double() for k, v in W.items()},
                                                             {k: (v.double() if v.is_floating_point() else v)
                                                              for k, v in inputs.items()}, cfg, 0.1)
    # the emulation: same classes on CPU tensors with the kernels module swapped (restored below)
    mp = _pytest.MonkeyPatch()
    try:
        E.install(mp)
        emu, _, _ = _speech_case("s_real", "bfloat16", device="cpu")
        emu.store.load_state_dict(state)
        crit2 = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
        lg = emu(inputs, is_training=True)
        loss_emu = float(crit2.reduce_loss(inputs, lg))
        emu.backward(crit2.backward())
        logits_emu = lg.float()
        grads_emu = {n: p.grad.detach().float().clone() for n, p in emu.store.params.items()}
    finally:
        mp.undo()
    e_hip, g_hip = _grad_errors(grads_hip, grads_ref)
    e_emu, g_emu = _grad_errors(grads_emu, grads_ref)
    ratio = {n: e_hip[n] / max(e_emu[n], 1e-12) for n in e_hip}
    worst_n = max(ratio, key=ratio.get)
    rep = {"logits_hip": rel_err(logits_hip, logits_ref), "logits_emulation": rel_err(logits_emu, logits_ref),
           "loss_hip_abs_err": abs(loss - float(loss_ref)), "loss_emulation_abs_err": abs(loss_emu - float(loss_ref)),
           "grad_global_hip": g_hip, "grad_global_emulation": g_emu,
           "grad_worst_hip": max(e_hip.values()), "grad_worst_emulation": max(e_emu.values()),
           "ratio_median": float(np.median(list(ratio.values()))), "ratio_max": ratio[worst_n], "ratio_max_tensor": worst_n,
           "tensors": len(ratio), "tensors_above_1e-2_hip": sum(v > 1e-2 for v in e_hip.values()),
           "tensors_above_1e-2_emulation": sum(v > 1e-2 for v in e_emu.values()), "first_encoder_ffn": fused}
    for k, v in rep.items():
        REPORT[f"{tag}.{k}"] = v
    return rep, e_hip, e_emu


def _assert_bf16_noise_level(rep, e_hip, e_emu):
    # loss / logits: the north-star bf16 bar
    assert rep["logits_hip"] <= 1e-2 and rep["loss_hip_abs_err"] <= 1e-2, rep
    # gradients: tensor by tensor no farther from the oracle than exact kernels with the same rounding points.  Device and
    # emulation are two realisations of the same rounding noise; measured on this case: ratio 0.74 .. 1.65, median 0.93 in round
    # 2; with the fp32 residual stream of round 5 (a lower noise floor under the same spread) 0.5 .. 2.95, median 0.97, the
    # maximum on the LayerNorm beta / q_transform bias of the last decoder layer's cross attention -- the smallest gradients of
    # the model -- while the global figures stay equal (1.75e-2 device, 1.82e-2 emulation).  Hence 3.5 x per tensor with the
    # median and the global ratio pinned below; the small absolute term covers tensors whose emulation error happens to be tiny
    bad = {n: (e_hip[n], e_emu[n]) for n in e_hip if e_hip[n] > 3.5 * e_emu[n] + 2e-3}
    assert not bad, f"{len(bad)} gradient tensors farther from the oracle than bf16 rounding explains: {list(bad.items())[:6]}"
    assert 0.6 <= rep["ratio_median"] <= 1.3, rep
    assert rep["grad_global_hip"] <= 1.3 * rep["grad_global_emulation"] + 1e-3, rep
    # 1.5 x the worst per-tensor error measured on this case (6.5e-2, profiles/r02_model_parity_report.json)
    assert rep["grad_worst_hip"] <= 0.1, rep


def test_speech_transformer_s_bf16_gradient_error_is_rounding_noise():
    """bf16 gradients of the real configuration: per tensor within 3.5x of the float64 emulation's distance from the oracle,
    median ratio 0.6 - 1.3, global figure within 1.3x (see _assert_bf16_noise_level)."""
    rep, e_hip, e_emu = s_real_bf16_noise_report("st[s_real,bfloat16].noise")
    assert rep["first_encoder_ffn"]["one_launch_pair"] is False      # 675 rows: the two-GEMM feed-forward
    _assert_bf16_noise_level(rep, e_hip, e_emu)


def test_speech_transformer_s_parity_with_the_benchmark_kernel_selection():
    """The same comparison with the kernel selection bench.py runs with forced at the small batch: the one-launch feed-forward
    pair (engaged from 16 384 rows on, i.e. never at B = 3) with 128-row workgroups, in forward and backward.  The switches are
    read once per process, so the case runs in a child interpreter."""
    import subprocess
    import sys
    env = dict(os.environ, NST_FFN_MIN_ROWS="1")
    code = ("import json, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_model as T\n"
            "rep, e_hip, e_emu = T.s_real_bf16_noise_report('child')\n"
            "print('REPORT ' + json.dumps({'rep': rep, 'e_hip': e_hip, 'e_emu': e_emu}))\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("REPORT ")][-1]
    got = json.loads(line[len("REPORT "):])
    rep, e_hip, e_emu = got["rep"], got["e_hip"], got["e_emu"]
    for k, v in rep.items():
        REPORT[f"st[s_real,bfloat16].bench_selection.{k}"] = v
    assert rep["first_encoder_ffn"] == {"rows": 675, "one_launch_pair": True}, rep["first_encoder_ffn"]
    _assert_bf16_noise_level(rep, e_hip, e_emu)


def test_benchmark_shape_bf16_step_against_the_fp32_hip_path():
    """B = 128 x 900 frames, the batch bench.py times -- fused feed-forward on full 128-row tiles, 4 500-workgroup conv2 patch
    grids, 256-unit XCD-pinned split-K weight gradients -- in bf16 against the exact-fp32 HIP path on the same batch and the
    same (bf16-rounded) weights.  The fp32 path is pinned on the float64 oracle at 8e-6 (real configuration, B = 3); a float64
    oracle run of this batch would take the better part of an hour on the host.  Bounds: the bf16 rounding-noise level measured
    against the oracle at B = 3 (loss / logits: north-star 1e-2)."""
    from neurst_amd.criterions import build_criterion
    res = {}
    state = None
    for dtype in ("bfloat16", "float32"):
        model, inputs, cfg = _speech_case("s_real", dtype, batch=128)
        if dtype == "bfloat16":
            state = {n: p.data.detach().cpu().clone() for n, p in model.store.params.items()}
            for n, p in model.store.params.items():   # the fp32 path multiplies the same rounded weights
                if n.endswith("/kernel") and "conv1" not in n or n.endswith("shared/weights"):
                    state[n] = p.compute.detach().float().cpu()
        else:
            model.store.load_state_dict(state)
        dinp = {k: v.to(DEV) for k, v in inputs.items()}
        crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
        logits = model(dinp, is_training=True)
        loss = float(crit.reduce_loss(dinp, logits))
        rows = model._encoder._stacking_layers[0]._ffn_layer.layer._saved[1].shape[0]
        model.backward(crit.backward())
        torch.cuda.synchronize()
        res[dtype] = (logits.float().cpu(), loss, {n: p.grad.detach().float().cpu().clone() for n, p in model.store.params.items()}, rows)
        del model, logits, dinp
        torch.cuda.empty_cache()
    (lg16, loss16, g16, rows16), (lg32, loss32, g32, _) = res["bfloat16"], res["float32"]
    assert rows16 == 128 * 225
    per, glob = _grad_errors(g16, g32)
    tag = "st[s_real,B128].bf16_vs_fp32_hip"
    REPORT[tag + ".logits"], REPORT[tag + ".loss_abs_err"] = rel_err(lg16, lg32), abs(loss16 - loss32)
    REPORT[tag + ".grad_global_rel_l2"], REPORT[tag + ".grad_worst_rel_l2"] = glob, max(per.values())
    REPORT[tag + ".grad_worst_tensor"] = max(per, key=per.get)
    assert math.isfinite(loss16) and abs(loss16 - loss32) <= 1e-2 * max(1.0, abs(loss32)), (loss16, loss32)
    assert REPORT[tag + ".logits"] <= 3e-2, REPORT[tag + ".logits"]
    assert glob <= 3e-2, f"global gradient rel-L2 between the bf16 and the fp32 HIP step: {glob:.3e}"
    assert max(per.values()) <= 0.1, (REPORT[tag + ".grad_worst_tensor"], max(per.values()))


# ------------------------------------------------------------------------------------------------ text Transformer (§8(f) rank 1)
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("case", ["toy", "small_shared", "mid", "waitk3", "waitk1_mid", "base_shape", "big_shape"])
def test_text_transformer_forward_backward(case, dtype):
    """Transformer (embedding source side, optional shared source/target embedding) through the Seq2Seq task inputs:
    logits, loss and every gradient against the oracle's transformer_logits + criterion + autograd.  The waitk cases run
    WaitkTransformer (neurst/models/waitk_transformer.py): monotonic encoder + lagged cross attention."""
    from neurst_amd.criterions import build_criterion
    from neurst_amd.models import build_model
    from neurst_amd.models.transformer import Transformer
    from neurst_amd.tasks import build_task
    from neurst_amd.utils import compat
    cases = {  # d, H, enc, dec, ffn, B, S, L, Vs, Vt, share
        "toy": (8, 2, 2, 2, 10, 2, 5, 4, 11, 13, False),
        "small_shared": (64, 2, 2, 2, 128, 3, 17, 9, 40, 40, True),
        "mid": (256, 4, 2, 1, 512, 4, 33, 21, 300, 200, False),
        "waitk3": (64, 2, 2, 2, 128, 3, 17, 9, 40, 37, False),
        "waitk1_mid": (256, 4, 2, 2, 512, 3, 90, 70, 300, 200, False),
        # BASELINE config #4's real hyper-parameters (neurst/models/transformer.py:100-240: transformer_base / transformer_big --
        # d, heads, filter size, 32 003-word vocabularies) at 2 + 2 layers, 4 pairs of 32 + 32 positions: the 256 x 256 tile
        # kernel's dispatch (d_model >= 512) and the zero-padded vocabulary rows (32 003 is not a multiple of 8) inside a model
        "base_shape": (512, 8, 2, 2, 2048, 4, 32, 32, 32003, 32003, False),
        "big_shape": (1024, 16, 2, 2, 4096, 4, 32, 32, 32003, 32003, False),
    }
    d, H, ne, nd, ffn, B, S, L, Vs, Vt, share = cases[case]
    wait_k = {"waitk3": 3, "waitk1_mid": 1}.get(case, None)
    p = dict(Transformer.build_model_args_by_name("transformer_toy")["model.params"])
    p.update({"modality.dim": d, "modality.share_source_target_embedding": share, "encoder.num_layers": ne,
              "decoder.num_layers": nd, "encoder.hidden_size": d, "decoder.hidden_size": d,
              "encoder.num_attention_heads": H, "decoder.num_attention_heads": H, "encoder.filter_size": ffn,
              "decoder.filter_size": ffn})
    for k in list(p):
        if k.endswith("dropout_rate"):
            p[k] = 0.0
    task = build_task({"task.class": "translation", "task.params": {"src_vocab_size": Vs, "trg_vocab_size": Vt}})
    if wait_k is None:
        model = task.build_model({"model.class": "Transformer", "model.params": p}, device=DEV, dtype=dtype, init_seed=5)
    else:
        model = task.build_model({"model.class": "WaitkTransformer", "model.params": dict(p, wait_k=wait_k)}, device=DEV,
                                 dtype=dtype, init_seed=5)
        assert model.wait_k == wait_k and model.decode_lagging(True, None) == wait_k and model.decode_lagging(False, 4) == wait_k + 4
    g = torch.Generator().manual_seed(21)
    sd = {}
    for n, prm in model.store.params.items():
        if n.endswith("/bias") or n.endswith("/beta"):
            sd[n] = torch.randn(prm.shape, generator=g) * 0.05
        elif n.endswith("/gamma"):
            sd[n] = 1.0 + torch.randn(prm.shape, generator=g) * 0.1
    model.store.load_state_dict(sd, strict=False)

    def side(Lx, V, lens):
        ids = torch.randint(0, V - 3, (B, Lx), generator=g)
        return torch.where(torch.arange(Lx)[None] >= (lens[:, None] - 1), torch.full_like(ids, V - 1), ids)
    src_len = torch.tensor([S - (i * S) // (2 * B) for i in range(B)])
    trg_len = torch.tensor([max(1, L - i) for i in range(B)])
    batch = {"feature": side(S, Vs, src_len), "label": side(L, Vt, trg_len)}
    inputs = task.example_to_input(batch, compat.ModeKeys.TRAIN)
    assert inputs["src_length"].tolist() == src_len.tolist() and inputs["trg_length"].tolist() == trg_len.tolist()

    W = {n: prm.data.detach().cpu().clone() for n, prm in model.store.params.items()}
    if dtype == "bfloat16":
        for n, prm in model.store.params.items():
            if n.endswith("/kernel") or n.endswith("/weights"):
                W[n] = prm.compute.detach().float().cpu()
    cfg = {"num_enc": ne, "num_dec": nd, "num_heads": H}
    if wait_k is not None:
        cfg.update({"attention_monotonic": True, "wait_k": wait_k})
    loss_ref, logits_ref, grads_ref = O.text_train_step_reference({k: v.double() for k, v in W.items()}, inputs, cfg, 0.1)
    dinp = {k: v.to(DEV) for k, v in inputs.items()}
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(dinp, is_training=True)
    loss = crit.reduce_loss(dinp, logits)
    model.backward(crit.backward())
    tol = TOL[dtype]
    tag = f"text[{case},{dtype}]"
    check(tag + ".logits", logits, logits_ref, tol * (3 if dtype == "bfloat16" else 1))
    REPORT[tag + ".loss_abs_err"] = abs(float(loss) - float(loss_ref))
    assert abs(float(loss) - float(loss_ref)) <= tol * max(1.0, abs(float(loss_ref)))
    num = den = 0.0
    bad = []
    for n, prm in model.store.params.items():
        gg, r = prm.grad.detach().float().cpu().double(), grads_ref[n].double()
        e = rel_err(prm.grad, grads_ref[n]) if dtype == "float32" else float((gg - r).norm() / max(float(r.norm()), 1e-12))
        REPORT[f"{tag}.grad.{n}"] = e
        num += float(((gg - r) ** 2).sum())
        den += float((r ** 2).sum())
        # bf16: 1.5 x the worst per-tensor error measured on these small cases (8.3e-2, profiles/r01_model_parity_report.json;
        # few rows, so single ReLU-gate flips weigh most here)
        if not (e <= (2e-3 if dtype == "float32" else 0.125)):
            bad.append((n, e))
    glob = math.sqrt(num / max(den, 1e-30))
    REPORT[tag + ".grad_global_rel_l2"] = glob
    assert not bad, f"{tag}: gradients out of tolerance: {bad[:8]}"
    assert glob <= (1e-3 if dtype == "float32" else 3e-2), f"{tag}: global gradient rel-L2 error {glob:.3e}"


def test_gradient_accumulation_and_tied_embedding():
    from neurst_amd.criterions import build_criterion
    model, inputs, cfg = _speech_case("small", "float32")
    dinp = {k: v.to(DEV) for k, v in inputs.items()}
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    model.backward(crit_backward(model, crit, dinp))
    g1 = model.store.grad.clone()
    model.backward(crit_backward(model, crit, dinp), accumulate=True)
    check("accumulate.2x", model.store.grad, 2 * g1.cpu().double(), 1e-5)
    model.backward(crit_backward(model, crit, dinp))
    check("overwrite.1x", model.store.grad, g1.cpu().double(), 1e-5)


def crit_backward(model, crit, dinp):
    logits = model(dinp, is_training=True)
    crit.reduce_loss(dinp, logits)
    return crit.backward()


def test_train_steps_match_oracle_adam():
    """3 optimizer steps (fp32): loss trajectory and final weights vs the oracle's Keras-Adam + Noam loop."""
    from neurst_amd.criterions import build_criterion
    from neurst_amd.optimizers import build_lr_schedule, build_optimizer
    from neurst_amd.training.train_step import TrainStep
    model, inputs, cfg = _speech_case("small", "float32")
    W = {n: p.data.detach().cpu().double() for n, p in model.store.params.items()}
    dinp = {k: v.to(DEV) for k, v in inputs.items()}
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    sched_args = {"dmodel": 64, "warmup_steps": 4, "initial_factor": 2.0, "end_factor": 1.0, "start_decay_at": 2, "decay_steps": 4}
    opt = build_optimizer({"optimizer.class": "Adam", "optimizer.params": {"epsilon": 1e-9, "beta_1": 0.9, "beta_2": 0.98}})
    opt.bind(model.store)
    opt.learning_rate = build_lr_schedule({"lr_schedule.class": "noam", "lr_schedule.params": sched_args})
    step = TrainStep(model, crit, opt)
    m = {n: torch.zeros_like(v) for n, v in W.items()}
    v_ = {n: torch.zeros_like(v) for n, v in W.items()}
    oinp = {k: (v.double() if v.is_floating_point() else v) for k, v in inputs.items()}
    mask = None
    for t in range(1, 4):
        loss = step(dinp)
        loss_ref, _, g = O.train_step_reference(W, oinp, cfg, 0.1)
        if mask is None:
            # Adam turns a gradient that is zero up to rounding (e.g. the key bias: softmax is shift invariant)
            # into a +-lr step whose sign is noise; such elements are excluded from the weight comparison.
            mask = {n: (g[n].abs() > 1e-4 * g[n].abs().max()) for n in g}
        lr = O.noam_lr(t - 1, **sched_args)
        for n in W:
            W[n], m[n], v_[n] = O.keras_adam_step(W[n], g[n], m[n], v_[n], t, lr)
        REPORT[f"train.loss_err.step{t}"] = abs(float(loss) - float(loss_ref))
        assert abs(float(loss) - float(loss_ref)) < 1e-3
    worst = 0.0
    for n, p in model.store.params.items():
        diff = (p.data.detach().cpu().double() - W[n]).abs()[mask[n]]
        if diff.numel():
            worst = max(worst, float(diff.max()) / max(float(W[n].abs().max()), 1e-6))
    REPORT["train.final_weight_worst"] = worst
    assert worst < 2e-3


def test_dropout_training_step_runs_and_is_reproducible():
    from neurst_amd.criterions import build_criterion
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    hp = get_hyper_parameters("speech_transformer_toy")
    V = 20
    losses = []
    for rep in range(2):
        model = build_model(hp, {"audio_feature_dim": 16, "audio_feature_channels": 1},
                            {"vocab_size": V, "eos_id": V - 1, "bos_id": V - 2, "unk_id": V - 3}, device=DEV,
                            dtype="bfloat16", seed=99)
        g = torch.Generator().manual_seed(1)
        inputs = {"src": torch.randn(4, 30, 16, 1, generator=g).to(DEV), "src_length": torch.tensor([30, 28, 20, 30]).to(DEV),
                  "trg": torch.randint(0, V - 3, (4, 5), generator=g).to(DEV), "trg_length": torch.tensor([5, 5, 4, 3]).to(DEV)}
        inputs["trg_input"] = torch.cat([torch.full((4, 1), V - 2, device=DEV), inputs["trg"][:, :-1]], 1)
        crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
        logits = model(inputs, is_training=True)
        loss = crit.reduce_loss(inputs, logits)
        model.backward(crit.backward())
        assert torch.isfinite(model.store.grad).all()
        losses.append(float(loss))
        eval_logits = model(inputs, is_training=False)
        assert torch.isfinite(eval_logits.float()).all()
    assert losses[0] == losses[1]  # same seed, same step -> identical Philox masks


def test_cli_training_from_tfrecord_shards(tmp_path):
    """The reference's recipe layout end to end on the MI355X (body in tests/cli_flows.py, shared with the CPU tier)."""
    from cli_flows import cli_tfrecord_flow
    cli_tfrecord_flow(tmp_path, REPORT)


# ------------------------------------------------------------------------------------------------ inference (rank 3)
def _toy_speech_model(dtype, V=20, seed=5):
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    hp = get_hyper_parameters("speech_transformer_toy")
    return build_model(hp, {"audio_feature_dim": 16, "audio_feature_channels": 1},
                       {"vocab_size": V, "eos_id": V - 1, "bos_id": V - 2, "unk_id": V - 3}, device=DEV, dtype=dtype, seed=seed)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_incremental_decoding_matches_full_forward(dtype):
    """Step t of the cached decoder (multi_head_attention.py:254-290 cache branch, transformer_decoder.py:105-147) must give
    the logits the full teacher-forced forward gives at position t."""
    V, B, L = 20, 3, 7
    model = _toy_speech_model(dtype)
    g = torch.Generator().manual_seed(3)
    src = torch.randn(B, 40, 16, 1, generator=g).to(DEV)
    src_length = torch.tensor([40, 33, 21]).to(DEV)
    trg = torch.randint(0, V - 3, (B, L), generator=g).to(DEV)
    trg_input = torch.cat([torch.full((B, 1), V - 2, device=DEV), trg[:, :-1]], 1)
    full = model({"src": src, "src_length": src_length, "trg_input": trg_input}, is_training=False).float()
    fn, init, _ = model.get_symbols_to_logits_fn({"src": src, "src_length": src_length}, beam_size=1, decode_padded_length=L)
    assert init["decoder_input"].tolist() == [V - 2] * B and init["eos_id"] == V - 1
    cache = init["decoder_internal_cache"]
    for t in range(L):
        step = fn(trg_input[:, t], cache, t).float()
        check(f"decode.step{t}[{dtype}]", step, full[:, t].cpu(), 5 * TOL[dtype])
    assert all(st["self_attention"]["len"] == L for st in cache["decoding_states"].values())
    with pytest.raises(RuntimeError):
        fn(trg_input[:, 0], cache, L)           # the cache holds exactly decode_padded_length positions


def test_beam_search_on_the_model_matches_oracle_search():
    """The real cache (K/V buffers re-ordered with the beams) against the oracle search that scores every hypothesis with a
    fresh full forward of the same model -- a wrong gather of the cache changes the hypotheses."""
    from neurst_amd.layers.search import BeamSearch
    from oracle import beam_search_oracle as BO
    V, B = 20, 2
    model = _toy_speech_model("float32", seed=11)
    g = torch.Generator().manual_seed(8)
    src = torch.randn(B, 36, 16, 1, generator=g).to(DEV)
    src_length = torch.tensor([36, 25]).to(DEV)
    inputs = {"src": src, "src_length": src_length}

    def prefix_logits(sample, prefix):
        ti = torch.tensor([prefix], device=DEV)
        out = model({"src": src[sample:sample + 1], "src_length": src_length[sample:sample + 1], "trg_input": ti}, is_training=False)
        return out[0, -1].float().cpu().numpy()
    for beam, top_k, alpha in ((1, 1, 0.6), (3, 2, 0.6), (4, 1, 1.0)):
        search = BeamSearch(beam_size=beam, top_k=top_k, length_penalty=alpha, maximum_decode_length=8, extra_decode_length=50)
        hyp, scores = search(model, inputs)
        want_h, want_s = BO.beam_search(prefix_logits, B, V - 2, V - 1, V - 3, V, beam_size=beam, top_k=top_k, length_penalty=alpha,
                                        extra_decode_length=50, maximum_decode_length=8, encoder_len=9)
        assert hyp.shape == (B * top_k, 8)
        assert hyp.cpu().tolist() == want_h.tolist(), (beam, top_k, alpha)
        assert np.allclose(scores.cpu().numpy(), want_s, rtol=2e-3, atol=2e-3)
        REPORT[f"search.beam{beam}.score0"] = float(scores[0])
    # greedy == argmax roll-out of the full forward
    hyp, _ = BeamSearch(beam_size=1, maximum_decode_length=6)(model, inputs)
    prefix = torch.full((B, 1), V - 2, device=DEV)
    for t in range(6):
        nxt = model({"src": src, "src_length": src_length, "trg_input": prefix}, is_training=False)[:, -1].float()
        nxt[:, V - 3] = -1e9                    # UNK is masked by the search
        tok = nxt.argmax(-1)
        done = (prefix[:, 1:] == V - 1).any(1)
        tok = torch.where(done, torch.full_like(tok, V - 1), tok)
        assert hyp[:, t].tolist() == tok.tolist()
        prefix = torch.cat([prefix, tok[:, None]], 1)


def test_graph_replayed_decoding_equals_eager_decoding():
    """HIP-graph capture of the decoding steps (one graph per time index over static buffers) must not change a single
    hypothesis, also for a SECOND batch that replays the graphs captured for the first one."""
    from neurst_amd.layers.search import BeamSearch
    V, B = 20, 3
    model = _toy_speech_model("bfloat16", seed=21)
    eager = BeamSearch(beam_size=3, top_k=2, maximum_decode_length=9)
    graphed = BeamSearch(beam_size=3, top_k=2, maximum_decode_length=9, use_graphs=True)
    for seed in (1, 2, 3):
        g = torch.Generator().manual_seed(seed)
        inputs = {"src": torch.randn(B, 44, 16, 1, generator=g).to(DEV), "src_length": torch.tensor([44, 30, 37]).to(DEV)}
        h0, s0 = eager(model, inputs)
        h1, s1 = graphed(model, inputs)
        assert torch.equal(h0, h1) and torch.equal(s0, s1), seed
    assert len(model._decode_sessions) == 1 and len(next(iter(model._decode_sessions.values())).graphs) >= 1


def test_waitk_incremental_decoding_matches_training_mask():
    """WaitkTransformer inference (waitk_transformer.py:103-104: lagging + time positions visible at step `time`) must give,
    step by step, the logits of the teacher-forced forward under the training-time wait-k mask."""
    from neurst_amd.tasks import build_task
    from neurst_amd.models.transformer import Transformer
    Vs, Vt, B, S, L, k = 30, 26, 3, 14, 8, 2
    p = dict(Transformer.build_model_args_by_name("transformer_toy")["model.params"])
    for n in list(p):
        if n.endswith("dropout_rate"):
            p[n] = 0.0
    task = build_task({"task.class": "translation", "task.params": {"src_vocab_size": Vs, "trg_vocab_size": Vt}})
    model = task.build_model({"model.class": "WaitkTransformer", "model.params": dict(p, wait_k=k)}, device=DEV, dtype="float32", init_seed=9)
    g = torch.Generator().manual_seed(2)
    src = torch.randint(0, Vs - 3, (B, S), generator=g).to(DEV)
    src_length = torch.tensor([S, S - 3, S - 6]).to(DEV)
    trg_input = torch.cat([torch.full((B, 1), Vt - 2), torch.randint(0, Vt - 3, (B, L - 1), generator=g)], 1).to(DEV)
    full = model({"src": src, "src_length": src_length, "trg_input": trg_input}, is_training=False).float()
    fn, init, _ = model.get_symbols_to_logits_fn({"src": src, "src_length": src_length}, beam_size=1, decode_padded_length=L)
    for t in range(L):
        step = fn(trg_input[:, t], init["decoder_internal_cache"], t).float()
        check(f"waitk.decode.step{t}", step, full[:, t].cpu(), 5e-3)
    # and the mask matters: a full-attention Transformer with the same weights disagrees at the early steps
    model.wait_k = None
    other = model({"src": src, "src_length": src_length, "trg_input": trg_input}, is_training=False).float()
    assert float((other[:, 0] - full[:, 0]).abs().max()) > 1e-4
