import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # the GPU tests run in the configuration of the training entry points (one hardware queue per stream-priority class);
    # applied here, before any test touches the device -- importing neurst_amd does not do it
    from neurst_amd.runtime import configure_training_process
    configure_training_process()


def load_golden(name):
    import numpy as np
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrays = {k: z[k] for k in z.files}
    W = {k[2:]: torch.from_numpy(v.astype("float32")) for k, v in arrays.items() if k.startswith("w:")}
    rest = {k: v for k, v in arrays.items() if not k.startswith("w:")}
    return rest, W


@pytest.fixture(scope="session")
def golden():
    return load_golden


def load_reference_pt_case(tag):
    """Golden of the reference's own PyTorch SpeechTransformer run under tests/golden/make_golden.py
    (gen_neurst_pt_speech_transformer): inputs, weights (TF names / layouts), logits, loss and every gradient."""
    import torch
    r, W = load_golden(tag)
    grads = {k[2:]: torch.from_numpy(v) for k, v in r.items() if k.startswith("g:")}
    inputs = {k: torch.from_numpy(r[k]) for k in ("src", "src_length", "trg", "trg_input", "trg_length")}
    post = bool(int(r["post_norm"])) if "post_norm" in r else False
    cfg = {"num_enc": int(r["n_enc"]), "num_dec": int(r["n_dec"]), "num_heads": 2, "layer_norm": True,
           "timing": str(r["timing"]) or None, "encoder_post_normalize": post, "decoder_post_normalize": post}
    return inputs, W, cfg, torch.from_numpy(r["expected_logits"]), float(r["expected_loss"]), grads


def build_speech_model_for_reference_case(W, cfg, logits_ref, device, dtype="float32"):
    """speech_transformer_toy with the layer counts / timing / post-norm / weight tying of a reference-PT golden case,
    loaded with the golden's weights."""
    from neurst_amd.models import build_model
    from neurst_amd.utils.hparams_sets import get_hyper_parameters
    p = dict(get_hyper_parameters("speech_transformer_toy")["model.params"])
    p.update({"encoder.num_layers": cfg["num_enc"], "decoder.num_layers": cfg["num_dec"], "modality.timing": cfg["timing"],
              "encoder.post_normalize": cfg["encoder_post_normalize"], "decoder.post_normalize": cfg["decoder_post_normalize"],
              "modality.share_embedding_and_softmax_weights": "softmax_linear/kernel" not in W})
    for k in list(p):
        if k.endswith("dropout_rate"):
            p[k] = 0.0
    V = logits_ref.shape[-1]
    model = build_model({"model.class": "SpeechTransformer", "model.params": p}, {"audio_feature_dim": 80, "audio_feature_channels": 1},
                        {"vocab_size": V, "eos_id": V - 1, "bos_id": V - 2, "unk_id": V - 3}, device=device, dtype=dtype)
    assert set(model.store.params) == set(W), set(model.store.params) ^ set(W)
    model.store.load_state_dict(W)
    return model


def load_reference_pt_text_case(tag):
    """Same for the reference's PyTorch text Transformer (gen_neurst_pt_transformer)."""
    import torch
    r, W = load_golden(tag)
    grads = {k[2:]: torch.from_numpy(v) for k, v in r.items() if k.startswith("g:")}
    inputs = {k: torch.from_numpy(r[k]) for k in ("src", "src_length", "trg", "trg_input", "trg_length")}
    cfg = {"num_enc": 2, "num_dec": 2, "num_heads": 2}
    return inputs, W, cfg, torch.from_numpy(r["expected_logits"]), float(r["expected_loss"]), grads, bool(int(r["share"]))


def build_text_model_for_reference_case(W, logits_ref, share, device, dtype="float32"):
    from neurst_amd.models import build_model
    from neurst_amd.models.transformer import Transformer
    p = dict(Transformer.build_model_args_by_name("transformer_toy")["model.params"])
    p["modality.share_source_target_embedding"] = share
    for k in list(p):
        if k.endswith("dropout_rate"):
            p[k] = 0.0
    Vt = logits_ref.shape[-1]
    Vs = Vt if share else W["input_symbol_modality/emb/weights"].shape[0]
    meta = lambda V: {"vocab_size": V, "eos_id": V - 1, "bos_id": V - 2, "unk_id": V - 3}   # noqa: E731
    model = build_model({"model.class": "Transformer", "model.params": p}, meta(Vs), meta(Vt), device=device, dtype=dtype)
    assert set(model.store.params) == set(W), set(model.store.params) ^ set(W)
    model.store.load_state_dict(W)
    return model
