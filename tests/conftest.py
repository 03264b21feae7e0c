import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


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
    cfg = {"num_enc": int(r["n_enc"]), "num_dec": int(r["n_dec"]), "num_heads": 2, "layer_norm": True,
           "timing": str(r["timing"]) or None}
    return inputs, W, cfg, torch.from_numpy(r["expected_logits"]), float(r["expected_loss"]), grads
