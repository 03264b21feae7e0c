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
