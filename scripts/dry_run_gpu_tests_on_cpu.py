#!/usr/bin/env python3
"""Developer aid (NOT a test tier, NOT a product path): runs the `-m gpu` tests in-process on the CPU with
oracle/kernel_emulation.py installed over neurst_amd.kernels and every test module's DEV switched to "cpu".

What it is for: when no GPU box is at hand, this shows whether a change to the HOST side (layer scheduling, caches, test
code itself) still drives the GPU parity tests to the end, and -- because the emulation rounds to bf16 between kernels
like the device path does -- what bf16 error level to expect.  It says nothing about the HIP kernels: the real tier
is `pytest -m gpu` on an MI355X.  Tests that need device dropout masks, HIP graphs, streams or the kernels' own probes
are reported as skipped.

    python scripts/dry_run_gpu_tests_on_cpu.py [pytest args, e.g. tests/test_gpu_model.py -k waitk]
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SKIP_MARKERS = ("No HIP GPUs", "no ROCm-capable device", "no dropout", "Torch not compiled with CUDA", "CUDA", "cuda", "ROCm device tensors", "probe")


class _Emulate(object):
    def pytest_sessionstart(self, session):
        from neurst_amd import kernels as K
        from oracle import kernel_emulation as E
        for n in E._NAMES:
            setattr(K, n, getattr(E, n))
        import torch
        torch.cuda.synchronize = lambda *a, **k: None     # host-side flow only: there is no device to wait for
        torch.cuda.empty_cache = lambda *a, **k: None

    def pytest_collection_modifyitems(self, session, config, items):
        for mod in list(sys.modules.values()):   # test modules import helpers (and DEV) from each other
            if getattr(mod, "__name__", "").startswith("test_") and getattr(mod, "DEV", None) is not None:
                mod.DEV = "cpu"

    @pytest.hookimpl(hookwrapper=True)
    def pytest_runtest_call(self, item):
        outcome = yield
        exc = outcome.excinfo
        if exc is not None and not isinstance(exc[1], AssertionError) and any(m in str(exc[1]) for m in SKIP_MARKERS):
            outcome.force_exception(pytest.skip.Exception(f"needs the device: {str(exc[1])[:80]}"))


if __name__ == "__main__":
    args = sys.argv[1:] or [os.path.join(ROOT, "tests")]
    sys.exit(pytest.main(["-m", "gpu", "-q", "-p", "no:cacheprovider", "-rs"] + args, plugins=[_Emulate()]))
