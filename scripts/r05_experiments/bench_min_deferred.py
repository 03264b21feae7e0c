#!/usr/bin/env python3
"""bench.py with _StepCapture.min_deferred (weight-gradient calls that make a layer boundary cut the captured step) replaced:
   python scripts/r05_experiments/bench_min_deferred.py <n> [bench.py arguments]"""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
n = int(sys.argv[1])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
import neurst_amd.training.train_step as ts  # noqa: E402

_init = ts._StepCapture.__init__


def patched(self, *a, **k):
    _init(self, *a, **k)
    self.min_deferred = n


ts._StepCapture.__init__ = patched
runpy.run_path(sys.argv[0], run_name="__main__")
