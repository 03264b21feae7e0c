#!/usr/bin/env python3
"""Developer aid: pure HOST time of one benchmark-shaped training step (speech_transformer_s, 128 x 900 frames, bf16) with the
library calls replaced by no-ops -- i.e. the Python / ctypes / allocator work needed to ISSUE the step's ~450 launches.
Runs on the CPU (no GPU needed); says how far the step is from being launch-bound (lever 5).

    python scripts/host_overhead.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurst_amd import kernels as K
class FakeLib:
    def __getattr__(self, n):
        if n == "nst_attention_dropout_mask_bytes":
            return lambda d: 128 * d.contents.B * d.contents.H * ((d.contents.Tq + 15)//16) * ((d.contents.Tk + 63)//64) if hasattr(d,'contents') else 1024
        return lambda *a: 0
calls = [0]
class CountLib(FakeLib):
    def __getattr__(self, n):
        f = FakeLib.__getattr__(self, n)
        def g(*a):
            calls[0] += 1
            return f(*a)
        return g
K.lib = CountLib()
K._p = lambda t: None if t is None else t.data_ptr()
K._stream = lambda: 0
_ws = {}
def _workspace(nbytes, device):
    if 'w' not in _ws: _ws['w'] = torch.empty(64<<20, dtype=torch.uint8)
    return _ws['w']
K._workspace = _workspace
import ctypes
# attention mask bytes needs desc: patch attention_fwd's mask alloc by making byref passthrough
orig_byref = K.C.byref
from neurst_amd.criterions import build_criterion
from neurst_amd.optimizers import build_optimizer, build_lr_schedule
from neurst_amd.tasks import build_task
from neurst_amd.training.train_step import TrainStep
from neurst_amd.utils.hparams_sets import get_hyper_parameters
from neurst_amd.data.datasets.synthetic_speech import SyntheticSpeechDataset
from neurst_amd.utils import compat
hp = get_hyper_parameters("speech_transformer_s")
B,T,F,V = 128, 900, 80, 8008
task = build_task({"task.class": "speech2text", "task.params": {"audio_feature_dim": F, "vocab_size": V}})
model = task.build_model(hp, device="cpu", dtype="bfloat16", seed=1, init_seed=42)
crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
opt = build_optimizer({"optimizer.class": hp["optimizer.class"], "optimizer.params": hp["optimizer.params"]}); opt.bind(model.store)
opt.learning_rate = build_lr_schedule({"lr_schedule.class": hp["lr_schedule.class"], "lr_schedule.params": hp["lr_schedule.params"]})
step = TrainStep(model, crit, opt, None)
ds = SyntheticSpeechDataset({"batch_per_gpu": B, "frames": T, "feature_dim": F, "trg_len": 75, "vocab_size": V, "seed": 1})
it = ds.build_iterator(map_func=lambda b: task.example_to_input(b, compat.ModeKeys.TRAIN), device="cpu")
batch = next(it)
import cProfile, pstats
for i in range(3): step(batch)
calls[0] = 0
n, best = 10, 1e9
for i in range(n):
    t0 = time.perf_counter()
    step(batch)
    best = min(best, time.perf_counter() - t0)
t0 = time.perf_counter()
model.store.grad.zero_()          # on the GPU this is one asynchronous memset; on the CPU it is a real 117 MB pass
zero = time.perf_counter() - t0
print(f"host time per step with null kernels: best of {n} = {best*1e3:.2f} ms (of which the CPU-side gradient memset "
      f"{zero*1e3:.2f} ms), library calls per step: {calls[0]//n}")
pr = cProfile.Profile(); pr.enable()
for i in range(3): step(batch)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
