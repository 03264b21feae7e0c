#!/usr/bin/env python3
"""Fixture of tests/test_gpu_model.py::test_speech_transformer_s_bf16_step_at_batch_32_against_the_oracle_fixture.

The float64 oracle (oracle/neurst_oracle.py) on the REAL speech_transformer_s configuration (12 + 6 layers, d = 256, ffn = 2048,
C = 256, V = 8008) at a batch of 32 ragged 900-frame utterances -- a batch at which bf16 rounding noise has averaged down, and
which no test can afford to push through the CPU oracle on the GPU box (minutes and tens of GB).  Run on the build host:

    python tests/golden/make_oracle_b32.py [32|128]   # 1 min / 9 GB at 32, ~5 min / ~40 GB at 128 (the benchmark batch);
                                                      # writes tests/golden/oracle_s_real_b<batch>.npz (~0.3 MB)

Kept per gradient tensor (and for the logits): the L2 norm and 64 fixed +-1 projections (oracle/projections.py) of
  * the oracle's result on the bf16-rounded GEMM weights the device path uses, and
  * the float64 EMULATION of every C-ABI contract (oracle/kernel_emulation.py: the same layer classes with bf16 rounding
    points between exact kernels) -- the distance of a perfect bf16 path from the oracle at this batch.
Weights and inputs are NOT stored: the test rebuilds them from the same seeds (tests/test_gpu_model.py::_speech_case)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import neurst_oracle as O  # noqa: E402
from oracle import kernel_emulation as E  # noqa: E402
from oracle.projections import project_all, sign_projections  # noqa: E402

BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 32     # 128 = the benchmark batch (~5 min, ~40 GB)


def main():
    import pytest
    import test_gpu_model as T
    from neurst_amd.criterions import build_criterion
    t0 = time.time()
    mp = pytest.MonkeyPatch()
    try:
        E.install(mp)      # the layer classes over emulated kernels: CPU tensors, bf16 rounding points, exact arithmetic inside
        model, inputs, cfg = T._speech_case("s_real", "bfloat16", device="cpu", ragged_batch=BATCH)
        W = {n: p.data.detach().clone() for n, p in model.store.params.items()}
        for n, p in model.store.params.items():      # the oracle sees the bf16-rounded GEMM weights the device path uses
            if n.endswith("/kernel") and "conv1" not in n or n.endswith("shared/weights"):
                W[n] = p.compute.detach().float()
        crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
        lg = model(inputs, is_training=True)
        loss_emu = float(crit.reduce_loss(inputs, lg))
        model.backward(crit.backward())
        grads_emu = {n: p.grad.detach().double().clone() for n, p in model.store.params.items()}
        logits_emu = lg.double().clone()
        del model, lg
    finally:
        mp.undo()
    print(f"emulation done: loss {loss_emu:.6f}, {time.time() - t0:.0f} s", flush=True)
    loss_ref, logits_ref, grads_ref = O.train_step_reference({k: v.double() for k, v in W.items()},
                                                             {k: (v.double() if v.is_floating_point() else v)
                                                              for k, v in inputs.items()}, cfg, 0.1)
    print(f"oracle done: loss {float(loss_ref):.6f}, {time.time() - t0:.0f} s", flush=True)
    names, proj_ref, norm_ref = project_all(grads_ref)
    names2, proj_emu, norm_emu = project_all(grads_emu)
    assert names == names2
    diff = {n: grads_emu[n] - grads_ref[n].double() for n in names}
    err_emu = np.array([float(diff[n].norm() / max(float(grads_ref[n].double().norm()), 1e-12)) for n in names])
    glob_emu = float(np.sqrt(sum(float((diff[n] ** 2).sum()) for n in names) / sum(float((grads_ref[n].double() ** 2).sum()) for n in names)))
    out = os.path.join(ROOT, "tests", "golden", f"oracle_s_real_b{BATCH}.npz")
    np.savez_compressed(
        out, names=np.array(names), batch=BATCH, loss_ref=float(loss_ref), loss_emu=loss_emu,
        grad_proj_ref=proj_ref.numpy(), grad_norm_ref=norm_ref.numpy(), grad_proj_emu=proj_emu.numpy(), grad_norm_emu=norm_emu.numpy(),
        grad_rel_l2_emu=err_emu, grad_global_rel_l2_emu=glob_emu,
        logits_proj_ref=sign_projections(logits_ref, "logits").numpy(), logits_norm_ref=float(logits_ref.double().norm()),
        logits_proj_emu=sign_projections(logits_emu, "logits").numpy(),
        logits_max_abs_ref=float(logits_ref.abs().max()), logits_max_abs_err_emu=float((logits_emu - logits_ref.double()).abs().max()))
    pe = float(np.linalg.norm(proj_emu.numpy() - proj_ref.numpy()) / np.linalg.norm(proj_ref.numpy()))
    print(f"wrote {out}: emulation vs oracle gradients global rel-L2 {glob_emu:.3e} (by projections {pe:.3e}), "
          f"worst tensor {err_emu.max():.3e}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
