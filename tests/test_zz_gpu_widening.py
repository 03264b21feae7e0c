"""GPU parity of the options widened after the hot path met its bar (§8(f) rank 1 leftovers): post-norm stacks
(`encoder.post_normalize` / `decoder.post_normalize`, common_layers.py:86-92) and untied logits (`softmax_linear`,
encoder_decoder_model.py:63-67).  Same oracle, same tolerances as tests/test_gpu_model.py; the host schedule of these
options is additionally pinned on CPU by tests/test_host_path_cpu.py.  (The file sorts last on purpose: these cases
were added when no GPU time was left in the round, so they run after every test that has already been seen green.)"""
import math

import pytest
import torch

from oracle import neurst_oracle as O
from test_gpu_model import DEV, REPORT, TOL, _speech_case, check, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_gemm_dgrad_with_residual_epilogue(dtype):
    """dx = dz @ W^T + r in one launch (trans_b + residual): the post-norm wrapper's backward relies on it."""
    from neurst_amd import kernels as K
    td = torch.float32 if dtype == "float32" else torch.bfloat16
    g = torch.Generator().manual_seed(3)
    for M, N, Kd in ((150, 96, 264), (257, 256, 768), (64, 40, 50)):
        dz = (torch.randn(M, Kd, generator=g) * 0.5).to(td)
        W = (torch.randn(N, Kd, generator=g) * 0.5).to(td)
        r = torch.randn(M, N, generator=g).to(td)
        out = K.gemm(dz.to(DEV), W.to(DEV), M, N, Kd, trans_b=True, residual=r.to(DEV))
        check(f"gemm_dgrad_residual[{dtype},{M}x{N}x{Kd}]", out, dz.double() @ W.double().t() + r.double(), TOL[dtype])


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("variant", ["post_norm", "post_norm_encoder_only", "untied_softmax", "post_norm_untied"])
def test_post_norm_and_untied_softmax(variant, dtype):
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
    model, inputs, cfg = _speech_case("small", dtype, **extra)
    cfg.update(cfg_extra)
    W = {n: p.data.detach().cpu().clone() for n, p in model.store.params.items()}
    if dtype == "bfloat16":  # the oracle sees the same (bf16-rounded) GEMM weights the device path uses
        for n, p in model.store.params.items():
            if n.endswith("/kernel") and "conv1" not in n or n.endswith("/weights"):
                W[n] = p.compute.detach().float().cpu()
    loss_ref, logits_ref, grads_ref = O.train_step_reference(
        {k: v.double() for k, v in W.items()}, {k: (v.double() if v.is_floating_point() else v) for k, v in inputs.items()},
        cfg, 0.1)
    dinp = {k: v.to(DEV) for k, v in inputs.items()}
    crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
    logits = model(dinp, is_training=True)
    loss = crit.reduce_loss(dinp, logits)
    model.backward(crit.backward())
    tol, tag = TOL[dtype], f"st_{variant}[{dtype}]"
    check(tag + ".logits", logits, logits_ref, tol * (3 if dtype == "bfloat16" else 1))
    assert abs(float(loss) - float(loss_ref)) <= tol * max(1.0, abs(float(loss_ref)))
    num = den = 0.0
    bad = []
    for n, p in model.store.params.items():
        g, r = p.grad.detach().float().cpu().double(), grads_ref[n].double()
        e = rel_err(p.grad, grads_ref[n]) if dtype == "float32" else float((g - r).norm() / max(float(r.norm()), 1e-12))
        REPORT[f"{tag}.grad.{n}"] = e
        num += float(((g - r) ** 2).sum())
        den += float((r ** 2).sum())
        if not e <= (2e-3 if dtype == "float32" else 0.25):
            bad.append((n, e))
    glob = math.sqrt(num / max(den, 1e-30))
    REPORT[tag + ".grad_global_rel_l2"] = glob
    assert not bad, f"{tag}: gradients out of tolerance: {bad[:8]}"
    assert glob <= (1e-3 if dtype == "float32" else 3e-2), f"{tag}: global gradient rel-L2 error {glob:.3e}"
