#!/usr/bin/env python3
"""Which bf16 rounding points of the training step carry the gradient error against the float64 oracle?  (CPU only.)

The float64 emulation of the C-ABI contracts (oracle/kernel_emulation.py) rounds every kernel OUTPUT to the model's dtype.
Here the model is built in float32 (so that output conversion is harmless) and the bf16 rounding is re-applied by
wrappers, one CLASS of kernel outputs at a time:

  resid     the forward residual stream: embedding / front dense output, every GEMM (or feed-forward pair) whose epilogue
            adds the residual
  ln_y      LayerNorm outputs
  proj      q|k|v, q, k|v projections (forward GEMMs without residual)
  ctx       attention context
  ffn_h     the saved feed-forward hidden activation (ReLU output)
  logits    the logits
  conv1     conv1 + LayerNorm + ReLU output (one kernel: the largest activation of the model)          } round 5 lumped these
  conv2     conv2 output (before its LayerNorm)                                                       } three into "conv" /
  ln_relu   the LayerNorm + ReLU output behind conv2 = the input of the front end's output_dense       } "ln_y"; split in round 6
  dlogits   d(loss)/d(logits)
  ln_dx     LayerNorm backward output = the BACKWARD residual stream (dx + d(residual))
  dgrad     outputs of input-gradient GEMMs (d context, d(LN output), d memory, d decoder output)
  dqkv      attention backward outputs
  ffn_dh    the gated hidden gradient
  conv_dx   conv2 data gradient
  +delta    (with resid lifted) the sub-layer's contribution dropout(layer(LN(x))) is rounded to bf16 BEFORE it is added to the
            fp32 stream: what a fused residual-add + LayerNorm kernel reading a bf16 GEMM output computes

`all` = every class rounded (must reproduce the bf16 emulation of tests/golden/make_oracle_b32.py); `all-X` = class X kept
in fp32.  Output: profiles/r06_rounding_point_study_b<batch>.json (global rel-L2 of the 280 gradient tensors against the oracle run on
the same bf16-rounded weights, worst tensor, logits error).

    python scripts/rounding_point_study.py [batch=32] [config ...]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import kernel_emulation as E  # noqa: E402
from oracle import neurst_oracle as O  # noqa: E402

EXTRA = ["delta"]   # not a class of the bf16 path: only meaningful with `resid` lifted (see gemm below)
CLASSES = ["resid", "ln_y", "proj", "ctx", "ffn_h", "logits", "conv1", "conv2", "ln_relu", "dlogits", "ln_dx", "dgrad", "dqkv", "ffn_dh", "conv_dx"]
ROUND = set(CLASSES)
PHASE = ["fwd"]
VOCAB_MIN = 8000


def _r(t, cls):
    if t is not None and cls in ROUND and t.dtype == torch.float32:
        t.copy_(t.bfloat16().float())
    return t


def _wrap():
    o = {n: getattr(E, n) for n in E._NAMES}

    def layernorm_fwd(x, *a, **k):
        y, m, r = o["layernorm_fwd"](x, *a, **k)
        relu = k.get("relu", a[3] if len(a) > 3 else False)
        return _r(y, "ln_relu" if relu else "ln_y"), m, r

    def layernorm_bwd(*a, **k):
        assert k.get("emit_dropout") is None
        return _r(o["layernorm_bwd"](*a, **k), "ln_dx")

    def gemm(A, B, M, N, K, trans_a=False, trans_b=False, **k):
        out = o["gemm"](A, B, M, N, K, trans_a=trans_a, trans_b=trans_b, **k)
        if trans_a or out.dtype != A.dtype or k.get("split_k", 1) > 1:
            return out                                    # weight gradients stay fp32 on the device as well
        if N >= VOCAB_MIN:
            cls = "logits"
        elif k.get("relu"):
            cls = "ffn_h"
        elif k.get("gate_src") is not None:
            cls = "ffn_dh"
        elif PHASE[0] == "fwd":
            cls = "resid" if (k.get("residual") is not None or k.get("posenc") is not None) else "proj"
        else:
            cls = "dgrad"
        assert not k.get("accumulate"), "accumulating activation GEMMs would need the rounding after the sum: not on this path"
        if cls == "resid" and "resid" not in ROUND and "delta" in ROUND and k.get("residual") is not None:
            # fp32 residual stream whose sub-layer contribution travels as bf16: x_new = x_old + bf16(dropout(layer(LN(x_old))))
            res = k["residual"]
            out.copy_(res + (out - res).bfloat16().float())
            return out
        return _r(out, cls)

    def attention_fwd(*a, **k):
        out, lse, mask = o["attention_fwd"](*a, **k)
        return _r(out, "ctx"), lse, mask

    def attention_bwd(q, k_, v, out, dout, lse, dq, dk, dv, *a, **k):
        o["attention_bwd"](q, k_, v, out, dout, lse, dq, dk, dv, *a, **k)
        for t in (dq, dk, dv):      # strided views into the packed buffer: round in place
            if "dqkv" in ROUND:
                t.copy_(t.bfloat16().float())

    def conv1_ln_relu_fwd(*a, **k):
        y, m, r = o["conv1_ln_relu_fwd"](*a, **k)
        return _r(y, "conv1"), m, r

    def conv2_fwd(*a, **k):
        return _r(o["conv2_fwd"](*a, **k), "conv2")

    def conv2_dgrad(*a, **k):
        return _r(o["conv2_dgrad"](*a, **k), "conv_dx")

    def embedding_fwd(*a, **k):
        return _r(o["embedding_fwd"](*a, **k), "resid")

    def scale_posenc_dropout_fwd(*a, **k):
        return _r(o["scale_posenc_dropout_fwd"](*a, **k), "resid")

    def ls_xent_bwd(*a, **k):
        return _r(o["ls_xent_bwd"](*a, **k), "dlogits")

    new = dict(o)
    new.update(layernorm_fwd=layernorm_fwd, layernorm_bwd=layernorm_bwd, gemm=gemm, attention_fwd=attention_fwd,
               attention_bwd=attention_bwd, conv1_ln_relu_fwd=conv1_ln_relu_fwd, conv2_fwd=conv2_fwd, conv2_dgrad=conv2_dgrad,
               embedding_fwd=embedding_fwd, scale_posenc_dropout_fwd=scale_posenc_dropout_fwd, ls_xent_bwd=ls_xent_bwd)

    def gemm_wgrad_group(items, table=None):
        for x, dz, out, acc, cs, cs_acc in items:
            o["gemm"](x, dz, x.shape[1], dz.shape[1], x.shape[0], trans_a=True, out=out, accumulate=acc, colsum_out=cs,
                      colsum_accumulate=cs_acc)
    new["gemm_wgrad_group"] = gemm_wgrad_group
    return new


def run_emulation(batch, rounded):
    import pytest
    import test_gpu_model as T
    from neurst_amd import kernels as K
    from neurst_amd.criterions import build_criterion
    ROUND.clear()
    ROUND.update(rounded)
    mp = pytest.MonkeyPatch()
    try:
        for n, f in _wrap().items():
            mp.setattr(K, n, f)
        model, inputs, cfg = T._speech_case("s_real", "float32", device="cpu", ragged_batch=batch)
        for n, p in model.store.params.items():        # the GEMM weights the bf16 device path computes with
            if n.endswith("/kernel") and "conv1" not in n or n.endswith("shared/weights"):
                p.data.copy_(p.data.bfloat16().float())
        model.store.refresh_shadow()
        model.store.refresh_transposed()          # the packed k|v copies follow the rounded kernels
        W = {n: p.data.detach().clone() for n, p in model.store.params.items()}
        crit = build_criterion({"criterion.class": "label_smoothed_cross_entropy", "criterion.params": {"label_smoothing": 0.1}})
        PHASE[0] = "fwd"
        lg = model(inputs, is_training=True)
        loss = float(crit.reduce_loss(inputs, lg))
        PHASE[0] = "bwd"
        model.backward(crit.backward())
        grads = {n: p.grad.detach().double().clone() for n, p in model.store.params.items()}
        return W, inputs, cfg, loss, lg.double().clone(), grads
    finally:
        PHASE[0] = "fwd"
        mp.undo()


def compare(grads, grads_ref, logits, logits_ref):
    num = den = 0.0
    worst, worst_name = 0.0, ""
    per = {}
    for n, r in grads_ref.items():
        r = r.double()
        d = grads[n] - r
        e = float(d.norm() / max(float(r.norm()), 1e-12))
        per[n] = (e, float((d ** 2).sum()))
        if e > worst:
            worst, worst_name = e, n
        num += float((d ** 2).sum())
        den += float((r ** 2).sum())
    top = sorted(per.items(), key=lambda kv: -kv[1][1])[:5]
    return {"grad_global_rel_l2": float(np.sqrt(num / den)), "grad_worst": worst, "grad_worst_name": worst_name,
            "n_above_1e-2": sum(1 for e, _ in per.values() if e > 1e-2),
            "top_error_mass": [(n, round(s / num, 4), round(e, 5)) for n, (e, s) in top],
            "logits_rel_l2": float((logits - logits_ref.double()).norm() / logits_ref.double().norm())}


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    configs = sys.argv[2:] or (["all"] + [f"all-{c}" for c in CLASSES] + ["all-resid-ln_dx", "none"])
    torch.set_num_threads(os.cpu_count() or 8)
    out_path = os.path.join(ROOT, "profiles", f"r06_rounding_point_study_b{batch}.json")
    results = json.load(open(out_path))["results"] if os.path.exists(out_path) else {}
    ref = None
    for c in configs:
        t0 = time.time()
        if c == "none":
            rounded = set()
        else:
            toks = c.split("+")[0].split("-")
            rounded = (set(CLASSES) - set(toks[1:])) | set(c.split("+")[1:])
            assert toks[0] == "all" and all(x in CLASSES for x in toks[1:]) and all(x in EXTRA for x in c.split("+")[1:]), c
        W, inputs, cfg, loss, logits, grads = run_emulation(batch, rounded)
        if ref is None:
            loss_ref, logits_ref, grads_ref = O.train_step_reference(
                {k: v.double() for k, v in W.items()}, {k: (v.double() if v.is_floating_point() else v) for k, v in inputs.items()},
                cfg, 0.1)
            ref = (float(loss_ref), logits_ref, grads_ref)
            print(f"oracle: loss {ref[0]:.6f} ({time.time() - t0:.0f} s)", flush=True)
        res = compare(grads, ref[2], logits, ref[1])
        res["loss_abs_err"] = abs(loss - ref[0])
        res["seconds"] = round(time.time() - t0, 1)
        results[c] = res
        print(f"{c:22s} global {res['grad_global_rel_l2']:.3e}  worst {res['grad_worst']:.3e}  >1e-2: {res['n_above_1e-2']:3d}  "
              f"logits {res['logits_rel_l2']:.2e}  top {res['top_error_mass'][:2]}", flush=True)
        json.dump({"batch": batch, "classes": CLASSES, "results": results}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
