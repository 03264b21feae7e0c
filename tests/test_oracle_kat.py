"""Known-answer tests for the parts of the oracle that NO reference test pins (criterion, gradients, Adam/Noam,
data-parallel averaging): closed forms and finite differences.  Parity for these rows is 'unpinned' w.r.t. the
reference; these tests only guarantee the restatement is self-consistent with the documented math."""
import math

import numpy as np

import pytest
import torch

from oracle import neurst_oracle as O


def test_ls_xent_uniform_logits_closed_form():
    # uniform logits: log_softmax = -ln V for every class -> xent = ln V (soft targets sum to 1);
    # minus the normalising constant of label_smoothed_cross_entropy.py:131-136
    V, eps = 11, 0.1
    logits = torch.zeros(2, 3, V)
    labels = torch.tensor([[1, 2, 3], [4, 5, 6]])
    nll, ns, nt = O.label_smoothed_cross_entropy(logits, labels, torch.tensor([3, 2]), eps)
    conf, low = 1 - eps, eps / (V - 1)
    norm = -(conf * math.log(conf) + (V - 1) * low * math.log(low + 1e-20))
    per_tok = math.log(V) - norm
    assert torch.allclose(nll, torch.tensor([3 * per_tok, 2 * per_tok]), atol=1e-6)
    assert ns.tolist() == [2.0] and nt.tolist() == [3.0, 2.0]
    assert abs(float(O.reduce_loss(nll, nt)) - per_tok) < 1e-6


def test_ls_xent_zero_smoothing_is_plain_nll_and_perfect_prediction_is_zero():
    logits = torch.randn(2, 4, 7, generator=torch.Generator().manual_seed(0))
    labels = torch.randint(0, 7, (2, 4), generator=torch.Generator().manual_seed(1))
    nll, _, _ = O.label_smoothed_cross_entropy(logits, labels, torch.tensor([4, 4]), 0.0)
    ref = torch.nn.functional.cross_entropy(logits.reshape(-1, 7), labels.reshape(-1), reduction="none").reshape(2, 4).sum(1)
    assert torch.allclose(nll, ref, atol=1e-5)
    # with smoothing, logits equal to log(soft target) reach the minimum, which the constant shifts to 0
    V, eps = 7, 0.2
    soft = torch.full((1, 1, V), eps / (V - 1))
    soft[0, 0, 3] = 1 - eps
    nll, _, _ = O.label_smoothed_cross_entropy(torch.log(soft), torch.tensor([[3]]), torch.tensor([1]), eps)
    assert abs(float(nll)) < 1e-6


def test_oracle_gradients_match_finite_differences_fp64():
    cfg = {"num_enc": 1, "num_dec": 1, "num_heads": 2, "layer_norm": True, "d_model": 8, "channels": 3, "ffn": 6}
    V, B, T, F, L = 7, 2, 9, 8, 3
    W = O.init_speech_transformer_weights(cfg, V, F, 1, seed=5, dtype=torch.float64)
    g = torch.Generator().manual_seed(2)
    for k in W:
        if k.endswith("bias") or k.endswith("beta"):
            W[k] = torch.randn(W[k].shape, generator=g, dtype=torch.float64) * 0.1
    trg = torch.randint(0, V - 3, (B, L), generator=g)
    inputs = {"src": torch.randn(B, T, F, 1, generator=g, dtype=torch.float64), "src_length": torch.tensor([T, T - 4]),
              "trg": trg, "trg_length": torch.tensor([L, L - 1]),
              "trg_input": torch.cat([torch.full((B, 1), V - 2), trg[:, :-1]], 1)}
    loss, _, grads = O.train_step_reference(W, inputs, cfg, 0.1)

    def f(Wp):
        logits = O.speech_transformer_logits(inputs, Wp, cfg)
        nll, _, nt = O.label_smoothed_cross_entropy(logits, inputs["trg"], inputs["trg_length"], 0.1)
        return float(O.reduce_loss(nll, nt))

    rng = torch.Generator().manual_seed(3)
    for name in ["input_audio_modality/conv1/kernel", "input_audio_modality/conv2/kernel",
                 "input_audio_modality/ln1/gamma", "target_symbol_modality/shared/weights",
                 "TransformerEncoder/layer_0/self_attention_prepost_wrapper/self_attention/qkv_transform/kernel",
                 "TransformerDecoder/layer_0/encdec_attention_prepost_wrapper/encdec_attention/kv_transform/kernel",
                 "TransformerDecoder/layer_0/ffn_prepost_wrapper/ffn/dense1/bias", "TransformerDecoder/output_ln/beta"]:
        flat = W[name].reshape(-1)
        for _ in range(3):
            i = int(torch.randint(0, flat.numel(), (1,), generator=rng))
            h = 1e-6
            Wp = {k: v.clone() for k, v in W.items()}
            Wp[name].reshape(-1)[i] += h
            Wm = {k: v.clone() for k, v in W.items()}
            Wm[name].reshape(-1)[i] -= h
            fd = (f(Wp) - f(Wm)) / (2 * h)
            an = float(grads[name].reshape(-1)[i])
            assert abs(fd - an) < 1e-6 + 1e-4 * abs(an), (name, i, fd, an)


def test_keras_adam_first_step_and_epsilon_placement():
    p, g = torch.tensor([1.0, -2.0]), torch.tensor([0.5, -0.25])
    p1, m1, v1 = O.keras_adam_step(p, g, torch.zeros(2), torch.zeros(2), 1, 0.1, 0.9, 0.98, 1e-9)
    # t=1: m = 0.1 g, v = 0.02 g^2, lr_t = lr*sqrt(0.02)/0.1 -> update = lr * sign(g) (epsilon negligible)
    assert torch.allclose(p1, p - 0.1 * torch.sign(g), atol=1e-6)
    # epsilon is OUTSIDE the bias correction: with a huge epsilon the step is lr_t*m/eps
    p2, _, _ = O.keras_adam_step(p, g, torch.zeros(2), torch.zeros(2), 1, 0.1, 0.9, 0.98, 1e3)
    lr_t = 0.1 * math.sqrt(1 - 0.98) / (1 - 0.9)
    assert torch.allclose(p2, p - lr_t * (0.1 * g) / (torch.sqrt(0.02 * g * g) + 1e3), atol=1e-9)


def test_noam_schedule_values():
    # speech_transformer_s schedule: factor 3.5 -> 1.5 between steps 50k and 100k, warmup 25k, d=256
    kw = dict(dmodel=256, warmup_steps=25000, initial_factor=3.5, end_factor=1.5, start_decay_at=50000, decay_steps=50000)
    assert abs(O.noam_lr(0, **kw) - 3.5 * 256 ** -0.5 * (1 / 25000) / math.sqrt(25000)) < 1e-12
    assert abs(O.noam_lr(24999, **kw) - 3.5 * 256 ** -0.5 / math.sqrt(25000)) < 1e-12
    assert abs(O.noam_lr(74999, **kw) - 2.5 * 256 ** -0.5 / math.sqrt(75000)) < 1e-12
    assert abs(O.noam_lr(199999, **kw) - 1.5 * 256 ** -0.5 / math.sqrt(200000)) < 1e-12


def test_dp_average_equals_single_process_when_token_counts_match():
    """hvd.Average of per-rank token-mean gradients == gradient of the concatenated batch iff every rank has the same
    number of target tokens; otherwise it is the documented mean-of-means."""
    cfg = {"num_enc": 1, "num_dec": 1, "num_heads": 2, "layer_norm": True, "d_model": 8, "channels": 3, "ffn": 6}
    V, T, F, L = 7, 9, 8, 3
    W = O.init_speech_transformer_weights(cfg, V, F, 1, seed=1, dtype=torch.float64)
    g = torch.Generator().manual_seed(4)

    def batch(B, tl):
        trg = torch.randint(0, V - 3, (B, L), generator=g)
        return {"src": torch.randn(B, T, F, 1, generator=g, dtype=torch.float64), "src_length": torch.full((B,), T),
                "trg": trg, "trg_length": torch.tensor(tl), "trg_input": torch.cat([torch.full((B, 1), V - 2), trg[:, :-1]], 1)}

    b0, b1 = batch(2, [3, 2]), batch(2, [2, 3])       # 5 tokens each
    cat = {k: torch.cat([b0[k], b1[k]]) for k in b0}
    _, _, g0 = O.train_step_reference(W, b0, cfg, 0.1)
    _, _, g1 = O.train_step_reference(W, b1, cfg, 0.1)
    _, _, gc = O.train_step_reference(W, cat, cfg, 0.1)
    names = list(W)
    avg = O.average_gradients([[g0[n] for n in names], [g1[n] for n in names]])
    for n, a in zip(names, avg):
        assert torch.allclose(a, gc[n], atol=1e-10), n
    b2 = batch(2, [1, 1])                               # 2 tokens: mean of means != global mean
    cat2 = {k: torch.cat([b0[k], b2[k]]) for k in b0}
    _, _, g2 = O.train_step_reference(W, b2, cfg, 0.1)
    _, _, gc2 = O.train_step_reference(W, cat2, cfg, 0.1)
    n = "target_symbol_modality/shared/bias"
    assert not torch.allclose((g0[n] + g2[n]) / 2, gc2[n], atol=1e-6)
    assert torch.allclose((5 * g0[n] + 2 * g2[n]) / 7, gc2[n], atol=1e-10)


def test_clip_gradients_known_answers():
    """tf.clip_by_value / tf.clip_by_norm semantics per tensor (gradaccum_keras_model.py:228-233)."""
    g = {"a": torch.tensor([3.0, -4.0]), "b": torch.tensor([0.3, 0.4])}
    byv = O.clip_gradients(g, clip_value=1.0)
    assert byv["a"].tolist() == [1.0, -1.0] and byv["b"].tolist() == pytest.approx([0.3, 0.4])
    byn = O.clip_gradients(g, clip_norm=1.0)
    assert byn["a"].tolist() == pytest.approx([0.6, -0.8]) and byn["b"].tolist() == pytest.approx([0.3, 0.4])   # ||b|| = 0.5 < 1
    assert O.clip_gradients(g)["a"] is g["a"]


def test_philox4x32_matches_random123_known_answers():
    """oracle/philox.py against Random123's published known-answer vectors (kat_vectors: philox4x32 with 10 and 7 rounds);
    the kernels run the 7-round variant of the same round function (nst_common.h)."""
    from oracle import philox as P

    def run(ctr, key, rounds):
        out = P.philox4x32([ctr[0]], [ctr[1]], [ctr[2]], [ctr[3]], key[0], key[1], rounds=rounds)
        return [int(w[0]) for w in out]
    assert run((0, 0, 0, 0), (0, 0), 10) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run((0xffffffff,) * 4, (0xffffffff,) * 2, 10) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), 10) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert run((0, 0, 0, 0), (0, 0), 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    # 16-bit fields: element idx -> word (idx % 8) // 2 of group idx // 8, low half first; threshold semantics
    f = P.fields16(seed=0, stream=0, n=8)
    assert [int(x) for x in f] == [0xb709, 0x5f6f, 0x3f64, 0x0d89, 0x1f81, 0x4f12, 0x0a48, 0x4f73]
    assert P.dropout_params16(0.25) == (16384, 65536.0 / 49152.0) and P.dropout_params16(0.0)[0] == 0
    k = P.keep_multiplier(123, 9, 1 << 16, 0.25)
    assert abs(float((k > 0).mean()) - 0.75) < 0.01 and set(np.unique(k)) == {0.0, 65536.0 / 49152.0}
