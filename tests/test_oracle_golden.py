"""Pins the CPU oracle against the reference's golden vectors (tests/golden/*.npz,
extracted by tests/golden/make_golden.py from the reference's own test files) and
against outputs of the reference's neurst_pt front-end.  Tolerance is the
reference's own: sum of squared differences < 1e-9 (literal vectors)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import neurst_oracle as O


def _ssd(a, b):
    return float(((a.detach().numpy().astype(np.float64) - b.astype(np.float64)) ** 2).sum())


def test_mha_cross_golden():
    # tests/neurst/layers/attentions/multi_head_attention_test.py:7-60
    r, W = load_golden("mha_cross")
    W = {"a/" + k: v for k, v in W.items()}
    out = O.cross_attention(torch.from_numpy(r["query"]), torch.from_numpy(r["memory"]), W, "a",
                            int(r["num_heads"]), None)
    assert _ssd(out, r["expected"]) < 1e-9


def test_mha_self_golden():
    # tests/neurst/layers/attentions/multi_head_attention_test.py:63-111
    r, W = load_golden("mha_self")
    W = {"a/" + k: v for k, v in W.items()}
    out = O.self_attention(torch.from_numpy(r["query"]), W, "a", int(r["num_heads"]), torch.from_numpy(r["bias"]))
    assert _ssd(out, r["expected"]) < 1e-9


def test_encoder_golden():
    # tests/neurst/layers/encoders/transformer_encoder_test.py:21-122
    r, W = load_golden("transformer_encoder")
    W = O.fill_default_biases(W)
    out = O.transformer_encoder(torch.from_numpy(r["inputs"]), torch.from_numpy(r["input_padding"]), W,
                                "TransformerEncoder", int(r["num_layers"]), int(r["num_heads"]))
    assert _ssd(out, r["expected"]) < 1e-9


def test_decoder_golden():
    # tests/neurst/layers/decoders/transformer_decoder_test.py:20-158
    r, W = load_golden("transformer_decoder")
    W = O.fill_default_biases(W)
    out = O.transformer_decoder(torch.from_numpy(r["decoder_inputs"]), torch.from_numpy(r["encoder_outputs"]),
                                torch.from_numpy(r["encoder_inputs_padding"]), W, "TransformerDecoder",
                                int(r["num_layers"]), int(r["num_heads"]))
    assert _ssd(out, r["expected"]) < 1e-9


def test_position_embedding_golden():
    # tests/neurst/layers/common_layers_test.py:96-152
    r, _ = load_golden("position_embedding")
    table = torch.from_numpy(r["table"])
    out2d = O.position_embedding(O.word_embedding(torch.from_numpy(r["inputs2d"]), table))
    assert _ssd(out2d, r["expected_2d"]) < 1e-9
    out1d = O.position_embedding(O.word_embedding(torch.from_numpy(r["inputs1d"]), table), time=3)
    assert _ssd(out1d, r["expected_1d_time3"]) < 1e-9


def test_full_transformer_logits_golden():
    # tests/neurst/models/transformer_test.py:23-666 (2+2 layers, d=8, H=2, ffn=10)
    r, W = load_golden("transformer_toy_logits")
    W = O.fill_default_biases(W)
    inputs = {"src": torch.from_numpy(r["src"]), "src_padding": torch.from_numpy(r["src_padding"]),
              "trg_input": torch.from_numpy(r["trg_input"])}
    logits = O.transformer_logits(inputs, W, {"num_enc": 2, "num_dec": 2, "num_heads": 2})
    assert _ssd(logits, r["expected"]) < 1e-9


@pytest.mark.parametrize("tag", ["frontend_ln", "frontend_noln", "frontend_ragged"])
def test_frontend_matches_reference_neurst_pt(tag):
    # outputs of the reference's own neurst_pt AudioConvSubsamplingLayer
    # (neurst_pt/layers/modalities/audio_modalities.py:22-100) run under the shim of make_golden.py;
    # tolerance of tests/neurst_pt/modalities/audio_modalities_test.py (5e-5 with LN)
    r, W = load_golden("neurst_pt_" + tag)
    out = O.audio_conv_subsample(torch.from_numpy(r["src"]), W, "input_audio_modality", bool(int(r["layer_norm"])))
    np.testing.assert_allclose(out.numpy(), r["expected"], atol=5e-5, rtol=0)


@pytest.mark.parametrize("tag", ["neurst_pt_st_1x1", "neurst_pt_st_2x2_ragged", "neurst_pt_st_2x2_postnorm_untied"])
def test_full_speech_transformer_logits_and_gradients_match_reference_neurst_pt(tag):
    """The reference's own PyTorch SpeechTransformer (neurst_pt/models/speech_transformer.py; its test pins it to the TF
    model at 5e-6, tests/neurst_pt/models/speech_transformer_test.py:157) executed under the shim of make_golden.py:
    full-model logits, and -- torch autograd over the REFERENCE's forward -- the gradient of the label-smoothed token-mean
    cross entropy w.r.t. every variable, mapped to TF names with the test's own assignment list.  Pins the oracle's
    forward AND backward of the whole encoder-decoder (ragged batch, sinusoid timing) on the reference itself; the third case
    runs the reference's post-norm wrappers (no output_ln) with untied logits (its own softmax Linear)."""
    from conftest import load_reference_pt_case
    inputs, W, cfg, logits_ref, loss_ref, grads_ref = load_reference_pt_case(tag)
    inputs = dict(inputs, src=inputs["src"].double())
    loss, logits, grads = O.train_step_reference({k: v.double() for k, v in W.items()}, inputs, cfg, 0.1)
    assert float((logits - logits_ref.double()).abs().max()) < 5e-6
    assert abs(float(loss) - loss_ref) < 1e-6
    assert set(grads_ref) == set(W)
    for n, g in grads_ref.items():
        err = float((grads[n].double() - g.double()).abs().max()) / max(float(g.abs().max()), 1e-6)
        assert err < 2e-5, (n, err)


@pytest.mark.parametrize("tag", ["neurst_pt_tr_2x2", "neurst_pt_tr_2x2_shared"])
def test_text_transformer_logits_and_gradients_match_reference_neurst_pt(tag):
    """The reference's own PyTorch text Transformer (neurst_pt/models/transformer.py, pinned to TF by
    tests/neurst_pt/models/transformer_test.py) + torch autograd: separate and shared source/target embeddings."""
    from conftest import load_reference_pt_text_case
    inputs, W, cfg, logits_ref, loss_ref, grads_ref, _ = load_reference_pt_text_case(tag)
    loss, logits, grads = O.text_train_step_reference({k: v.double() for k, v in W.items()}, inputs, cfg, 0.1)
    assert float((logits - logits_ref.double()).abs().max()) < 5e-6 and abs(float(loss) - loss_ref) < 1e-6
    assert set(grads_ref) == set(W)
    for n, g in grads_ref.items():
        err = float((grads[n].double() - g.double()).abs().max()) / max(float(g.abs().max()), 1e-6)
        assert err < 2e-5, (n, err)


@pytest.mark.parametrize("ls", [0.0, 0.1, 0.35])
def test_criterion_matches_the_reference_code(ls):
    """criterion_reference.npz: the reference's own LabelSmoothedCrossEntropy code executed over a torch-backed stand-in of
    the TensorFlow primitives it calls (make_golden.py::gen_criterion): per-sentence NLL sums, token counts, the reduced
    loss and its gradient w.r.t. the logits."""
    r, _ = load_golden("criterion_reference")
    logits = torch.from_numpy(r["logits"]).double().requires_grad_(True)
    nll, _, ntok = O.label_smoothed_cross_entropy(logits, torch.from_numpy(r["trg"]), torch.from_numpy(r["trg_length"]), ls)
    loss = O.reduce_loss(nll, ntok)
    (g,) = torch.autograd.grad(loss, logits)
    key = f"ls{ls}_length"
    np.testing.assert_allclose(nll.detach().numpy(), r[key + ":nll_sum"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ntok.numpy(), r[key + ":n_tokens"], rtol=0, atol=0)
    assert abs(float(loss.detach()) - float(r[key + ":loss"])) < 2e-6
    np.testing.assert_allclose(g.numpy(), r[key + ":dlogits"], rtol=0, atol=2e-7)


def test_attention_bias_builders_match_the_reference_functions():
    """layer_utils_reference.npz: outputs of the reference's own padding / lower-triangle / wait-k bias builders
    (layer_utils.py:19-78, run over the TensorFlow stand-in of make_golden.py).  The wait-k training bias is what the
    attention kernels implement as `causal_offset = k - 1`; the step form is the decoder's `decode_lagging` mask."""
    r, _ = load_golden("layer_utils_reference")
    for i, (m, k, q) in enumerate(r["waitk_cases"].tolist()):
        want = r[f"waitk_train_{i}"]
        assert np.array_equal(O.waitk_attention_bias(m, k, q).numpy(), want)
        ii, jj = np.arange(q)[:, None], np.arange(m)[None, :]
        assert np.array_equal(want == 0, jj <= ii + (k - 1))                       # the kernels' causal_offset form
        step = r[f"waitk_step_{i}"]
        assert np.array_equal(step == 0, np.arange(m) < k) and set(np.unique(step)) <= {0.0, np.float32(O.FLOAT_MIN)}
    for n in (1, 2, 5):
        assert np.array_equal(O.lower_triangle_attention_bias(n).numpy(), r[f"lower_triangle_{n}"])
    assert np.array_equal(O.input_padding_to_bias(torch.from_numpy(r["padding"])).numpy(), r["padding_bias"])


def test_causal_bias_matrix():
    # tests/neurst_pt/layers/layer_utils_test.py:20
    b = O.lower_triangle_attention_bias(3)[0, 0]
    assert torch.equal(b == 0, torch.tril(torch.ones(3, 3)).bool())
    assert float(b[0, 1]) == -1e9


def test_length_after_conv():
    # neurst/models/speech_transformer.py:182-183
    for l, e in [(1, 1), (2, 1), (3, 1), (4, 1), (5, 2), (11, 3), (900, 225), (899, 225), (897, 225), (896, 224)]:
        assert O.length_after_conv(l) == e
