"""Beam search against the CPU oracle and the helper semantics the reference tests
(tests/neurst/layers/search/beam_search_test.py).  CPU only: the search is device agnostic torch code, the HIP model
step is covered by tests/test_gpu_model.py::test_incremental_decoding_*."""
import numpy as np
import pytest
import torch

from neurst_amd.layers.search.beam_search import length_penalty_term, sequence_beam_search, stack_beam_size
from oracle import beam_search_oracle as BO


def test_stack_beam_size_semantics():
    """beam_search_test.py:132-155: tile each batch entry beam_size times, entry major (tf1codebase_stack_beam_size)."""
    x1 = torch.tensor([1, 2, 3])
    assert stack_beam_size(x1, 2).tolist() == [1, 1, 2, 2, 3, 3]
    x2 = torch.arange(6).view(2, 3)
    assert stack_beam_size(x2, 3).tolist() == [[0, 1, 2]] * 3 + [[3, 4, 5]] * 3
    x3 = torch.arange(24).view(2, 3, 4)
    ref = x3.repeat(1, 2, 1).reshape(4, 3, 4)   # the TF1 formulation: tile on axis 1, then fold into the batch axis
    assert torch.equal(stack_beam_size(x3, 2), ref)
    nested = stack_beam_size({"a": x1, "b": [x2, None], "n": 5}, 2)
    assert nested["a"].shape == (6,) and nested["b"][0].shape == (4, 3) and nested["b"][1] is None and nested["n"] == 5


def test_length_penalty():
    l = torch.tensor([1, 5, 20])
    assert torch.allclose(length_penalty_term(l, 0.6), ((5.0 + l.float()) / 6.0) ** -0.6)
    assert torch.allclose(length_penalty_term(l, -1.0), 1.0 / l.float())
    assert torch.allclose(length_penalty_term(l, None), 1.0 / l.float())
    assert torch.allclose(length_penalty_term(l, 0.0), torch.ones(3))


class _ToyLM(object):
    """Deterministic 'model': the next-symbol logits depend on the sample, the step and a hash of the whole prefix.  The
    torch side carries the prefix hash in a CACHE that must be re-ordered with the beams; the oracle side recomputes it
    from the explicit prefix -- a wrong gather in the search shows up as different hypotheses."""

    def __init__(self, vocab, batch, seed, eos_boost=0.0):
        rng = np.random.RandomState(seed)
        self.table = rng.randn(batch, 64, vocab).astype(np.float32) * 2.0
        self.vocab, self.eos_boost = vocab, eos_boost

    def _logits(self, sample, state, t, eos_id):
        row = self.table[sample, (state * 7 + t * 3) % 64].copy()
        row[eos_id] += self.eos_boost * t
        return row

    def prefix_fn(self, eos_id):
        def fn(sample, prefix):
            state = 0
            for y in prefix[1:]:
                state = (state * 31 + y + 1) % 1009
            return self._logits(sample, state, len(prefix) - 1, eos_id)
        return fn

    def step_fn(self, beam, eos_id):
        def fn(ids, cache, time):
            if time > 0:
                cache["state"] = (cache["state"] * 31 + ids + 1) % 1009
            rows = [self._logits(i // beam, int(cache["state"][i]), time, eos_id) for i in range(ids.shape[0])]
            return torch.from_numpy(np.stack(rows))

        def reorder(cache, beam_ids):
            cache["state"] = cache["state"].index_select(0, beam_ids)
            return cache
        return fn, reorder


@pytest.mark.parametrize("beam,top_k,alpha,min_len,eos_boost,enable_unk", [
    (1, 1, 0.6, 0, 0.0, False), (4, 1, 0.6, 0, 0.3, False), (4, 4, 1.0, 0, 0.5, False), (3, 2, -1.0, 0, 0.4, True),
    (5, 3, 0.0, 6, 1.5, False), (2, 1, 0.6, 0, 3.0, False)])
def test_beam_search_matches_oracle(beam, top_k, alpha, min_len, eos_boost, enable_unk):
    vocab, batch, bos, eos, unk = 17, 3, 15, 16, 14
    lm = _ToyLM(vocab, batch, seed=beam * 10 + top_k, eos_boost=eos_boost)
    kw = dict(beam_size=beam, top_k=top_k, length_penalty=alpha, extra_decode_length=4, maximum_decode_length=12,
              minimum_decode_length=min_len, enable_unk=enable_unk)
    want_h, want_s = BO.beam_search(lm.prefix_fn(eos), batch, bos, eos, unk, vocab, encoder_len=5, **kw)
    fn, reorder = lm.step_fn(beam, eos)
    init = {"decoder_input": torch.full((batch,), bos), "decoder_internal_cache": {"state": torch.zeros(batch * beam, dtype=torch.int64)},
            "encoder_inputs_maxlen": 5, "eos_id": eos, "unk_id": unk}
    got_h, got_s = sequence_beam_search(fn, init, reorder_cache_fn=reorder, **kw)
    assert got_h.shape == (batch * top_k, 12)
    assert got_h.tolist() == want_h.tolist()
    assert np.allclose(got_s.numpy(), want_s, rtol=1e-5, atol=1e-5)
    # scores of a sample come out best first; nothing after the first EOS but EOS; UNK never appears unless enabled
    s = got_s.view(batch, top_k)
    assert (s[:, :-1] >= s[:, 1:]).all()
    for row in got_h.tolist():
        if eos in row:
            assert all(v == eos for v in row[row.index(eos):])
        if not enable_unk:
            assert unk not in row
    if min_len:
        assert all((row + [eos]).index(eos) >= min_len - 1 for row in got_h.tolist())
    assert (got_h[:, 9:] == eos).all()            # min(5 + 4, 12) = 9 search steps, the rest is EOS padding


# ------------------------------------------------------------------------------------------------ sampling search
def test_top_k_and_top_p_filters():
    """sampling.py:67-92 on hand-checked rows."""
    import math
    from neurst_amd.layers.search.sampling import top_k_logits, top_p_logits
    FM = -1e9
    lg = torch.tensor([[1.0, 3.0, 2.0, 0.0], [0.5, 0.5, -1.0, 4.0]])
    assert torch.equal(top_k_logits(lg, 0), lg)
    assert top_k_logits(lg, 2).tolist() == [[FM, 3.0, 2.0, FM], [0.5, 0.5, FM, 4.0]]      # ties at the k-th value stay
    assert top_k_logits(lg, 1).tolist() == [[FM, 3.0, FM, FM], [FM, FM, FM, 4.0]]
    # probabilities of row 0 in descending order: 3 -> .644, 2 -> .237, 1 -> .087, 0 -> .032
    p = torch.softmax(lg[0], -1)
    assert abs(float(p[1]) - math.exp(3) / sum(math.exp(x) for x in (1, 3, 2, 0))) < 1e-6
    assert top_p_logits(lg[:1], 0.5).tolist() == [[FM, 3.0, FM, FM]]          # .644 already reaches .5
    assert top_p_logits(lg[:1], 0.7).tolist() == [[FM, 3.0, 2.0, FM]]         # .644 < .7 <= .881
    assert top_p_logits(lg[:1], 0.9).tolist() == [[1.0, 3.0, 2.0, FM]]
    assert top_p_logits(lg[:1], 1.0).tolist()[0][:3] == [1.0, 3.0, 2.0]


def test_sampling_search_on_a_toy_language_model():
    """sequence_sampling_search (sampling.py:95-283): top_k = 1 is the greedy roll-out; UNK is never drawn; EOS is not drawn
    before the minimum length; `sample_num` continuations per input; reproducible for a seed."""
    from neurst_amd.layers.search.sampling import Sampling, sequence_sampling_search, top_k_logits
    V, eos, unk, bos = 7, 6, 4, 5
    g = torch.Generator().manual_seed(0)
    table = torch.randn(V, V, generator=g) * 2.0

    def fn(ids, cache, time):       # a first-order "language model": next-token logits depend on the previous token and time
        return table[ids] + 0.1 * time
    init = {"decoder_input": torch.tensor([bos, 1, 2]), "decoder_internal_cache": {}, "encoder_inputs_maxlen": 5, "eos_id": eos,
            "unk_id": unk}
    hyp = sequence_sampling_search(fn, init, lambda lg: top_k_logits(lg, 1), sample_num=1, extra_decode_length=3,
                                   maximum_decode_length=12, minimum_decode_length=0)
    assert hyp.shape == (3, 12)
    for b, first in enumerate((bos, 1, 2)):
        prev, t = first, 0
        while t < 8:
            lg = table[prev] + 0.1 * t
            lg[unk] = -1e9
            nxt = int(lg.argmax())
            assert int(hyp[b, t]) == nxt
            prev, t = nxt, t + 1
            if nxt == eos:
                break
    assert (hyp[:, 8:] == eos).all()                                    # 5 + 3 search steps, then EOS padding
    gen = torch.Generator().manual_seed(3)
    many = sequence_sampling_search(fn, init, lambda lg: lg, sample_num=50, extra_decode_length=3, maximum_decode_length=8,
                                    minimum_decode_length=4, generator=gen, sync_every=1)
    assert many.shape == (150, 8) and not (many == unk).any() and not (many[:, :3] == eos).any()
    assert len({tuple(r) for r in many[:50].tolist()}) > 5              # the copies of one input really differ
    again = sequence_sampling_search(fn, init, lambda lg: lg, sample_num=50, extra_decode_length=3, maximum_decode_length=8,
                                     minimum_decode_length=4, generator=torch.Generator().manual_seed(3), sync_every=1)
    assert torch.equal(many, again)
    s = Sampling({"sample_num": 2, "top_p": 0.9, "maximum_decode_length": 8, "seed": 1})
    assert s.top_k == 2
    with pytest.raises(NotImplementedError):
        Sampling({"top_p": 0.5, "top_k": 3})


def test_beam_search_matches_the_reference_search_code():
    """tests/golden/beam_search_reference.npz: hypotheses and scores of the reference's OWN sequence_beam_search
    (layers/search/beam_search.py:254-440 + layer_utils helpers, executed unmodified over the torch-backed TensorFlow
    stand-in of tests/golden/make_golden.py) on the toy language model above -- first-beam-only step 0, finished beams,
    UNK / minimum-length masks, both length penalties, top-k extraction, cache re-ordering, EOS padding."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "beam_search_reference.npz"))
    vocab, batch, bos, eos, unk, enc_len, extra, max_len = (int(v) for v in z["setup"])
    for i, (beam, top_k, alpha, min_len, eos_boost, enable_unk) in enumerate(z["configs"].tolist()):
        beam, top_k, min_len, enable_unk = int(beam), int(top_k), int(min_len), bool(enable_unk)
        lm = _ToyLM(vocab, batch, seed=beam * 10 + top_k, eos_boost=eos_boost)
        fn, reorder = lm.step_fn(beam, eos)
        init = {"decoder_input": torch.full((batch,), bos), "decoder_internal_cache": {"state": torch.zeros(batch * beam, dtype=torch.int64)},
                "encoder_inputs_maxlen": enc_len, "eos_id": eos, "unk_id": unk}
        hyp, scores = sequence_beam_search(fn, init, reorder_cache_fn=reorder, beam_size=beam, top_k=top_k, length_penalty=alpha,
                                           extra_decode_length=extra, maximum_decode_length=max_len, minimum_decode_length=min_len,
                                           enable_unk=enable_unk)
        assert hyp.tolist() == z[f"hyp_{i}"].tolist(), i
        assert np.allclose(scores.numpy(), z[f"scores_{i}"], rtol=1e-5, atol=1e-5), i
        # and the independent hypothesis-list oracle agrees with the reference as well
        want_h, want_s = BO.beam_search(lm.prefix_fn(eos), batch, bos, eos, unk, vocab, encoder_len=enc_len, beam_size=beam,
                                        top_k=top_k, length_penalty=alpha, extra_decode_length=extra, maximum_decode_length=max_len,
                                        minimum_decode_length=min_len, enable_unk=enable_unk)
        assert want_h.tolist() == z[f"hyp_{i}"].tolist() and np.allclose(want_s, z[f"scores_{i}"], rtol=1e-5, atol=1e-5)


def test_sampling_filters_match_the_reference_code():
    """tests/golden/sampling_filters_reference.npz: the reference's own top_k_logits / top_p_logits (sampling.py:67-92 over the
    TensorFlow stand-in of make_golden.py), including a row with tied logits."""
    import os
    from neurst_amd.layers.search.sampling import top_k_logits, top_p_logits
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampling_filters_reference.npz"))
    lg = torch.from_numpy(z["logits"])
    for k in (0, 1, 3, 13):
        assert np.array_equal(top_k_logits(lg, k).numpy(), z[f"top_k_{k}"]), k
    for p in (0.1, 0.5, 0.9, 0.999):
        assert np.array_equal(top_p_logits(lg, p).numpy(), z[f"top_p_{p}"]), p


def test_ensemble_beam_search_matches_the_reference_search_code():
    """Two sub-models: the step distribution is the weighted sum of their probabilities (beam_search.py:104-116); golden from
    the reference's own search code (make_golden.py::gen_beam_search)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "beam_search_reference.npz"))
    vocab, batch, bos, eos, unk, enc_len, extra, max_len = (int(v) for v in z["setup"])
    beam = 3
    lm_a, lm_b = _ToyLM(vocab, batch, seed=101, eos_boost=0.4), _ToyLM(vocab, batch, seed=202, eos_boost=0.2)
    (fa, ra), (fb, rb) = lm_a.step_fn(beam, eos), lm_b.step_fn(beam, eos)

    def fn(ids, cache, time):
        return [fa(ids, cache["a"], time), fb(ids, cache["b"], time)]

    def reorder(cache, beam_ids):
        return {"a": ra(cache["a"], beam_ids), "b": rb(cache["b"], beam_ids)}
    init = {"decoder_input": torch.full((batch,), bos), "encoder_inputs_maxlen": enc_len, "eos_id": eos, "unk_id": unk,
            "decoder_internal_cache": {"a": {"state": torch.zeros(batch * beam, dtype=torch.int64)},
                                       "b": {"state": torch.zeros(batch * beam, dtype=torch.int64)}}}
    hyp, scores = sequence_beam_search(fn, init, reorder_cache_fn=reorder, beam_size=beam, top_k=2, length_penalty=0.6,
                                       extra_decode_length=extra, maximum_decode_length=max_len,
                                       ensemble_weights=z["ens_weights"].tolist())
    assert hyp.tolist() == z["ens_hyp"].tolist()
    assert np.allclose(scores.numpy(), z["ens_scores"], rtol=1e-5, atol=1e-5)
