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
