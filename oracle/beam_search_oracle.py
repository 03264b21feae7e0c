"""CPU restatement of the reference's beam search, for tests only (neurst/layers/search/beam_search.py).

TEST INFRASTRUCTURE: only tests/ may import this.  PARITY UNPINNED: the reference's own tests cover the helper functions
(stack_beam_size, one_entry_bias; tests/neurst/layers/search/beam_search_test.py) but hold no golden search output, and
TensorFlow cannot run here; this file restates the algorithm hypothesis by hypothesis, with explicit Python lists and a
STATELESS scoring function (full prefix in, next-symbol logits out), so that it shares neither the tensor formulation nor
the cache handling of neurst_amd/layers/search/beam_search.py.
"""
import math

import numpy as np

FLOAT_MIN = -1.e9


def _log_softmax(x):
    x = x.astype(np.float32)
    m = x.max()
    return (x - m - np.log(np.exp(x - m).sum(dtype=np.float32))).astype(np.float32)


def _penalty(length, alpha):
    """beam_search.py:24-41."""
    if alpha is None or alpha < 0.0:
        return np.float32(1.0) / np.float32(length)
    return np.float32(((5.0 + np.float32(length)) / 6.0) ** (-alpha))


def beam_search(prefix_logits_fn, batch, bos_id, eos_id, unk_id, vocab, beam_size=4, top_k=1, length_penalty=0.6,
                extra_decode_length=50, maximum_decode_length=256, minimum_decode_length=0, encoder_len=None, enable_unk=False):
    """prefix_logits_fn(sample index, [bos, y1, ..., yt]) -> float array [vocab] (logits of the next symbol).
    Returns (hypotheses [batch * top_k, maximum_decode_length] padded with EOS, scores [batch * top_k])."""
    max_steps = maximum_decode_length if encoder_len is None else min(encoder_len + extra_decode_length, maximum_decode_length)
    max_steps = max(max_steps, minimum_decode_length)
    # beam_search.py:300-322: every sample starts with beam_size copies of (BOS, log prob 0, length 0, unfinished)
    beams = [[{"ids": [], "lp": np.float32(0), "len": 0, "fin": False} for _ in range(beam_size)] for _ in range(batch)]
    steps = 0
    while steps < max_steps and not all(h["fin"] for hs in beams for h in hs):
        new_beams = []
        for s in range(batch):
            cands = []
            for b, h in enumerate(beams[s]):
                if steps == 0 and b > 0:
                    break  # :185-189: at time 0 only the first beam's distribution is used
                lp = _log_softmax(np.asarray(prefix_logits_fn(s, [bos_id] + h["ids"])))
                if h["fin"]:  # :117-130: a finished beam continues with EOS at no cost, nothing else
                    lp = np.full(vocab, FLOAT_MIN, np.float32)
                    lp[eos_id] = 0.0
                if unk_id is not None and not enable_unk:  # :133-140
                    lp[unk_id] += np.float32(FLOAT_MIN)
                if steps < minimum_decode_length - 1:  # :381-389
                    lp[eos_id] += np.float32(FLOAT_MIN)
                total = lp + h["lp"]
                nlen = h["len"] + 1 - int(h["fin"])
                score = total * _penalty(nlen, length_penalty)
                for v in range(vocab):
                    cands.append((float(score[v]), b, v, total[v], nlen))
            # :192 top_k over the flattened [beam * vocab] scores; ties resolve to the lower flat index
            cands.sort(key=lambda c: (-c[0], c[1] * vocab + c[2]))
            chosen = cands[:beam_size]
            new_beams.append([{"ids": beams[s][b]["ids"] + [v], "lp": np.float32(t), "len": n, "fin": v == eos_id}
                              for _, b, v, t, n in chosen])
        beams = new_beams
        steps += 1
    hyps, scores = [], []
    for s in range(batch):  # :218-251
        ranked = sorted(range(beam_size), key=lambda b: (-float(beams[s][b]["lp"] * _penalty(beams[s][b]["len"], length_penalty)), b))
        for b in ranked[:top_k]:
            h = beams[s][b]
            hyps.append(h["ids"] + [eos_id] * (maximum_decode_length - len(h["ids"])))
            scores.append(float(h["lp"] * _penalty(h["len"], length_penalty)))
    return np.asarray(hyps, dtype=np.int64), np.asarray(scores, dtype=np.float32)
