"""neurst_amd/metrics against outputs of the reference's own pure-Python metric functions (tests/golden/metrics.json, written by
tests/golden/make_golden.py::gen_metrics from neurst/metrics/bleu.py and neurst/metrics/wer.py)."""
import json
import os

import pytest

from neurst_amd.metrics import bleu as B
from neurst_amd.metrics import build_metric
from neurst_amd.metrics import wer as W

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.json")))


def _close(a, b):
    if isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _close(x, y)
    else:
        assert a == pytest.approx(b, rel=1e-12, abs=1e-12)


def test_bleu_counts_and_scores_match_the_reference_functions():
    hyps, refs = G["hyps"], G["refs"]
    _close(B.bleu_count(hyps[:6], refs[:6]), G["bleu_count"])
    _close(B.corpus_bleu(hyps[:6], refs[:6]), G["corpus_bleu"])
    _close(B.corpus_bleu(hyps[:3], [r[:1] for r in refs[:3]]), G["corpus_bleu_single"])
    _close(B.corpus_bleu(hyps[:6], refs[:6], max_n=2), G["corpus_bleu_2gram"])
    _close([B.sentence_bleu(h, r) for h, r in zip(hyps, refs) if h], G["sentence_bleu"])


def test_tokenizer_and_unescape_match_the_reference_functions():
    texts = G["raw"] + G["hyps"]
    assert [B.commonly_tokenize(x) for x in texts] == G["commonly_tokenize"]
    assert [B.unescape(x) for x in texts] == G["unescape"]


def test_wer_alignment_counts_match_the_reference_function():
    got = [list(W._wer(r[0].split(), h.split())) for h, r in zip(G["hyps"], G["refs"])]
    got += [list(W._wer(list("kitten"), list("sitting"))), list(W._wer([], ["a"])), list(W._wer(["a", "b"], []))]
    assert got == G["wer"]


def test_metric_classes():
    m = build_metric({"metric.class": "bleu", "metric.params": {"language": "en"}})
    refs = [r[0] for r in G["refs"][:6]]
    m.set_groundtruth(refs)
    res = m(G["hyps"][:6])
    un = B.unescape   # the class un-escapes Moses entities before scoring (bleu.py:368-372)
    assert res["tok_bleu"] == pytest.approx(100 * B.corpus_bleu([un(h) for h in G["hyps"][:6]], [[un(r)] for r in refs])[0][0])
    assert m.get_value(res) == res["tok_bleu"] and m.greater_or_eq(res, {"tok_bleu": res["tok_bleu"] - 1})
    assert m(refs)["tok_bleu"] == pytest.approx(100.0) and m(refs)["detok_bleu"] == pytest.approx(100.0)
    # explicit multi-reference ground truth: [set 0, set 1]
    two = m(G["hyps"][:6], [[r[0] for r in G["refs"][:6]], [r[1] for r in G["refs"][:6]]])
    assert two["tok_bleu"] == pytest.approx(100 * B.corpus_bleu([un(h) for h in G["hyps"][:6]],
                                                                 [[un(x) for x in r] for r in G["refs"][:6]])[0][0])
    m.flag = "uncased_detok_bleu"
    assert m.get_value(res) == res["uncased_detok_bleu"]
    w = build_metric({"metric.class": "WER", "metric.params": {"language": "en"}})
    w.set_groundtruth(["The cat sat on the mat.", "Hello, world!"])
    r = w(["the cat sat on mat", "hello there world"])
    assert r["WER"] == pytest.approx(100 * 2 / 8) and r["WER-deletions"] == pytest.approx(12.5) and r["WER-insertions"] == pytest.approx(12.5)
    assert w.greater_or_eq({"WER": 10.0}, {"WER": 20.0}) and not w.greater_or_eq({"WER": 30.0}, {"WER": 20.0})
    c = build_metric({"metric.class": "cer", "metric.params": {"language": "zh"}})
    assert c(["我 爱你"], ["我爱 他"])["CER"] == pytest.approx(100 / 3)
