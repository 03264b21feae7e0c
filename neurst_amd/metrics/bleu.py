"""BLEU (neurst/metrics/bleu.py): the reference's own n-gram counting / corpus and sentence BLEU (:32-137) and its
mteval-v13a style tokenizer `commonly_tokenize` (:291-308), restated; pinned on outputs of the reference functions
(tests/golden/metrics.json).

What differs from the reference's metric CLASS: its headline numbers come from the `sacrebleu` package and its `tok_bleu`
from the Moses tokenizer (sacremoses) -- neither is installed in this image.  Here
  tok_bleu    corpus BLEU over whitespace tokens (text is expected tokenised, e.g. the de-BPE'd pipeline output);
  detok_bleu  corpus BLEU over `commonly_tokenize`d text (13a rules) -- the reference routes this flag to sacrebleu's 13a
              BLEU, which agrees with it whenever every n-gram order has a match (sacrebleu only adds smoothing of zero counts);
  sacre_bleu / chrf are served by sacrebleu when it is importable and raise otherwise.
"""
import collections
import math
import re

from neurst_amd.metrics import register_metric
from neurst_amd.metrics.metric import Metric


def _ngrams(tokens, max_n):
    c = collections.Counter()
    for n in range(1, max_n + 1):
        for i in range(len(tokens) - n + 1):
            c[(n,) + tuple(tokens[i:i + n])] += 1
    return c


def bleu_count(hypothesis, references, max_n=4):
    """bleu.py:32-88 -> (clipped matches per order, hypothesis n-grams per order, hypothesis length, closest reference length);
    references[m] is the list of reference strings of sentence m."""
    clip, total = [0] * max_n, [0] * max_n
    len_hyp = len_ref = 0
    for hyp, refs in zip(hypothesis, references):
        x = hyp.split()
        ys = [r.split() for r in refs]
        closest = min(ys, key=lambda y: (abs(len(y) - len(x)), len(y)))   # smallest |diff|, then the shorter reference
        len_hyp += len(x)
        len_ref += len(closest)
        ref_max = collections.Counter()
        for y in ys:
            for g, n in _ngrams(y, max_n).items():
                ref_max[g] = max(ref_max[g], n)
        for g, n in _ngrams(x, max_n).items():
            total[g[0] - 1] += n
            clip[g[0] - 1] += min(n, ref_max.get(g, 0))
    return clip, total, len_hyp, len_ref


def _combine(precisions, len_hyp, len_ref, max_n):
    bp = math.exp(1 - len_ref / len_hyp) if len_hyp < len_ref else 1.0
    log_sum = sum((math.log(p) if p > 0 else -9999999999.0) for p in precisions)
    return [bp * math.exp(log_sum / float(max_n))] + list(precisions), [bp, len_hyp / len_ref, len_hyp, len_ref]


def corpus_bleu(hypothesis, references, max_n=4):
    """bleu.py:91-112: ([bleu, p1..pn], [brevity penalty, length ratio, hypothesis length, reference length])."""
    clip, total, lh, lr = bleu_count(hypothesis, references, max_n)
    return _combine([(c / t if t > 0 else 0) for c, t in zip(clip, total)], lh, lr, max_n)


def sentence_bleu(hypothesis, references, max_n=4):
    """bleu.py:115-137: one sentence, add-0.01 smoothing of every order."""
    clip, total, lh, lr = bleu_count([hypothesis], [references], max_n)
    return _combine([(c + 0.01) / (t + 0.01) for c, t in zip(clip, total)], lh, lr, max_n)


_SGML = ((r"-\n", ""), (r"\n", " "), (r"&quot;", '"'), (r"&amp;", "&"), (r"&lt;", "<"), (r"&gt;", ">"))
_RULES = ((r"([\{-~\[-` -&\(-\+:-@\/])", r" \1 "),   # punctuation
          (r"([^0-9])([\.,])", r"\1 \2 "),            # period / comma unless preceded by a digit
          (r"([\.,])([^0-9])", r" \1 \2"),            # ... unless followed by a digit
          (r"([0-9])(-)", r"\1 \2 "))                 # dash preceded by a digit


def commonly_tokenize(s):
    """bleu.py:291-308 (multi-bleu-detok.perl / mteval-v13a.pl rules)."""
    for patt, repl in _SGML:
        s = re.sub(patt, repl, s)
    s = " " + s + " "
    for patt, repl in _RULES:
        s = re.sub(patt, repl, s)
    return " ".join(s.split())


_ESCAPES = (("&amp;", "&"), ("&#124;", "|"), ("&lt;", "<"), ("&gt;", ">"), ("&apos;", "'"), ("&quot;", '"'),
            ("&#91;", "["), ("&#93;", "]"))


def unescape(s):
    """bleu.py:311-336: Moses escapes back to characters (in the reference's order)."""
    for esc, ch in _ESCAPES:
        s = s.replace(esc, ch)
    return s


@register_metric(["sacre_bleu", "tok_bleu", "detok_bleu", "chrf", "uncased_sacre_bleu", "uncased_tok_bleu",
                  "uncased_detok_bleu", "uncased_chrf", "bleu"])
class BLEU(Metric):
    def __init__(self, language="en", *args, **kwargs):
        super().__init__()
        self._language = language
        self._flag = "tok_bleu"
        self._refs = None

    @staticmethod
    def _transposed(groundtruth):
        """[refs of set 0, refs of set 1, ...] (or one flat list) -> per sentence lists."""
        if isinstance(groundtruth[0], str):
            groundtruth = [groundtruth]
        return [list(r) for r in zip(*groundtruth)]

    def set_groundtruth(self, groundtruth):
        assert isinstance(groundtruth, list)
        self._refs_for_sacre = [groundtruth] if isinstance(groundtruth[0], str) else groundtruth
        self._refs = self._transposed(groundtruth)

    def _score(self, hypo, groundtruth, lc, tok):
        refs = self._refs if groundtruth is None else self._transposed(groundtruth)
        prep = (lambda t: tok(unescape(t.lower() if lc else t)))
        try:
            return corpus_bleu([prep(h) for h in hypo], [[prep(r) for r in rr] for rr in refs])[0][0] * 100
        except (IndexError, ZeroDivisionError):
            return 0.

    def tok_bleu(self, hypo, groundtruth=None, lc=False):
        return self._score(hypo, groundtruth, lc, lambda t: t)

    def detok_bleu(self, hypo, groundtruth=None, lc=False):
        return self._score(hypo, groundtruth, lc, commonly_tokenize)

    def sacre_bleu(self, hypo, groundtruth=None, lc=False):
        import sacrebleu  # not part of this image: ImportError is the honest answer
        refs = self._refs_for_sacre if groundtruth is None else ([groundtruth] if isinstance(groundtruth[0], str) else groundtruth)
        return sacrebleu.corpus_bleu(hypo, refs, lowercase=lc, tokenize={"zh": "zh", "ja": "ja-mecab"}.get(self._language, "13a")).score

    def get_value(self, result):
        if not isinstance(result, dict):
            return float(result)
        for k in (self._flag, self._flag.lower(), "tok_bleu"):
            if k in result:
                return result[k]
        raise KeyError(self._flag)

    def call(self, hypothesis, groundtruth=None):
        res = {"tok_bleu": self.tok_bleu(hypothesis, groundtruth), "detok_bleu": self.detok_bleu(hypothesis, groundtruth),
               "uncased_tok_bleu": self.tok_bleu(hypothesis, groundtruth, lc=True),
               "uncased_detok_bleu": self.detok_bleu(hypothesis, groundtruth, lc=True)}
        try:
            res["sacre_bleu"] = self.sacre_bleu(hypothesis, groundtruth)
            res["uncased_sacre_bleu"] = self.sacre_bleu(hypothesis, groundtruth, lc=True)
        except ImportError:
            pass
        return res
