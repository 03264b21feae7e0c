"""Byte-pair-encoding sub-tokenizer (the `bpe` subtokenizer of neurst/data/text/bpe.py, i.e. Sennrich's subword-nmt
`apply_bpe`, codes file version 0.1 / 0.2): a word is split into characters (the last one carrying the end-of-word mark
`</w>`), the adjacent pair with the best rank in the merge table is joined until no listed pair is left, and all pieces
but the last get the separator `@@`.

Pinned on the reference's own data: re-applying tests/examples/codes.bpe4k.{en,zh} to the de-BPE'd lines of
tests/examples/train.example.*.bpe.txt reproduces all 7595 + 7595 lines (checked when the fixture is generated,
tests/golden/make_golden_data.py::gen_bpe; the committed fixture holds the merge table and a sample of lines).
"""


class BPE(object):
    def __init__(self, codes=None, separator="@@", glossaries=None):
        self.separator = separator
        self.glossaries = list(glossaries or [])
        self.version = (0, 1)
        self.ranks = {}
        self._cache = {}
        if codes is not None:
            self.init_subtokenizer(codes)

    def init_subtokenizer(self, codes):
        """codes: path of a subword-nmt codes file, or its lines."""
        if isinstance(codes, str):
            with open(codes, encoding="utf-8") as fp:
                lines = fp.read().split("\n")
        else:
            lines = list(codes)
        if lines and lines[0].startswith("#version"):
            self.version = tuple(int(x) for x in lines[0].split(":")[1].strip().split("."))
            lines = lines[1:]
        merges = [tuple(l.rstrip("\r").split(" ")) for l in lines if l.strip()]
        # later duplicates never win: the FIRST occurrence of a pair defines its rank (subword-nmt keeps reversed(enumerate))
        self.ranks = {}
        for i, pair in enumerate(merges):
            if len(pair) == 2 and pair not in self.ranks:
                self.ranks[pair] = i
        self._cache = {}
        return self

    def _encode_word(self, word):
        if word in self._cache:
            return self._cache[word]
        if self.version == (0, 1):
            symbols = list(word) + ["</w>"]
        else:
            symbols = list(word[:-1]) + [word[-1] + "</w>"]
        while len(symbols) > 1:
            best, best_rank = None, None
            for pair in zip(symbols[:-1], symbols[1:]):
                r = self.ranks.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            first, second = best
            out, i = [], 0
            while i < len(symbols):
                if i < len(symbols) - 1 and symbols[i] == first and symbols[i + 1] == second:
                    out.append(first + second)
                    i += 2
                else:
                    out.append(symbols[i])
                    i += 1
            symbols = out
        if symbols[-1] == "</w>":
            symbols = symbols[:-1]
        elif symbols[-1].endswith("</w>"):
            symbols[-1] = symbols[-1][:-4]
        self._cache[word] = symbols
        return symbols

    def tokenize(self, text, return_str=False):
        words = text.split() if isinstance(text, str) else list(text)
        out = []
        for w in words:
            if w in self.glossaries:
                out.append(w)
                continue
            pieces = self._encode_word(w)
            out.extend(p + self.separator for p in pieces[:-1])
            out.append(pieces[-1])
        return " ".join(out) if return_str else out

    def detokenize(self, text, return_str=True):
        s = text if isinstance(text, str) else " ".join(text)
        s = s.replace(self.separator + " ", "")
        if s.endswith(self.separator):
            s = s[:-len(self.separator)]
        return s if return_str else s.split()
