"""Vocabulary and the text data pipeline of the target side (neurst/data/text/vocab.py:27-160,
neurst/data/data_pipelines/text_data_pipeline.py:25-160).

`<UNK>`, `<SEQ_BEG>`, `<SEQ_END>` are appended behind the vocabulary file's tokens, padding is EOS
(meta: pad_id == eos_id, PaddingMode.EOS_AS_PADDING).  encode() = whitespace split of an already tokenised /
sub-tokenised line -> ids (+ EOS); the moses / jieba / BPE / sentencepiece tokenisers of the reference are outside
the training hot path's scope (SURVEY §8(f) rank 2 is the on-disk formats): a pipeline configured with one raises for
unprocessed text instead of silently skipping it.
"""
import json

from neurst_amd.data.bpe import BPE
from neurst_amd.utils import compat


class SentencePiece(object):
    """The `spm` sub-tokenizer (neurst/data/text/spm.py:28-91): EncodeAsPieces / DecodePieces of the same library."""

    def __init__(self, model_path):
        import sentencepiece
        self._sp = sentencepiece.SentencePieceProcessor()
        assert self._sp.Load(model_path), "Fail to load spm model: {}".format(model_path)

    def tokenize(self, text, return_str=False):
        pieces = self._sp.EncodeAsPieces(text if isinstance(text, str) else " ".join(text))
        return " ".join(pieces) if return_str else pieces

    def detokenize(self, text, return_str=True):
        out = self._sp.DecodePieces(text.split() if isinstance(text, str) else list(text))
        return out if return_str else out.split()


class Vocab(object):
    def __init__(self, tokens, extra_tokens=None, lowercase=False):
        assert isinstance(tokens, list), "`tokens` must be a list of string tokens"
        if lowercase:
            uniq = []
            for t in tokens:
                t = t.lower()
                if t not in uniq:
                    uniq.append(t)
            tokens = uniq
        self._token_list = list(tokens)
        if isinstance(extra_tokens, list):
            for t in extra_tokens:
                if t not in self._token_list:
                    self._token_list.append(t)
        self._token_to_id_dict = dict((w, i) for i, w in enumerate(self._token_list))
        self._lowercase = lowercase
        self._extra_tokens = extra_tokens

    @property
    def tokens(self):
        return self._token_list

    @property
    def vocab_size(self):
        return len(self._token_list)

    @staticmethod
    def load_tokens(vocab_path=None, tokens=None):
        """vocab.py:74-101: one token per line (first whitespace-separated field; quoted tokens keep inner spaces)."""
        skip_empty = True
        if not ((vocab_path is None) ^ (tokens is None)):
            raise ValueError("Either `vocab_path` or `tokens` should be provided.")
        if vocab_path:
            with open(vocab_path, encoding="utf-8") as f:
                if vocab_path.endswith(".json"):
                    tokens = list(json.load(f).keys())
                    skip_empty = False
                else:
                    tokens = [line.strip("\n") for line in f]
        cleaned = []
        for word in tokens:
            if len(word) > 1 and ((word.startswith("'") and word.endswith("'")) or (word.startswith('"') and word.endswith('"'))):
                word = word[1:-1]
            elif word.strip() != "" and skip_empty:
                word = word.strip().split()[0]
            if word == "" and skip_empty:
                continue
            cleaned.append(word)
        return cleaned

    @staticmethod
    def get_unique(codebook, token):
        n = 0
        while token in codebook:  # the reference appends random digits; any unused name serves
            token += str(n % 10)
            n += 1
        return token

    def map_token_to_id(self, tokens, unknown_default=None):
        def _map(t):
            if self._lowercase and t not in (self._extra_tokens or []):
                t = t.lower()
            return self._token_to_id_dict.get(t, unknown_default)
        if isinstance(tokens, list):
            return [_map(t) for t in tokens]
        assert isinstance(tokens, str)
        return _map(tokens)

    def map_id_to_token(self, ids):
        if isinstance(ids, (list, tuple)):
            return [self._token_list[int(i)] for i in ids]
        return self._token_list[int(ids)]


class TextDataPipeline(Vocab):
    def __init__(self, vocab_path, language="en", tokenizer=None, subtokenizer=None, subtokenizer_codes=None,
                 glossaries=None, reverse_sequence=False, bos_id=None, eos_id=None, unk_id=None, pad_id=None, **kwargs):
        self._config = dict(vocab_path=vocab_path, language=language, tokenizer=tokenizer, subtokenizer=subtokenizer,
                            subtokenizer_codes=subtokenizer_codes, glossaries=glossaries, reverse_sequence=reverse_sequence)
        self._language = language
        self._reverse_sequence = reverse_sequence
        self._tokenizer, self._subtokenizer = tokenizer, subtokenizer
        self._bpe = None
        if subtokenizer is not None and str(subtokenizer).lower() == "bpe":
            if subtokenizer_codes is None:
                raise ValueError("subtokenizer=bpe needs subtokenizer_codes")
            self._bpe = BPE(subtokenizer_codes, glossaries=glossaries)
        elif subtokenizer is not None and str(subtokenizer).lower() in ("spm", "sentencepiece"):
            if subtokenizer_codes is None:
                raise ValueError("subtokenizer=spm needs subtokenizer_codes (the SentencePiece model file)")
            self._bpe = SentencePiece(subtokenizer_codes)
        tokens = Vocab.load_tokens(tokens=vocab_path) if isinstance(vocab_path, list) else Vocab.load_tokens(vocab_path=vocab_path)
        unk_token = Vocab.get_unique(tokens, "<UNK>") if unk_id is None else tokens[unk_id]
        bos_token = Vocab.get_unique(tokens, "<SEQ_BEG>") if bos_id is None else tokens[bos_id]
        eos_token = Vocab.get_unique(tokens, "<SEQ_END>") if eos_id is None else tokens[eos_id]
        pad_token = eos_token if pad_id is None else tokens[pad_id]
        assert unk_token != bos_token != eos_token
        Vocab.__init__(self, tokens, [unk_token, bos_token, eos_token, pad_token], lowercase=False)
        self._eos_id = self.map_token_to_id(eos_token)
        self._bos_id = self.map_token_to_id(bos_token)
        self._unk_id = self.map_token_to_id(unk_token)
        self._pad_id = self.map_token_to_id(pad_token)

    def get_config(self):
        return dict(self._config)

    @property
    def meta(self):
        return {"language": self._language, "vocab_size": self.vocab_size, "eos_id": self._eos_id, "bos_id": self._bos_id,
                "unk_id": self._unk_id, "pad_id": self._eos_id,
                "padding_mode": compat.PaddingMode.EOS_AS_PADDING if self._eos_id == self._pad_id else compat.PaddingMode.DEFAULT}

    def preprocess(self, text):
        """text_data_pipeline.py:93-99: tokenizer, then sub-tokenizer.  Built: no / whitespace tokenizer, BPE and SentencePiece
        sub-tokenizers; anything else (moses, jieba ...) raises instead of being skipped."""
        if self._tokenizer and str(self._tokenizer).lower() not in ("none", "space", "whitespace"):
            raise NotImplementedError(f"tokenizer={self._tokenizer} is not built: pass tokenised text or projected ids")
        if self._subtokenizer and self._bpe is None and str(self._subtokenizer).lower() != "none":
            raise NotImplementedError(f"subtokenizer={self._subtokenizer} is not built (only bpe and spm)")
        if self._bpe is not None:
            text = self._bpe.tokenize(text, return_str=True)
        return text

    def postprocess(self, text):
        return self._bpe.detokenize(text) if self._bpe is not None else text

    def encode(self, text, is_processed=False):
        """text_data_pipeline.py:109-125."""
        if not is_processed:
            text = self.preprocess(text)
        if isinstance(text, str):
            text = text.split()
        ids = self.map_token_to_id(text, unknown_default=self._unk_id)
        if self._reverse_sequence:
            ids = ids[::-1]
        return ids + [self._eos_id]

    def decode(self, ids):
        """text_data_pipeline.py:127-150 without detokenisers: strip BOS, cut at the first EOS, join the tokens."""
        ids = [int(x) for x in ids]
        if ids and ids[0] == self._bos_id:
            ids = ids[1:]
        if self._eos_id in ids:
            ids = ids[:ids.index(self._eos_id)]
        toks = self.map_id_to_token(ids)
        if self._reverse_sequence:
            toks = toks[::-1]
        return self.postprocess(" ".join(toks))
