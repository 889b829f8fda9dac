"""Host-side BERT tokeniser (uncased basic tokenisation + greedy WordPiece).

The reference gets this from tokenizers==0.22.1 via sentence-transformers (rag/embedding.py:33,65;
requirements.txt:148); the algorithm restated here is the published BERT one: clean -> lower-case
+ accent strip -> whitespace / punctuation split -> longest-match-first WordPiece with '##'
continuations, words longer than 100 chars -> [UNK], then [CLS] ... [SEP] with truncation to the
model's max_seq_length.  ``HashTokenizer`` stands in when no vocab.txt exists (synthetic weights):
same splitting, ids from a stable hash, so plumbing and benchmarks run fully offline.
"""
from __future__ import annotations

import os
import re
import unicodedata
import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or
            0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or
            0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


_ASCII_CTRL = {c: None for c in list(range(0, 9)) + [11, 12] + list(range(14, 32)) + [127]}   # Cc except \t \n \r
_ASCII_TOKEN = re.compile(r"[0-9A-Za-z]+|[!-/:-@\[-`{-~]")


def basic_tokenize(text: str, lower: bool = True, strip_accents=None) -> List[str]:
    """BERT basic tokenisation.  `strip_accents=None` follows the lower-casing flag (the BertNormalizer rule)."""
    if text.isascii():
        # same result as the general path below for 7-bit text (no accents, no CJK, every printable non-alphanumeric
        # character is punctuation and stands alone), two orders of magnitude faster: the MMR step re-tokenises the
        # retrieved page-long chunks on every query (reference rag/retrieval.py:238-239)
        text = text.translate(_ASCII_CTRL)
        return _ASCII_TOKEN.findall(text.lower() if lower else text)
    strip = lower if strip_accents is None else bool(strip_accents)
    cleaned = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch) in ("Cc", "Cf") and ch not in "\t\n\r"):
            continue
        if _is_cjk(cp):
            cleaned.append(f" {ch} ")
        elif ch in " \t\n\r" or unicodedata.category(ch) == "Zs":
            cleaned.append(" ")
        else:
            cleaned.append(ch)
    words: List[str] = []
    for tok in "".join(cleaned).split():
        if lower:
            tok = tok.lower()
        if strip:
            tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
        cur = ""
        for ch in tok:
            if _is_punct(ch):
                if cur:
                    words.append(cur)
                    cur = ""
                words.append(ch)
            else:
                cur += ch
        if cur:
            words.append(cur)
    return words


class WordPieceTokenizer:
    def __init__(self, vocab: Dict[str, int], lower: bool = True, unk="[UNK]", cls="[CLS]", sep="[SEP]", pad="[PAD]",
                 strip_accents=None):
        self.vocab, self.lower, self.strip_accents = vocab, lower, strip_accents
        self.unk_id, self.cls_id, self.sep_id = vocab[unk], vocab[cls], vocab[sep]
        self.pad_id = vocab.get(pad, 0)

    @classmethod
    def from_vocab_file(cls, path: str, lower: bool = True, strip_accents=None) -> "WordPieceTokenizer":
        with open(path, encoding="utf-8") as fh:
            vocab = {line.rstrip("\n"): i for i, line in enumerate(fh)}
        return cls(vocab, lower=lower, strip_accents=strip_accents)

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > 100:
            return [self.unk_id]
        out, start = [], 0
        while start < len(word):
            end, found = len(word), None
            while start < end:
                piece = ("##" if start else "") + word[start:end]
                if piece in self.vocab:
                    found = self.vocab[piece]
                    break
                end -= 1
            if found is None:
                return [self.unk_id]
            out.append(found)
            start = end
        return out

    def encode(self, text: str, max_len: int) -> List[int]:
        ids: List[int] = []
        for w in basic_tokenize(text, self.lower, self.strip_accents):
            ids.extend(self._wordpiece(w))
            if len(ids) >= max_len - 2:
                break
        return [self.cls_id] + ids[: max_len - 2] + [self.sep_id]


class FastWordPieceTokenizer:
    """The same tokenisation through the `tokenizers` library (the reference's own backend: requirements.txt:148,
    via sentence-transformers) when it is importable: BertNormalizer + BertPreTokenizer + WordPiece, or the model
    directory's tokenizer.json as shipped.  Two orders of magnitude faster than the restatement above, which
    matters on the index-build side (the encoder takes ~17 M tokens/s).  `WordPieceTokenizer` stays the fallback
    and the specification; tests/test_host_cpu.py holds the two to identical ids."""

    def __init__(self, tok, cls_id: int, sep_id: int, pad_id: int):
        self._tok, self.cls_id, self.sep_id, self.pad_id = tok, cls_id, sep_id, pad_id
        self._max_len = None

    @classmethod
    def from_tokenizer_json(cls, path: str) -> "FastWordPieceTokenizer":
        """The model directory's own tokenizer.json, exactly as the reference's backend loads it (normaliser,
        casing and accent rules included); padding off, truncation set per call."""
        from tokenizers import Tokenizer
        tok = Tokenizer.from_file(path)
        tok.no_padding()
        ids = [tok.token_to_id(t) for t in ("[CLS]", "[SEP]", "[PAD]")]
        if ids[0] is None or ids[1] is None:
            raise ValueError(f"{path}: not a BERT-style tokenizer ([CLS]/[SEP] missing)")
        return cls(tok, ids[0], ids[1], ids[2] if ids[2] is not None else 0)

    @classmethod
    def from_vocab(cls, vocab: Dict[str, int], lower: bool = True, unk="[UNK]", cls_tok="[CLS]", sep="[SEP]", pad="[PAD]",
                   strip_accents=None):
        from tokenizers import Tokenizer
        from tokenizers.models import WordPiece
        from tokenizers.normalizers import BertNormalizer
        from tokenizers.pre_tokenizers import BertPreTokenizer
        from tokenizers.processors import TemplateProcessing
        tok = Tokenizer(WordPiece(vocab, unk_token=unk, max_input_chars_per_word=100))
        tok.normalizer = BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=strip_accents, lowercase=lower)
        tok.pre_tokenizer = BertPreTokenizer()
        tok.post_processor = TemplateProcessing(single=f"{cls_tok} $A {sep}",
                                                special_tokens=[(cls_tok, vocab[cls_tok]), (sep, vocab[sep])])
        return cls(tok, vocab[cls_tok], vocab[sep], vocab.get(pad, 0))

    @classmethod
    def from_vocab_file(cls, path: str, lower: bool = True, strip_accents=None) -> "FastWordPieceTokenizer":
        with open(path, encoding="utf-8") as fh:
            vocab = {line.rstrip("\n"): i for i, line in enumerate(fh)}
        return cls.from_vocab(vocab, lower=lower, strip_accents=strip_accents)

    def _limit(self, max_len: int):
        if self._max_len != max_len:
            self._tok.enable_truncation(max_length=max_len)
            self._max_len = max_len

    def encode(self, text: str, max_len: int) -> List[int]:
        self._limit(max_len)
        return self._tok.encode(text).ids

    def encode_batch(self, texts: Sequence[str], max_len: int) -> List[List[int]]:
        self._limit(max_len)
        return [e.ids for e in self._tok.encode_batch(list(texts))]


def make_wordpiece_tokenizer(vocab_path: str, lower: bool = True, strip_accents=None):
    """FastWordPieceTokenizer when the `tokenizers` library is importable (CRS_TOKENIZER=python forces the pure
    Python restatement), else WordPieceTokenizer."""
    if os.environ.get("CRS_TOKENIZER", "") != "python":
        try:
            return FastWordPieceTokenizer.from_vocab_file(vocab_path, lower=lower, strip_accents=strip_accents)
        except ImportError:
            pass
    return WordPieceTokenizer.from_vocab_file(vocab_path, lower=lower, strip_accents=strip_accents)


def tokenizer_from_model_dir(path: str):
    """The tokeniser a HuggingFace BERT checkpoint directory describes.  Casing comes from the TOKENIZER's own
    files, as in the reference stack (sentence-transformers hands the text to the HF tokenizer; the published
    all-MiniLM-L6-v2 ships sentence_bert_config.json do_lower_case=false next to an uncased vocab whose tokenizer
    lower-cases): tokenizer.json when the `tokenizers` library can load it, else vocab.txt with
    tokenizer_config.json's do_lower_case / strip_accents (absent -> lower-case, the uncased default)."""
    import json
    tj = os.path.join(path, "tokenizer.json")
    if os.path.exists(tj) and os.environ.get("CRS_TOKENIZER", "") != "python":
        try:
            return FastWordPieceTokenizer.from_tokenizer_json(tj)
        except ImportError:
            pass
    lower, strip = True, None
    tc = os.path.join(path, "tokenizer_config.json")
    if os.path.exists(tc):
        with open(tc) as fh:
            cfg = json.load(fh)
        lower = bool(cfg.get("do_lower_case", True))
        strip = cfg.get("strip_accents", None)
    return make_wordpiece_tokenizer(os.path.join(path, "vocab.txt"), lower=lower, strip_accents=strip)


class HashTokenizer:
    """vocab-free stand-in: one id per basic token, crc32 into [1000, vocab)."""

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size
        self.cls_id, self.sep_id, self.pad_id = (101, 102, 0) if vocab_size > 1000 else (1, 2, 0)
        self.lo = 1000 if vocab_size > 2000 else 3

    def encode(self, text: str, max_len: int) -> List[int]:
        span = self.vocab_size - self.lo
        ids = [self.lo + zlib.crc32(w.encode("utf-8")) % span for w in basic_tokenize(text)][: max_len - 2]
        return [self.cls_id] + ids + [self.sep_id]


def pad_batch(seqs: Sequence[Sequence[int]], pad_id: int = 0, short_steps: Sequence[int] = ()) -> Tuple[np.ndarray, np.ndarray]:
    """Right-pad to the longest sequence of the batch -> (ids int32 [B, S], lens int32 [B]).
    `short_steps` (ascending, e.g. (16, 32, 64)): a batch whose longest sequence fits one of them is padded
    up to it -- the fused short-sequence encoder kernels exist for exactly those lengths; padding is masked
    by `lens`, so the embeddings do not change."""
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    width = int(lens.max())
    for step in short_steps:
        if width <= step:
            width = step
            break
    out = np.full((len(seqs), width), pad_id, dtype=np.int32)
    for r, s in enumerate(seqs):
        out[r, : len(s)] = s
    return out, lens
