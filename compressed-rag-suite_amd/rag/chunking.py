"""Chunk record + text chunkers (host side, CPU string work).

Mirrors the public surface of /root/reference/rag/chunking.py (``Chunk`` :24-33, ``TextChunker``
:36-243: strategies 'semantic' / 'sentence' / 'fixed', ids ``chunk_{n}``, ``reset_chunk_ids``).
Out of the accelerated hot path (only the ``Chunk`` record shape matters to
it); nltk is optional here -- the reference's own fallback splitter (:158-161) is used without it.
"""
from __future__ import annotations

import logging
import re
from dataclasses import dataclass
from typing import List, Optional

logger = logging.getLogger(__name__)


@dataclass
class Chunk:
    """A text chunk with its provenance."""
    text: str
    chunk_id: str
    start_char: int
    end_char: int
    page_number: Optional[int] = None
    section: Optional[str] = None
    tokens: Optional[int] = None


def _load_punkt():
    try:  # optional: exact sentence boundaries when nltk + punkt data are installed
        import nltk
        return nltk.data.load('tokenizers/punkt/english.pickle')
    except Exception:
        return None


class TextChunker:
    """Splits cleaned page text into ``Chunk`` records."""

    def __init__(self, config: dict):
        self.strategy = config.get('strategy', 'semantic')
        self.chunk_size = config.get('chunk_size', 512)
        self.chunk_overlap = config.get('chunk_overlap', 50)
        self.min_chunk_size = config.get('min_chunk_size', 100)
        self.sent_tokenizer = _load_punkt()
        self._global_chunk_id = 0
        logger.info(f"Initialized TextChunker with strategy: {self.strategy} "
                    f"(chunk size {self.chunk_size}, overlap {self.chunk_overlap})")

    # ids ------------------------------------------------------------------------------------
    def reset_chunk_ids(self):
        self._global_chunk_id = 0

    def _get_next_chunk_id(self) -> str:
        cid = f"chunk_{self._global_chunk_id}"
        self._global_chunk_id += 1
        return cid

    def _create_chunk(self, text: str, start_char: int, page_num: Optional[int]) -> Chunk:
        return Chunk(text=text, chunk_id=self._get_next_chunk_id(), start_char=start_char,
                     end_char=start_char + len(text), page_number=page_num, tokens=len(text.split()))

    def _get_overlap(self, text: str) -> str:
        words = text.split()
        return text if len(words) <= self.chunk_overlap else ' '.join(words[-self.chunk_overlap:])

    # dispatch ---------------------------------------------------------------------------------
    def chunk(self, text: str, page_num: Optional[int] = None) -> List[Chunk]:
        if not text or not text.strip():
            logger.warning("Empty text provided to chunker")
            return []
        try:
            impl = {"semantic": self._semantic_chunking, "sentence": self._sentence_chunking,
                    "fixed": self._fixed_size_chunking}[self.strategy]
        except KeyError:
            raise ValueError(f"Unknown chunking strategy: {self.strategy}")
        return impl(text, page_num)

    # strategies -------------------------------------------------------------------------------
    def _semantic_chunking(self, text: str, page_num: Optional[int]) -> List[Chunk]:
        """Paragraph-packing: paragraphs (>= 20 chars) accumulate until chunk_size characters would
        be exceeded; a flushed chunk donates its last chunk_overlap words to the next one."""
        out: List[Chunk] = []
        buf, start = "", 0
        for para in (p.strip() for p in re.split(r'\n\n+', text)):
            if len(para) < 20:
                continue
            if len(buf) + len(para) > self.chunk_size:
                if len(buf) >= self.min_chunk_size:
                    out.append(self._create_chunk(buf.strip(), start, page_num))
                    tail = self._get_overlap(buf)
                    start += len(buf) - len(tail)
                    buf = tail + " "
                else:
                    start += len(buf)
                    buf = ""
            buf += para + "\n\n"
        if len(buf.strip()) >= self.min_chunk_size:
            out.append(self._create_chunk(buf.strip(), start, page_num))
        return out

    def _split_sentences(self, text: str) -> List[str]:
        if self.sent_tokenizer is not None:
            try:
                return self.sent_tokenizer.tokenize(text)
            except Exception as e:
                logger.warning(f"Sentence tokenization failed: {e}. Falling back to simple split.")
        return [s.strip() + '.' for s in re.split(r'[.!?]+', text) if s.strip()]

    def _sentence_chunking(self, text: str, page_num: Optional[int]) -> List[Chunk]:
        out: List[Chunk] = []
        buf, start = "", 0
        for sent in self._split_sentences(text):
            if len(buf) + len(sent) > self.chunk_size and buf.strip():
                out.append(self._create_chunk(buf.strip(), start, page_num))
                start += len(buf)
                buf = ""
            buf += sent + " "
        if buf.strip():
            out.append(self._create_chunk(buf.strip(), start, page_num))
        return out

    def _fixed_size_chunking(self, text: str, page_num: Optional[int]) -> List[Chunk]:
        words = text.split()
        step = max(1, self.chunk_size - self.chunk_overlap)
        out: List[Chunk] = []
        for n, i in enumerate(range(0, len(words), step)):
            out.append(self._create_chunk(' '.join(words[i:i + self.chunk_size]),
                                          n * (self.chunk_size - self.chunk_overlap), page_num))
        return out
