"""Document loading + cleaning (host side).  Public surface of /root/reference/rag/document_processing.py
(``DocumentProcessor.process_file/process_pdf/process_text/process_string``); out of the accelerated
path.  PDF extraction needs PyPDF2, which is optional: without it ``.txt`` input still works."""
from __future__ import annotations

import logging
import re
from pathlib import Path
from typing import Dict, List, Tuple

logger = logging.getLogger(__name__)


class DocumentProcessor:
    def __init__(self, config: dict):
        self.remove_headers = config.get('remove_headers', True)
        self.remove_citations = config.get('remove_citations', True)
        self.extract_sections_flag = config.get('extract_sections', False)

    def process_file(self, filepath: str) -> List[Tuple[str, int]]:
        path = Path(filepath)
        if not path.exists():
            raise FileNotFoundError(f"File not found: {filepath}")
        suffix = path.suffix.lower()
        if suffix == '.pdf':
            return self.process_pdf(filepath)
        if suffix in ('.txt', '.md'):
            return self.process_text(filepath)
        raise ValueError(f"Unsupported file type: {suffix}")

    def process_pdf(self, filepath: str) -> List[Tuple[str, int]]:
        try:
            import PyPDF2
        except ImportError as e:
            raise ImportError("PDF input needs PyPDF2; convert the document to .txt or install it") from e
        pages = []
        with open(filepath, 'rb') as fh:
            for number, page in enumerate(PyPDF2.PdfReader(fh).pages, start=1):
                cleaned = self._clean_text(page.extract_text() or "")
                if cleaned:
                    pages.append((cleaned, number))
        return pages

    def process_text(self, filepath: str) -> List[Tuple[str, int]]:
        with open(filepath, 'r', encoding='utf-8') as fh:
            raw = fh.read()
        parts = raw.split('\f') if '\f' in raw else [raw]
        pages = [(self._clean_text(p), n) for n, p in enumerate(parts, start=1)]
        return [(t, n) for t, n in pages if t]

    def process_string(self, text: str) -> str:
        return self._clean_text(text)

    def _clean_text(self, text: str) -> str:
        """Whitespace collapse first (so no paragraph break survives -- SURVEY.md N3), then header,
        citation and URL removal, ligature / quote normalisation."""
        if not text:
            return ""
        text = re.sub(r'\s+', ' ', text)
        if self.remove_headers:
            text = re.sub(r'\n\s*\d+\s*\n', '\n', text)
            text = re.sub(r'Page \d+', '', text, flags=re.IGNORECASE)
            text = re.sub(r'^\d+$', '', text, flags=re.MULTILINE)
        if self.remove_citations:
            text = re.sub(r'\[\d+\]', '', text)
            text = re.sub(r'\([^)]*\d{4}[^)]*\)', '', text)
        text = re.sub(r'http[s]?://\S+', '', text)
        for bad, good in (('ï¬\x81', 'fi'), ('ï¬‚', 'fl'), ('ﬁ', 'fi'), ('ﬂ', 'fl'),
                          ('“', '"'), ('”', '"'), ('‘', "'"), ('’', "'")):
            text = text.replace(bad, good)
        return text.strip()

    def extract_sections(self, text: str) -> Dict[str, str]:
        sections: Dict[str, str] = {}
        current, lines = "preamble", []
        for line in text.split('\n'):
            if re.match(r'^\s*(\d+(\.\d+)*\.?\s+)?[A-Z][A-Za-z ]{2,60}$', line.strip()) and len(line.split()) <= 8:
                if lines:
                    sections[current] = '\n'.join(lines).strip()
                current, lines = line.strip(), []
            else:
                lines.append(line)
        if lines:
            sections[current] = '\n'.join(lines).strip()
        return sections
