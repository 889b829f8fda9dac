"""Document loading + cleaning (host side).  Same public surface and -- pinned by tests/golden/clean_text.json,
generated from the reference class itself -- the same outputs as /root/reference/rag/document_processing.py
(``DocumentProcessor.process_file/process_pdf/process_text/process_string/extract_sections``).  Out of the
accelerated path.  PDF extraction needs PyPDF2, which is optional: without it ``.txt`` /
``.md`` input still works."""
from __future__ import annotations

import logging
import re
from pathlib import Path
from typing import Dict, List, Tuple

logger = logging.getLogger(__name__)

try:
    import PyPDF2
except ImportError:          # optional, exactly as in the reference (:8-11)
    PyPDF2 = None

# What the reference's quote/OCR "normalisation" lines (:157-164) actually do, as observed by running them:
#   * 'ï¬' (U+00EF U+00AC, the mojibake of the fi ligature's first two UTF-8 bytes) -> 'fi'; the following
#     'ï¬‚' -> 'fl' rule can never fire because its prefix has just been rewritten;
#   * the straight-quote replaces are identities, and the last line's adjacent quote characters parse as ONE
#     triple-quoted literal, so it replaces the 16-character text  , "'").replace(  by an apostrophe.
# Real ligature characters (U+FB01/U+FB02) and curly quotes pass through unchanged.
_OCR_FI = 'ï¬'
_QUOTE_ARTEFACT = ', "\'").replace('


class DocumentProcessor:
    """Process documents and extract clean text (PDF, TXT, Markdown)."""

    def __init__(self, config: dict):
        self.remove_headers = config.get('remove_headers', True)
        self.remove_citations = config.get('remove_citations', True)
        self.extract_sections_flag = config.get('extract_sections', False)
        # the reference stores the flag under the method's own name (:31), which shadows `extract_sections` on
        # instances (calling it there raises TypeError); kept so attribute reads agree -- the section splitter
        # itself stays reachable as DocumentProcessor.extract_sections(processor, text)
        self.extract_sections = self.extract_sections_flag
        if PyPDF2 is None:
            logger.warning("PyPDF2 not installed. PDF processing will not work.")

    def process_file(self, filepath: str) -> List[Tuple[str, int]]:
        """(text, page_number) tuples of a file; FileNotFoundError / ValueError as the reference (:47-57)."""
        path = Path(filepath)
        if not path.exists():
            raise FileNotFoundError(f"File not found: {filepath}")
        suffix = path.suffix.lower()
        if suffix == '.pdf':
            return self.process_pdf(filepath)
        if suffix in ('.txt', '.md', '.markdown'):
            return self.process_text(filepath)
        raise ValueError(f"Unsupported file type: {suffix}")

    def process_pdf(self, filepath: str) -> List[Tuple[str, int]]:
        if PyPDF2 is None:
            raise ImportError("PyPDF2 is required for PDF processing. Install with: pip install PyPDF2")
        pages = []
        try:
            with open(filepath, 'rb') as fh:
                for number, page in enumerate(PyPDF2.PdfReader(fh).pages, start=1):
                    raw = page.extract_text()
                    if raw.strip():
                        cleaned = self._clean_text(raw)
                        if cleaned:
                            pages.append((cleaned, number))
            logger.info(f"Extracted {len(pages)} pages from PDF")
        except Exception as e:
            logger.error(f"Error processing PDF: {e}")
            raise
        return pages

    def process_text(self, filepath: str) -> List[Tuple[str, int]]:
        """A text file is ONE page (reference :103-117)."""
        try:
            with open(filepath, 'r', encoding='utf-8') as fh:
                cleaned = self._clean_text(fh.read())
            if not cleaned:
                logger.warning(f"No text extracted from {filepath}")
                return []
            return [(cleaned, 1)]
        except Exception as e:
            logger.error(f"Error processing text file: {e}")
            raise

    def process_string(self, text: str) -> str:
        return self._clean_text(text)

    def _clean_text(self, text: str) -> str:
        """Whitespace collapse FIRST (so no newline survives and the line-anchored header rules only ever see one
        line -- SURVEY N3), then page-number, citation and URL removal, the OCR / quote rules above, strip."""
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
        text = text.replace(_OCR_FI, 'fi')
        text = text.replace(_QUOTE_ARTEFACT, "'")
        return text.strip()

    def extract_sections(self, text: str) -> Dict[str, str]:
        """Section name -> text.  A non-empty stripped line is a header when it is a markdown '#'..'###' header, a
        Title-Case line (letters and spaces, optional trailing colon) or a numbered Title-Case line; text before
        the first header belongs to "Introduction"; a header with no body is dropped (reference :169-217)."""
        patterns = (r'^#{1,3}\s+(.+?)$', r'^([A-Z][A-Za-z\s]+):?\s*$', r'^\d+\.?\s+([A-Z][A-Za-z\s]+)$')
        sections: Dict[str, str] = {}
        name, body = "Introduction", []
        for line in text.split('\n'):
            line = line.strip()
            if not line:
                continue
            for pat in patterns:
                m = re.match(pat, line)
                if m:
                    if body:
                        sections[name] = '\n'.join(body)
                    name, body = m.group(1).strip(), []
                    break
            else:
                body.append(line)
        if body:
            sections[name] = '\n'.join(body)
        return sections
