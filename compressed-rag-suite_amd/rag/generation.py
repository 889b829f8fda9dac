"""Answer generation glue (LLM decode is a different hot path; SURVEY.md section 2 row 7: out of scope).
Keeps the names ``RAGPipeline`` and the evaluation harness call on /root/reference/rag/generation.py:
``generate``, ``generate_batch``, ``generate_without_context``, ``generate_batch_without_context``."""
from __future__ import annotations

import logging
from typing import List

logger = logging.getLogger(__name__)


class RAGGenerator:
    def __init__(self, model_interface, config: dict):
        self.model = model_interface
        self.max_new_tokens = config.get('max_new_tokens', 128)
        self.temperature = config.get('temperature', 0.3)
        self.top_p = config.get('top_p', 0.9)
        self.do_sample = config.get('do_sample', True)
        self.repetition_penalty = config.get('repetition_penalty', 1.15)
        self.use_chat_template = config.get('use_chat_template', True)
        self.max_context_chars = config.get('max_context_chars', 2000)

    def _truncate_context(self, context: str, max_chars: int = 2000) -> str:
        if len(context) <= max_chars:
            return context
        cut = context[:max_chars]
        stop = cut.rfind('. ')
        return cut[:stop + 1] if stop > max_chars // 2 else cut

    def _prompt(self, query: str, context: str) -> str:
        if context:
            return ("Answer the question using only the context below. Be concise.\n\n"
                    f"Context:\n{context}\n\nQuestion: {query}\nAnswer:")
        return f"Answer the question concisely.\n\nQuestion: {query}\nAnswer:"

    def _call(self, prompt: str) -> str:
        if self.model is None:
            raise RuntimeError("RAGGenerator has no model_interface; pass one to RAGPipeline.setup()")
        out = self.model.generate(prompt, max_new_tokens=self.max_new_tokens, temperature=self.temperature,
                                  top_p=self.top_p, do_sample=self.do_sample,
                                  repetition_penalty=self.repetition_penalty)
        return self._clean_answer(out)

    def _clean_answer(self, answer: str, max_sentences: int = 4) -> str:
        answer = (answer or "").strip()
        for marker in ("Question:", "Context:", "\n\n\n"):
            pos = answer.find(marker)
            if pos > 0:
                answer = answer[:pos].strip()
        parts = [p for p in answer.replace('\n', ' ').split('. ') if p]
        if len(parts) > max_sentences:
            answer = '. '.join(parts[:max_sentences]).rstrip('.') + '.'
        return answer

    def generate(self, query: str, context: str) -> str:
        return self._call(self._prompt(query, self._truncate_context(context, self.max_context_chars)))

    def generate_without_context(self, query: str) -> str:
        return self._call(self._prompt(query, ""))

    def generate_batch(self, queries: List[str], contexts: List[str], show_progress: bool = True) -> List[str]:
        return [self.generate(q, c) for q, c in zip(queries, contexts)]

    def generate_batch_without_context(self, queries: List[str], show_progress: bool = True) -> List[str]:
        return [self.generate_without_context(q) for q in queries]
