"""End-to-end pipeline orchestration -- the plugin surface callers keep using unchanged.

Public names, arguments and return values follow /root/reference/rag/pipeline.py (``RAGPipeline``
:18-341): ``setup``, ``index_documents``, ``retrieve``, ``validate_retrieval``, ``generate_answer``,
``query``, ``evaluate``, ``get_stats``.  The reference's ``main.py`` (:84-109) and
``evaluation/retrieval/benchmark.py`` (:218-283, :415-519, :858-910) drive exactly these.
The embed -> index -> retrieve components underneath are the MI355X ones of this package.
"""
from __future__ import annotations

import logging
import time
from pathlib import Path
from typing import Dict, List, Optional, Union

from rag.document_processing import DocumentProcessor
from rag.chunking import TextChunker, Chunk
from rag.embedding import EmbeddingModel
from rag.indexing import VectorStore
from rag.retrieval import ContextRetriever
from rag.generation import RAGGenerator

logger = logging.getLogger(__name__)


class RAGPipeline:
    """Wires document processing, chunking, embedding, the vector store, retrieval and generation."""

    def __init__(self, config: dict):
        self.config = config
        self.doc_processor = None
        self.chunker = None
        self.embedding_model = None
        self.vector_store = None
        self.retriever = None
        self.generator = None
        logger.info("RAG Pipeline initialized")

    def setup(self, model_interface):
        """Build every component from its config section (missing sections -> defaults)."""
        section = self.config.get
        self.doc_processor = DocumentProcessor(section('document_processing', {}))
        self.chunker = TextChunker(section('chunking', {}))
        self.embedding_model = EmbeddingModel(section('embedding', {}))
        self.vector_store = VectorStore(section('vector_store', {}))
        self.retriever = ContextRetriever(vector_store=self.vector_store, embedding_model=self.embedding_model,
                                          config=section('retrieval', {}))
        self.generator = RAGGenerator(model_interface=model_interface, config=section('generation', {}))
        logger.info("Pipeline setup complete!")

    # ---- indexing --------------------------------------------------------------------------------
    def _chunks_from(self, documents) -> List[Chunk]:
        chunks: List[Chunk] = []
        if isinstance(documents, str) and Path(documents).exists():
            pages = self.doc_processor.process_file(documents)
            logger.info(f"Extracted {len(pages)} pages from {documents}")
            for text, page_number in pages:
                chunks.extend(self.chunker.chunk(text, page_num=page_number))
        elif isinstance(documents, list):
            for number, doc in enumerate(documents, start=1):
                chunks.extend(self.chunker.chunk(self.doc_processor.process_string(doc), page_num=number))
        else:
            raise ValueError("documents must be a file path or list of strings")
        return chunks

    def index_documents(self, documents: Union[str, List[str]], show_progress: bool = True) -> float:
        """Chunk, embed (GPU encoder) and append to the slab; returns the wall time in seconds."""
        started = time.time()
        chunks = self._chunks_from(documents)
        logger.info(f"Created {len(chunks)} chunks")
        if hasattr(self.embedding_model, "embed_chunks_device"):
            # encoder output stays in HBM: no D2H -> numpy -> H2D round trip between the encoder and the slab append
            embeddings = self.embedding_model.embed_chunks_device(chunks, show_progress=show_progress)
        else:
            embeddings = self.embedding_model.embed_chunks(chunks, show_progress=show_progress)
        self.vector_store.create_index(chunks, embeddings)
        elapsed = time.time() - started
        logger.info(f"Indexing complete in {elapsed:.2f}s")
        return elapsed

    # ---- retrieval -------------------------------------------------------------------------------
    def retrieve(self, query: str, top_k: Optional[int] = None) -> List[Dict]:
        return self.retriever.retrieve(query, top_k=top_k)

    def retrieve_batch(self, queries: List[str], top_k: Optional[int] = None) -> List[List[Dict]]:
        """Additive: many queries through one encoder pass + one scan launch."""
        return self.retriever.retrieve_batch(queries, top_k=top_k)

    def validate_retrieval(self, query: str, expected_terms: List[str]) -> Dict:
        chunks = self.retrieve(query, top_k=5)
        found = [term for term in expected_terms
                 if any(term.lower() in chunk['text'].lower() for chunk in chunks)]
        return {'query': query, 'expected': expected_terms, 'found': found,
                'recall': len(found) / len(expected_terms), 'chunks': chunks}

    # ---- generation ------------------------------------------------------------------------------
    def generate_answer(self, query: str, contexts: Optional[List[Dict]] = None,
                        retrieve_if_none: bool = True) -> str:
        if contexts is None and retrieve_if_none:
            contexts = self.retriever.retrieve(query)
        context_str = '\n\n'.join(c.get('text', c.get('content', '')) for c in contexts) if contexts else ""
        if context_str:
            return self.generator.generate(query, context_str)
        return self.generator.generate_without_context(query)

    def query(self, question: str, return_context: bool = False, return_chunks: bool = False):
        retrieved = self.retriever.retrieve(question)
        context = self.retriever.get_context_string(question)
        answer = self.generator.generate(question, context)
        if not (return_context or return_chunks):
            return answer
        result = {'answer': answer}
        if return_context:
            result['context'] = context
        if return_chunks:
            result['chunks'] = retrieved
        return result

    def evaluate(self, test_questions: List[Dict[str, str]], compare_no_rag: bool = True,
                 show_progress: bool = True) -> Dict:
        questions = [qa['question'] for qa in test_questions]
        references = [qa['answer'] for qa in test_questions]
        logger.info(f"Evaluating on {len(questions)} questions")
        retrieved = [self.retriever.retrieve(q) for q in questions]
        contexts = ['\n\n'.join(chunk['text'] for chunk in chunks) for chunks in retrieved]
        predictions = self.generator.generate_batch(queries=questions, contexts=contexts,
                                                    show_progress=show_progress)
        no_rag = None
        if compare_no_rag:
            no_rag = self.generator.generate_batch_without_context(queries=questions,
                                                                   show_progress=show_progress)
        return {'questions': questions, 'references': references, 'predictions': predictions,
                'contexts': contexts, 'retrieved_chunks': retrieved, 'predictions_no_rag': no_rag}

    # ---- stats -----------------------------------------------------------------------------------
    def get_stats(self) -> Dict:
        stats = {
            'vector_store': self.vector_store.get_stats() if self.vector_store else {},
            'embedding_dim': self.embedding_model.get_dimension() if self.embedding_model else None,
            'config': self.config,
        }
        if self.embedding_model:
            em = self.embedding_model
            stats['embedding'] = {'model_name': em.model_name, 'dimension': em.get_dimension(),
                                  'device': em.device, 'batch_size': em.batch_size, 'normalize': em.normalize}
        if self.retriever:
            r = self.retriever
            stats['retrieval'] = {'top_k': r.top_k, 'similarity_threshold': r.similarity_threshold,
                                  'rerank': r.rerank, 'diversity_penalty': r.diversity_penalty,
                                  'distance_metric': r.distance_metric}
        return stats
