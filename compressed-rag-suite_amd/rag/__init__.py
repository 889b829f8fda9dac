"""RAG pipeline module -- MI355X-native embed / index / retrieve behind the reference's names
(/root/reference/rag/__init__.py:3-20).  Put ``compressed-rag-suite_amd/`` on ``sys.path`` and
``from rag import RAGPipeline`` keeps working for main.py and the evaluation harness."""

from rag.pipeline import RAGPipeline
from rag.document_processing import DocumentProcessor
from rag.chunking import TextChunker, Chunk
from rag.embedding import EmbeddingModel
from rag.indexing import VectorStore
from rag.retrieval import ContextRetriever
from rag.generation import RAGGenerator

__all__ = [
    'RAGPipeline',
    'DocumentProcessor',
    'TextChunker',
    'Chunk',
    'EmbeddingModel',
    'VectorStore',
    'ContextRetriever',
    'RAGGenerator',
]
