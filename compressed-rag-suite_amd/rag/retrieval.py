"""Context retrieval: embed the query, search the slab, score, threshold, rerank, diversify.

Same class / method names, arguments, return dicts and numeric behaviour as
/root/reference/rag/retrieval.py (``ContextRetriever`` :13-277) -- pinned bit for bit by
tests/golden/retrieve_cases.json, which holds the reference class's own outputs.  The scoring,
lexical rerank and MMR steps are a few hundred scalar operations per query and stay on the host
in fp64 Python exactly like the reference; the heavy steps they call (``embed``, ``search``) run
on the GPU.

Additive: ``retrieve_batch`` embeds and searches many queries in one launch each and then applies
the identical per-query post-processing; ``reuse_index_embeddings`` is NOT offered because the
reference re-embeds chunk texts for MMR (:238-239) and near-ties could reorder otherwise.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional

import numpy as np

from rag.indexing import VectorStore
from rag.embedding import EmbeddingModel

logger = logging.getLogger(__name__)


class ContextRetriever:
    """Retrieve relevant context for queries (with re-ranking and diversity mechanisms)."""

    def __init__(self, vector_store: VectorStore, embedding_model: EmbeddingModel, config: dict):
        self.vector_store = vector_store
        self.embedding_model = embedding_model
        self.top_k = config.get('top_k', 3)
        self.similarity_threshold = config.get('similarity_threshold', 0.0)
        self.rerank = config.get('rerank', False)
        self.diversity_penalty = config.get('diversity_penalty', 0.0)
        self.distance_metric = self._get_distance_metric()
        logger.info(f"Using distance metric: {self.distance_metric}")

    def _get_distance_metric(self) -> str:
        """The store's space ('cosine' unless the collection says otherwise or does not exist yet)."""
        try:
            collection = self.vector_store.collection
            if collection:
                return collection.metadata.get('hnsw:space', 'cosine')
        except Exception:
            pass
        return 'cosine'

    # ---- scoring ---------------------------------------------------------------------------------
    def _distance_to_similarity(self, distance: float) -> float:
        """Map a store distance to a [0, 1] score.  The cosine branch is the reference's
        ``1 - d^2/2`` with d clamped to [0, 2] (its :75-77), applied to d = 1 - cos as returned by
        the store: that IS the reference's ``score``, so it is kept as is."""
        metric = self.distance_metric
        if metric == 'cosine':
            d = max(0.0, min(2.0, distance))
            return max(0.0, min(1.0, 1.0 - (d * d / 2.0)))
        if metric == 'l2':
            return 1.0 / (1.0 + distance)
        if metric == 'ip':
            return max(0.0, min(1.0, (distance + 2.0) / 2.0))
        logger.warning(f"Unknown distance metric: {metric}, using default conversion")
        return max(0.0, 1.0 - (distance / 2.0))

    def _hits_to_chunks(self, ids, documents, metadatas, distances) -> List[Dict]:
        chunks = []
        for pos, chunk_id in enumerate(ids):
            distance = distances[pos]
            item = {
                'text': documents[pos],
                'score': self._distance_to_similarity(distance),
                'distance': distance,
                'metadata': metadatas[pos] if metadatas else {},
                'chunk_id': chunk_id,
            }
            if item['score'] >= self.similarity_threshold:
                chunks.append(item)
        return chunks

    def _post_process(self, query: str, chunks: List[Dict], k: int) -> List[Dict]:
        if not chunks:
            logger.warning(f"No chunks passed similarity threshold of {self.similarity_threshold}")
            return []
        if self.rerank and len(chunks) > k:
            chunks = self._rerank(query, chunks, k)
        else:
            chunks = chunks[:k]
        if self.diversity_penalty > 0 and len(chunks) > 1:
            chunks = self._apply_diversity(chunks)
        return chunks

    # ---- public ----------------------------------------------------------------------------------
    def retrieve(self, query: str, top_k: Optional[int] = None, filters: Optional[dict] = None) -> List[Dict]:
        """List of dicts with 'text', 'score', 'distance', 'metadata', 'chunk_id' (+ 'rerank_score'
        when re-ranked), at most k of them."""
        k = top_k or self.top_k
        try:
            query_embedding = self.embedding_model.embed(query)
            results = self.vector_store.search(query_embedding=query_embedding,
                                               top_k=k * 2 if self.rerank else k, where=filters)
            if not results['ids'][0]:
                logger.warning("No results found for query")
                return []
            metas = results['metadatas'][0] if results['metadatas'] else None
            chunks = self._hits_to_chunks(results['ids'][0], results['documents'][0], metas,
                                          results['distances'][0])
            return self._post_process(query, chunks, k)
        except Exception as e:
            logger.error(f"Retrieval failed: {e}")
            raise

    def retrieve_batch(self, queries: List[str], top_k: Optional[int] = None) -> List[List[Dict]]:
        """``[retrieve(q) for q in queries]`` with one encoder pass and one scan launch for the
        whole batch (requires the store's additive ``search_batch``)."""
        k = top_k or self.top_k
        if not queries:
            return []
        embeddings = self.embedding_model.embed(list(queries))
        results = self.vector_store.search_batch(embeddings, top_k=k * 2 if self.rerank else k)
        out = []
        for pos, query in enumerate(queries):
            if not results['ids'][pos]:
                out.append([])
                continue
            chunks = self._hits_to_chunks(results['ids'][pos], results['documents'][pos],
                                          results['metadatas'][pos], results['distances'][pos])
            out.append(self._post_process(query, chunks, k))
        return out

    def get_context_string(self, query: str, top_k: Optional[int] = None, separator: str = "\n\n") -> str:
        chunks = self.retrieve(query, top_k=top_k)
        return separator.join(chunk['text'] for chunk in chunks) if chunks else ""

    # ---- rerank / diversity ----------------------------------------------------------------------
    def _rerank(self, query: str, chunks: List[Dict], top_k: int) -> List[Dict]:
        """70 % semantic score + 30 % fraction of query tokens present in the chunk."""
        wanted = set(query.lower().split())
        norm = max(len(wanted), 1)
        for chunk in chunks:
            hits = len(wanted & set(chunk['text'].lower().split()))
            chunk['rerank_score'] = chunk['score'] * 0.7 + (hits / norm) * 0.3
        chunks.sort(key=lambda c: c.get('rerank_score', c['score']), reverse=True)
        return chunks[:top_k]

    def _apply_diversity(self, chunks: List[Dict]) -> List[Dict]:
        """Greedy maximal-marginal-relevance re-ordering over re-embedded chunk texts:
        value = lambda * score - (1 - lambda) * max(0, max cos to the already selected)."""
        if len(chunks) <= 1:
            return chunks
        lam = 1.0 - self.diversity_penalty
        vectors = self.embedding_model.embed([c['text'] for c in chunks])
        order = [0]
        pending = list(range(1, len(chunks)))
        while pending and len(order) < len(chunks):
            winner, winner_value = None, -float('inf')
            for cand in pending:
                closest = 0.0
                for chosen in order:
                    cos = np.dot(vectors[cand], vectors[chosen]) / (
                        np.linalg.norm(vectors[cand]) * np.linalg.norm(vectors[chosen]))
                    closest = max(closest, cos)
                value = lam * chunks[cand]['score'] - (1 - lam) * closest
                if value > winner_value:
                    winner, winner_value = cand, value
            if winner is None:
                break
            order.append(winner)
            pending.remove(winner)
        return [chunks[i] for i in order]
