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
        self.batch_queries = int(config.get('batch_queries', 64))    # additive: queries per launch of retrieve_batch's engine
        # additive: where the MMR step takes the chunks' vectors from.  'reembed' = encode the chunk texts again, as the
        # reference does (rag/retrieval.py:238-239); 'index' = the fp32 rows the store kept when those very texts were
        # indexed (the same encoder's output for the same text: equal up to the rounding of a different batch composition;
        # no tokenisation, no encoder pass); 'auto' = retrieve() re-embeds (reference parity), retrieve_batch() reads the index
        self.mmr_vectors = config.get('mmr_vectors', 'auto')
        if self.mmr_vectors not in ('auto', 'reembed', 'index'):
            raise ValueError(f"mmr_vectors must be 'auto', 'reembed' or 'index', got {self.mmr_vectors!r}")
        self._engine, self._engine_key = None, None
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
            if self.mmr_vectors == 'index' and self.diversity_penalty > 0:
                return self.retrieve_batch([query], top_k=top_k)[0] if filters is None else self._post_process(query, chunks, k)
            return self._post_process(query, chunks, k)
        except Exception as e:
            logger.error(f"Retrieval failed: {e}")
            raise

    # ---- batched retrieval (additive; the reference has no batched entry point) --------------------------------------
    def _engine_for(self, fetch: int, seq: int, n_batches: int = 0):
        """The throughput engine (rag/_engine.py: role lanes, hipGraph replay, several batches in flight) for this store /
        encoder pair, or None when the store's layout needs the general path (several shards, filters, top_k > 64)."""
        store, model = self.vector_store, self.embedding_model
        view = store.engine_view() if hasattr(store, "engine_view") else None
        enc = getattr(model, "model", None)
        if view is None or enc is None or fetch > 64 or not getattr(model, "normalize", True):
            return None
        # one encoder forward serves a group of batches (rag/_engine.py): never more of them than a call of this size brings
        cap = 1 << (max(1, n_batches).bit_length() - 1)
        key = (fetch, seq, view.n, int(view.slab.data_ptr()), self.batch_queries, min(cap, 16))
        if self._engine_key != key:
            from rag._engine import RetrievalEngine
            self._engine = RetrievalEngine(enc, view, self.batch_queries, seq, fetch, k_scan=store.refine_overfetch,
                                           refine=view.shadow is not None, exact=store.refine_exact, exact_cap=store.exact_cap,
                                           group_cap=cap)
            self._engine_key = key
        return self._engine

    def _search_many(self, queries: List[str], fetch: int):
        """Generator over PIECES of the query list, in order: each piece a list of per-query (scores fp32 [<= fetch], sidecar rows
        int64 [<= fetch]) numpy arrays, best first (or, once, a duck-typed store's search_batch dict).  With the throughput
        engine a piece is one device batch, yielded as soon as it is through -- the caller builds that batch's dicts while the
        device works on the following ones."""
        import numpy as _np
        model, store = self.embedding_model, self.vector_store
        qb = self.batch_queries
        eng = None
        if len(queries) >= qb and hasattr(store, 'engine_view') and store.collection is not None and store.collection.count() > 0:
            token_ids = model.tokenize(list(queries))
            longest = max(len(t) for t in token_ids)
            seq = next((st for st in (16, 32, 64) if longest <= st and st <= model.shape.max_seq), None)
            eng = self._engine_for(min(fetch, store.collection.count()), seq, -(-len(queries) // qb)) if seq else None
        if eng is None:                       # general path: one encoder pass + one search launch for the whole list
            emb = model.embed_device(list(queries)) if hasattr(model, "embed_device") else model.embed(list(queries))
            if not hasattr(store, "search_rows"):          # any store with the additive search_batch (duck-typed)
                yield store.search_batch(emb, top_k=fetch)
                return
            scores, rows = store.search_rows(emb, fetch)
            yield [(scores[i], rows[i]) for i in range(len(queries))]
            return
        pad = getattr(model.tokenizer, "pad_id", 0)

        def batches():
            for lo in range(0, len(token_ids), qb):
                ids = _np.full((min(qb, len(token_ids) - lo), eng.seq), pad, dtype=_np.int32)
                lens = _np.empty(ids.shape[0], dtype=_np.int32)
                for r, t in enumerate(token_ids[lo:lo + qb]):
                    ids[r, :len(t)] = t
                    lens[r] = len(t)
                yield ids, lens

        tally = {"queries": len(queries), "certified": 0, "escalated": 0, "unproven": 0}
        for s, r, st in eng.search_token_batches(batches()):
            tally["certified"] += int((st == 0).sum())
            tally["escalated"] += int((st == 1).sum()) if eng.exact else 0
            tally["unproven"] += int((st == 2).sum()) + (0 if eng.exact else int((st == 1).sum()))
            yield [(s[i], r[i]) for i in range(s.shape[0])]
        if eng.refine:
            store.last_exactness = tally

    def retrieve_batch(self, queries: List[str], top_k: Optional[int] = None) -> List[List[Dict]]:
        """``[retrieve(q) for q in queries]`` for many queries at once: the encoder forwards and scans of the whole list
        run through the throughput engine (Qb = ``batch_queries`` per launch, several launches in flight), scoring and the
        threshold are vectorised (same fp64 arithmetic as ``_distance_to_similarity``), and the MMR step re-embeds the chunk
        texts of ALL queries in one encoder pass (the reference re-embeds per query, rag/retrieval.py:238-239).
        Same dicts, same order as ``retrieve`` up to the rounding of batched encoder forwards."""
        k = top_k or self.top_k
        if not queries:
            return []
        store = self.vector_store
        col = store.collection
        if col is None:
            raise ValueError("No collection available. Create index first.")
        fetch = k * 2 if self.rerank else k
        per_query: List[List[Dict]] = []
        row_of: Dict[int, int] = {}           # id(chunk dict) -> sidecar row (our store only)
        ids_l, docs_l, metas_l = getattr(col, 'ids', None), getattr(col, 'documents', None), getattr(col, 'metadatas', None)
        queries = list(queries)
        for piece in self._search_many(queries, fetch):      # one device batch at a time: its dicts are built while the next ones run
            if isinstance(piece, dict):       # a duck-typed store's search_batch dict: lists per query
                ids_l = docs_l = metas_l = None
                piece = [(np.asarray(piece['distances'][p], dtype=np.float64), (piece['ids'][p], piece['documents'][p],
                          piece['metadatas'][p] if piece.get('metadatas') else None)) for p in range(len(queries))]
            for sc, rows in piece:
                query = queries[len(per_query)]
                if ids_l is None:
                    dist, (h_ids, h_docs, h_metas) = sc, rows
                    rows = np.arange(len(h_ids))
                else:
                    valid = rows >= 0
                    rows, sc = rows[valid], sc[valid]
                    dist = (np.float32(1.0) - sc.astype(np.float32)).astype(np.float64)  # the store's distances, as search() returns them
                    h_ids, h_docs, h_metas = ids_l, docs_l, metas_l
                if rows.size == 0:
                    logger.warning("No results found for query")
                    per_query.append([])
                    continue
                if self.distance_metric == 'cosine':                                          # _distance_to_similarity, vectorised (same fp64 ops)
                    d = np.minimum(np.maximum(dist, 0.0), 2.0)
                    score = np.minimum(np.maximum(1.0 - (d * d / 2.0), 0.0), 1.0)
                else:
                    score = np.array([self._distance_to_similarity(float(x)) for x in dist])
                keep = score >= self.similarity_threshold
                chunks = [{'text': h_docs[r], 'score': float(s_), 'distance': float(d_), 'metadata': h_metas[r] if h_metas else {},
                           'chunk_id': h_ids[r]}
                          for r, s_, d_, ok in zip(rows.tolist(), score.tolist(), dist.tolist(), keep.tolist()) if ok]
                if ids_l is not None:
                    for c_, r_ in zip(chunks, [r for r, ok in zip(rows.tolist(), keep.tolist()) if ok]):
                        row_of[id(c_)] = r_
                if not chunks:
                    logger.warning(f"No chunks passed similarity threshold of {self.similarity_threshold}")
                    per_query.append([])
                    continue
                if self.rerank and len(chunks) > k:
                    chunks = self._rerank(query, chunks, k)
                else:
                    chunks = chunks[:k]
                per_query.append(chunks)
        from_index = self.mmr_vectors in ('auto', 'index') and ids_l is not None and hasattr(store, 'rows_f32')
        if self.diversity_penalty > 0 and from_index:
            # the chunks' vectors straight from the index (one gather for the whole batch), then the reference's greedy MMR
            need = sorted({row_of[id(c)] for chunks in per_query if len(chunks) > 1 for c in chunks})
            vecs = store.rows_f32(need) if need else None
            if vecs is None:
                from_index = False
            else:
                at = {r: p for p, r in enumerate(need)}
                per_query = [self._apply_diversity(chunks, vectors=vecs[[at[row_of[id(c)]] for c in chunks]])
                             if len(chunks) > 1 else chunks for chunks in per_query]
        if self.diversity_penalty > 0 and not from_index:
            # ONE encoder pass for the chunk texts of every query (deduplicated), then the reference's greedy MMR per query
            texts, index = [], {}
            for chunks in per_query:
                if len(chunks) > 1:
                    for c in chunks:
                        if c['text'] not in index:
                            index[c['text']] = len(texts)
                            texts.append(c['text'])
            if texts:
                vectors = self.embedding_model.embed(texts)
                per_query = [self._apply_diversity(chunks, vectors=vectors[[index[c['text']] for c in chunks]])
                             if len(chunks) > 1 else chunks for chunks in per_query]
        return per_query

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

    def _apply_diversity(self, chunks: List[Dict], vectors=None) -> List[Dict]:
        """Greedy maximal-marginal-relevance re-ordering over re-embedded chunk texts:
        value = lambda * score - (1 - lambda) * max(0, max cos to the already selected).
        `vectors` (retrieve_batch): the chunks' embeddings, already at hand for the whole batch.

        Same values as the reference's triple loop (rag/retrieval.py:246-275), computed once each: the reference
        re-evaluates cos(candidate, chosen) for every chosen chunk in every round (~k^3/3 dot products and twice as many
        norms); a candidate's running maximum only ever changes by the newest chosen chunk, and max() of floats does not
        depend on the order it is taken in, so one cos per (candidate, newly chosen) pair -- k^2/2 -- gives bit-identical
        `closest`, `value` and order (pinned by tests/golden/retrieve_cases.json)."""
        if len(chunks) <= 1:
            return chunks
        lam = 1.0 - self.diversity_penalty
        if vectors is None:
            vectors = self.embedding_model.embed([c['text'] for c in chunks])
        norms = [np.linalg.norm(v) for v in vectors]
        order = [0]
        pending = list(range(1, len(chunks)))
        closest = {cand: 0.0 for cand in pending}
        newest = 0
        while pending and len(order) < len(chunks):
            winner, winner_value = None, -float('inf')
            for cand in pending:
                cos = np.dot(vectors[cand], vectors[newest]) / (norms[cand] * norms[newest])
                closest[cand] = max(closest[cand], cos)
                value = lam * chunks[cand]['score'] - (1 - lam) * closest[cand]
                if value > winner_value:
                    winner, winner_value = cand, value
            if winner is None:
                break
            order.append(winner)
            pending.remove(winner)
            newest = winner
        return [chunks[i] for i in order]
