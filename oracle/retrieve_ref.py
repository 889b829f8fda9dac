"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's retrieval
post-processing, written as plain functions over plain data.

Follows /root/reference/rag/retrieval.py:
  distance_to_similarity  <- ContextRetriever._distance_to_similarity   :55-91
  rerank                  <- ContextRetriever._rerank                    :190-217
  apply_diversity         <- ContextRetriever._apply_diversity           :219-277
  retrieve                <- ContextRetriever.retrieve                   :93-164
and the result-dict contract of VectorStore.search (/root/reference/rag/indexing.py:125-180).

Pinned: tests/golden/retrieve_cases.json and distance_table.json hold outputs of the reference's
own ContextRetriever (loaded standalone in the build container by oracle/make_golden.py, with a
duck-typed fake store / embedder); tests/test_oracle_golden.py replays them through this file.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


def distance_to_similarity(distance: float, metric: str = "cosine") -> float:
    """retrieval.py:70-91.  NB the cosine branch is 1 - d^2/2 applied to ChromaDB's d = 1 - cos;
    that is what the reference reports as `score`, so it is reproduced as is."""
    if metric == "cosine":
        d = min(2.0, max(0.0, distance))            # :75
        return min(1.0, max(0.0, 1.0 - (d * d / 2.0)))  # :76-77
    if metric == "l2":
        return 1.0 / (1.0 + distance)               # :82
    if metric == "ip":
        return min(1.0, max(0.0, (distance + 2.0) / 2.0))  # :87
    return max(0.0, 1.0 - (distance / 2.0))         # :91


def rerank(query: str, chunks: List[Dict], top_k: int) -> List[Dict]:
    """retrieval.py:201-217: 0.7 * score + 0.3 * |q_tokens ∩ chunk_tokens| / max(|q_tokens|, 1),
    stable sort descending, cut to top_k.  Mutates the dicts (adds 'rerank_score') like the reference."""
    q_tokens = set(query.lower().split())
    denom = max(len(q_tokens), 1)
    for ch in chunks:
        overlap = len(q_tokens & set(ch["text"].lower().split()))
        ch["rerank_score"] = ch["score"] * 0.7 + (overlap / denom) * 0.3
    # list.sort(reverse=True) is stable and keeps equal keys in original order
    ordered = sorted(chunks, key=lambda c: c.get("rerank_score", c["score"]), reverse=True)
    return ordered[:top_k]


def apply_diversity(chunks: List[Dict], diversity_penalty: float,
                    embed: Callable[[List[str]], np.ndarray]) -> List[Dict]:
    """retrieval.py:230-277: greedy MMR re-ordering.  lambda = 1 - penalty; the first chunk is
    always kept first; similarity to the selected set is floored at 0.0 (:255) and the first
    maximum wins ties (strict '>' at :266).  Never drops a chunk."""
    if len(chunks) <= 1:
        return chunks
    lam = 1.0 - diversity_penalty
    emb = embed([c["text"] for c in chunks])       # :238-239 -- a second encoder pass
    picked = [0]
    rest = list(range(1, len(chunks)))
    while len(picked) < len(chunks) and rest:
        best, best_val = None, -float("inf")
        for idx in rest:
            max_sim = 0.0
            for p in picked:
                sim = np.dot(emb[idx], emb[p]) / (np.linalg.norm(emb[idx]) * np.linalg.norm(emb[p]))
                max_sim = max(max_sim, sim)
            val = lam * chunks[idx]["score"] - (1 - lam) * max_sim
            if val > best_val:
                best_val, best = val, idx
        if best is None:
            break
        picked.append(best)
        rest.remove(best)
    return [chunks[i] for i in picked]


def retrieve(query: str, *, search: Callable[..., Dict], embed: Callable[[List[str]], np.ndarray],
             top_k: int = 3, similarity_threshold: float = 0.0, do_rerank: bool = False,
             diversity_penalty: float = 0.0, metric: str = "cosine",
             k_override: Optional[int] = None, filters: Optional[dict] = None) -> List[Dict]:
    """retrieval.py:110-160.  `search(query_embedding=..., top_k=..., where=...)` must return the
    ChromaDB-shaped dict of list-of-lists; `embed(str | list[str])` returns fp32 [n, d]."""
    k = k_override or top_k                                            # :110
    q_emb = embed(query)                                               # :114
    res = search(query_embedding=q_emb, top_k=k * 2 if do_rerank else k, where=filters)  # :117-121
    if not res["ids"][0]:                                              # :124-126
        return []
    out = []
    for i in range(len(res["ids"][0])):                                # :130-144
        dist = res["distances"][0][i]
        item = {
            "text": res["documents"][0][i],
            "score": distance_to_similarity(dist, metric),
            "distance": dist,
            "metadata": res["metadatas"][0][i] if res["metadatas"] else {},
            "chunk_id": res["ids"][0][i],
        }
        if item["score"] >= similarity_threshold:
            out.append(item)
    if not out:                                                        # :146-148
        return []
    if do_rerank and len(out) > k:                                     # :151-154
        out = rerank(query, out, k)
    else:
        out = out[:k]
    if diversity_penalty > 0 and len(out) > 1:                         # :157-158
        out = apply_diversity(out, diversity_penalty, embed)
    return out


# ------------------------------------------------------------------------------------------------
# VectorStore contract (rag/indexing.py:57-211) over a numpy matrix: the CPU statement of what
# create_index / search / get_stats must return, used to check the HIP-backed store's host logic.
def _where_ok(meta: dict, where: dict) -> bool:
    """One row against a ChromaDB `where` document, evaluated the slow, obvious way (chromadb==1.3.0 is not vendored; these are
    its documented operators: plain equality, $eq $ne $in $nin $gt $gte $lt $lte, $and / $or)."""
    for key, want in where.items():
        if key == "$and":
            ok = all(_where_ok(meta, w) for w in want)
        elif key == "$or":
            ok = any(_where_ok(meta, w) for w in want)
        else:
            have = meta.get(key)
            op, val = ("$eq", want)
            if isinstance(want, dict):
                (op, val), = want.items()
            num = lambda x: isinstance(x, (int, float)) and not isinstance(x, bool)           # noqa: E731
            if op == "$eq":
                ok = have == val
            elif op == "$ne":
                ok = have != val
            elif op == "$in":
                ok = isinstance(val, (list, tuple)) and have in val
            elif op == "$nin":
                ok = isinstance(val, (list, tuple)) and have not in val
            elif op in ("$gt", "$gte", "$lt", "$lte"):
                ok = num(have) and num(val) and {"$gt": have > val, "$gte": have >= val, "$lt": have < val, "$lte": have <= val}[op] if num(have) and num(val) else False
            else:
                ok = False
        if not ok:
            return False
    return True


def _doc_ok(text: str, cond: dict) -> bool:
    for op, val in cond.items():
        if op == "$contains":
            ok = val in text
        elif op == "$not_contains":
            ok = val not in text
        elif op == "$and":
            ok = all(_doc_ok(text, c) for c in val)
        elif op == "$or":
            ok = any(_doc_ok(text, c) for c in val)
        else:
            ok = True
        if not ok:
            return False
    return True


class StoreRef:
    def __init__(self):
        self.ids: List[str] = []
        self.docs: List[str] = []
        self.metas: List[dict] = []
        self.vecs = None  # fp32 [n, d], unit rows

    @staticmethod
    def chunk_metadata(chunk, fields: Optional[Sequence[str]] = None) -> dict:
        fields = ["page_number", "section", "tokens"] if fields is None else fields  # :95-96
        meta = {}
        for f in fields:                                                           # :99-108
            v = getattr(chunk, f, None)
            if v is not None:
                meta[f] = v if isinstance(v, (str, int, float)) else str(v)
        return meta

    def create_index(self, chunks, embeddings, metadata_fields=None):
        if len(chunks) == 0:                                                       # :71-73
            return
        if len(chunks) != len(embeddings):                                         # :75-76
            raise ValueError(f"Chunk count ({len(chunks)}) doesn't match embedding count ({len(embeddings)})")
        e = np.asarray(embeddings, dtype=np.float32)
        e = e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12)
        self.vecs = e if self.vecs is None else np.concatenate([self.vecs, e])
        self.ids += [c.chunk_id for c in chunks]                                   # :91
        self.docs += [c.text for c in chunks]                                      # :92
        self.metas += [self.chunk_metadata(c, metadata_fields) for c in chunks]

    def count(self) -> int:
        return len(self.ids)

    def search(self, query_embedding, top_k=5, where=None, where_document=None) -> Dict:
        if self.vecs is None:                                                      # :144-145
            raise ValueError("No collection available. Create index first.")
        if self.count() == 0:                                                      # :147-149
            return {"ids": [[]], "documents": [[]], "metadatas": [[]], "distances": [[]]}
        top_k = min(top_k, self.count())                                           # :152-153
        q = np.asarray(query_embedding, dtype=np.float32).reshape(-1)              # :156-168
        q = q / max(float(np.linalg.norm(q)), 1e-12)
        cos = self.vecs.astype(np.float64) @ q.astype(np.float64)
        mask = np.ones(self.count(), dtype=bool)
        if where:                                            # chromadb metadata filters the reference forwards (:129-130,174)
            mask &= np.array([_where_ok(m, where) for m in self.metas])
        if where_document:                                   # ... and its document filters
            mask &= np.array([_doc_ok(t, where_document) for t in self.docs])
        idx = np.nonzero(mask)[0]
        order = idx[np.lexsort((idx, -cos[idx]))][:top_k]
        return {
            "ids": [[self.ids[i] for i in order]],
            "documents": [[self.docs[i] for i in order]],
            "metadatas": [[self.metas[i] for i in order]],
            "distances": [[float(1.0 - cos[i]) for i in order]],   # cosine space: 1 - cos, ascending
        }

    def get_stats(self) -> Dict:                                                   # :198-211
        if self.vecs is None:
            return {"status": "empty", "count": 0}
        return {"name": "rag_documents", "count": self.count(), "metadata": {"hnsw:space": "cosine"}}


def isclose(a: float, b: float, tol: float = 1e-12) -> bool:
    return math.isclose(a, b, rel_tol=tol, abs_tol=tol)
