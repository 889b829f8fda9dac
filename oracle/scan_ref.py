"""ORACLE (test infrastructure, not product code) -- exact cosine top-k on the CPU.

Restates, as plain numpy, what the reference obtains from ChromaDB for a collection
created with ``metadata={"hnsw:space": "cosine"}``:

* ``collection.add(embeddings=...)``      -- /root/reference/rag/indexing.py:114-119
* ``collection.query(query_embeddings, n_results)`` -- /root/reference/rag/indexing.py:171-176
  returns, per query, the ``n_results`` nearest rows in ascending *cosine distance*
  ``1 - cos(q, c)`` (chromadb==1.3.0, /root/reference/requirements.txt:19, not vendored).

ChromaDB's HNSW search is approximate and its tie order is unspecified; this oracle is the
*exact* answer the approximate index converges to (parity unpinned at the
ChromaDB boundary -- no golden vector of the reference pins query results for synthetic
corpora).  Tie rule chosen here and in the HIP kernels alike: score descending, then id
ascending.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

NEG_INF = np.float32(-np.inf)


def l2_normalize_rows(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """Row-wise ``x / max(||x||_2, eps)`` in fp32 -- the Normalize() step applied by
    sentence-transformers when ``normalize_embeddings=True`` (/root/reference/rag/embedding.py:69)
    and, for an already-normalised input, ChromaDB's own cosine-space normalisation."""
    x = np.asarray(x, dtype=np.float32)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    return (x / np.maximum(n, eps)).astype(np.float32)


def quantize_rows_f16(x: np.ndarray) -> np.ndarray:
    """fp32 unit rows -> the fp16 slab rows the HIP store keeps (round-to-nearest-even)."""
    return np.asarray(x, dtype=np.float32).astype(np.float16)


def quantize_rows_i8(x: np.ndarray):
    """fp32 rows -> (int8 rows, fp32 per-row scale) with symmetric per-vector scaling
    ``s = max|x| / 127, q = rint(x / s)`` (SURVEY.md section 8(d) synthetic-input recipe)."""
    x = np.asarray(x, dtype=np.float32)
    amax = np.abs(x).max(axis=1)
    scale = (amax / np.float32(127.0)).astype(np.float32)
    safe = np.where(scale > 0, scale, np.float32(1.0)).astype(np.float32)
    q = np.rint(x / safe[:, None]).clip(-127, 127).astype(np.int8)
    return q, scale


def quantize_query_fx16(q16: np.ndarray):
    """What the int8 scan kernel does to each fp16 query before the integer MFMA (csrc/scan_i8.hip):
    sq = max|q| / 32512 (fp32), qi = rint(q / sq) in fp32 -> 16-bit fixed point.  Returns
    (qi int32 [nq, d], sq fp32 [nq]); the query the kernel effectively searches with is qi * sq."""
    q = np.asarray(q16, dtype=np.float16).astype(np.float32)
    amax = np.abs(q).max(axis=1)
    sq = np.where(amax > 0, amax / np.float32(32512.0), np.float32(1.0)).astype(np.float32)
    qi = np.rint(q / sq[:, None]).astype(np.int32)
    return qi, sq


def dequantized_queries(q16: np.ndarray) -> np.ndarray:
    qi, sq = quantize_query_fx16(q16)
    return qi.astype(np.float64) * sq.astype(np.float64)[:, None]


def _order(scores: np.ndarray, ids: np.ndarray) -> np.ndarray:
    # score descending, id ascending
    return np.lexsort((ids, -scores.astype(np.float64)))


def cosine_topk_ref(q: np.ndarray, slab: np.ndarray, k: int, *, scales: np.ndarray | None = None,
                    id_base: int = 0, block: int = 1 << 16, accumulate=np.float32):
    """Exact top-k inner product of ``q[nq,d]`` against ``slab[n,d]`` (rows are unit vectors,
    so the inner product is the cosine).  Inputs are taken *as stored* (fp16 / int8) and
    widened; ``scales`` (fp32 per row) multiplies int8 rows' dot products.

    Returns (scores fp32 [nq,k], ids int64 [nq,k]); slots beyond ``n`` hold (-inf, -1).
    """
    q = np.asarray(q)
    nq = q.shape[0]
    n = slab.shape[0]
    kk = min(k, n)
    best_s = np.full((nq, 0), NEG_INF, dtype=np.float32)
    best_i = np.zeros((nq, 0), dtype=np.int64)
    qa = q.astype(accumulate)
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        s = qa @ slab[lo:hi].astype(accumulate).T
        if scales is not None:
            s = s * scales[lo:hi].astype(accumulate)[None, :]
        s = s.astype(np.float32)
        ids = np.arange(lo, hi, dtype=np.int64)
        cat_s = np.concatenate([best_s, s], axis=1)
        cat_i = np.concatenate([best_i, np.broadcast_to(ids, (nq, hi - lo))], axis=1)
        if cat_s.shape[1] > 4 * max(kk, 1):
            # cheap prefilter: keep everything >= the kk-th largest value (ties kept)
            kth = np.partition(cat_s, cat_s.shape[1] - kk, axis=1)[:, cat_s.shape[1] - kk]
            new_s, new_i = [], []
            for r in range(nq):
                m = cat_s[r] >= kth[r]
                rs, ri = cat_s[r][m], cat_i[r][m]
                o = _order(rs, ri)[:kk]
                new_s.append(rs[o]); new_i.append(ri[o])
            best_s = np.stack(new_s); best_i = np.stack(new_i)
        else:
            new_s, new_i = [], []
            for r in range(nq):
                o = _order(cat_s[r], cat_i[r])[:kk]
                new_s.append(cat_s[r][o]); new_i.append(cat_i[r][o])
            best_s = np.stack(new_s); best_i = np.stack(new_i)
    out_s = np.full((nq, k), NEG_INF, dtype=np.float32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    out_s[:, :kk] = best_s
    out_i[:, :kk] = best_i + id_base
    return out_s, out_i


def full_scores_f64(q: np.ndarray, slab: np.ndarray, scales: np.ndarray | None = None) -> np.ndarray:
    """All nq x n inner products in fp64 (small cases only) -- used by the tests to accept
    rank swaps between entries whose true scores differ by less than fp32 rounding."""
    s = q.astype(np.float64) @ slab.astype(np.float64).T
    if scales is not None:
        s = s * scales.astype(np.float64)[None, :]
    return s


def merge_topk_ref(scores: np.ndarray, ids: np.ndarray, k: int):
    """Merge ``[G, nq, kin]`` partial lists (the per-shard results an all-gather delivers)
    into the global top-k per query; invalid slots have id < 0.  Same tie rule."""
    g, nq, kin = scores.shape
    out_s = np.full((nq, k), NEG_INF, dtype=np.float32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    for r in range(nq):
        s = scores[:, r, :].reshape(-1)
        i = ids[:, r, :].reshape(-1)
        m = i >= 0
        s, i = s[m], i[m]
        o = _order(s, i)[:k]
        out_s[r, :len(o)] = s[o]
        out_i[r, :len(o)] = i[o]
    return out_s, out_i


def recall_at_k(retrieved_ids, relevant_ids) -> float:
    """|retrieved ∩ relevant| / |relevant| -- /root/reference/evaluation/retrieval/retrieval_metrics.py:49-58."""
    rel = set(int(x) for x in relevant_ids if int(x) >= 0)
    if not rel:
        return 0.0
    got = set(int(x) for x in retrieved_ids if int(x) >= 0)
    return len(got & rel) / len(rel)


def synth_corpus(n: int, d: int, seed: int = 1234) -> np.ndarray:
    """Seeded unit-normalised Gaussian corpus, fp32."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d), dtype=np.float32)
    return l2_normalize_rows(x)


def synth_queries(corpus_f32: np.ndarray, nq: int, seed: int = 4321, noise: float = 0.1) -> np.ndarray:
    """50 % planted neighbours normalise(C[j] + noise*g), 50 % independent unit vectors."""
    rng = np.random.default_rng(seed)
    n, d = corpus_f32.shape
    q = rng.standard_normal((nq, d), dtype=np.float32)
    planted = np.arange(nq) % 2 == 0
    j = rng.integers(0, n, size=nq)
    q[planted] = corpus_f32[j[planted]] + noise * q[planted]
    return l2_normalize_rows(q)
