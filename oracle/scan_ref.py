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


def _select_block(cat_s: np.ndarray, cat_i: np.ndarray, kk: int):
    """Per row the kk best of (cat_s, cat_i) in (score desc, id asc) order, vectorised over the rows.
    Precondition (kept by cosine_topk_ref): within a row, equal scores appear in ascending id order from left to
    right -- so among ties the left-most entries are the ones to keep and a stable sort finishes the order."""
    nq, m = cat_s.shape
    if m <= kk:
        o = np.argsort(-cat_s.astype(np.float64), axis=1, kind="stable")
        return np.take_along_axis(cat_s, o, 1), np.take_along_axis(cat_i, o, 1)
    kth = np.partition(cat_s, m - kk, axis=1)[:, m - kk]              # the kk-th largest value per row (O(m))
    rows, cols = np.nonzero(cat_s >= kth[:, None])                    # row-major: cols ascending inside a row; >= kk per row
    sc = cat_s[rows, cols]
    o = np.lexsort((cols, -sc.astype(np.float64), rows))              # row, then score desc, then position (= id) asc
    rows, cols, sc = rows[o], cols[o], sc[o]
    first = np.searchsorted(rows, np.arange(nq))                      # start of every row's run
    take = (first[:, None] + np.arange(kk)[None, :]).reshape(-1)
    return sc[take].reshape(nq, kk), cat_i[rows[take], cols[take]].reshape(nq, kk)


def cosine_topk_ref(q: np.ndarray, slab: np.ndarray, k: int, *, scales: np.ndarray | None = None,
                    id_base: int = 0, block: int = 1 << 16, accumulate=np.float32, timing: dict | None = None):
    """Exact top-k inner product of ``q[nq,d]`` against ``slab[n,d]`` (rows are unit vectors,
    so the inner product is the cosine).  Inputs are taken *as stored* (fp16 / int8) and
    widened; ``scales`` (fp32 per row) multiplies int8 rows' dot products.

    Blocked: scores of a row block (BLAS sgemm), then per query the k best of (running list + block) --
    an O(block) partition to the k-th value, ties resolved (score desc, id asc) on the few survivors.
    ``timing`` (optional dict) receives the seconds spent in 'gemm' and in 'select'.

    Returns (scores fp32 [nq,k], ids int64 [nq,k]); slots beyond ``n`` hold (-inf, -1).
    """
    import time
    q = np.asarray(q)
    nq = q.shape[0]
    n = slab.shape[0]
    kk = min(k, n)
    best_s = np.full((nq, 0), NEG_INF, dtype=np.float32)
    best_i = np.zeros((nq, 0), dtype=np.int64)
    qa = q.astype(accumulate)
    t_gemm = t_sel = 0.0
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        t0 = time.perf_counter()
        # (rows x d) @ (d x nq), transposed afterwards: the same dot products as q @ rows^T, but BLAS gets a tall
        # row-major left operand (the other orientation runs ~7x slower in OpenBLAS at these shapes)
        s = np.ascontiguousarray((slab[lo:hi].astype(accumulate, copy=False) @ qa.T).T)
        if scales is not None:
            s = s * scales[lo:hi].astype(accumulate)[None, :]
        s = s.astype(np.float32, copy=False)
        t1 = time.perf_counter()
        # running list first (ids < lo, sorted score desc / id asc), then the block in id order: ties ascend left to right
        cat_s = np.concatenate([best_s, s], axis=1)
        cat_i = np.concatenate([best_i, np.broadcast_to(np.arange(lo, hi, dtype=np.int64), (nq, hi - lo))], axis=1)
        best_s, best_i = _select_block(cat_s, cat_i, kk)
        t_gemm += t1 - t0
        t_sel += time.perf_counter() - t1
    if timing is not None:
        timing["gemm"] = timing.get("gemm", 0.0) + t_gemm
        timing["select"] = timing.get("select", 0.0) + t_sel
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
