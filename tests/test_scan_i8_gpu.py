"""GPU parity: the int8 slab scan (crs_cosine_topk with CRS_SLAB_I8) vs the oracle.

The kernel searches with the query moved to 16-bit fixed point (oracle/scan_ref.quantize_query_fx16
restates that step), so oracle and kernel compute the same integer dot products; what remains is the
fp32 rounding of the final scaling, and the checks are as tight as for the fp16 slab."""
import numpy as np
import pytest

from oracle import scan_ref
from topk_check import check_topk

pytestmark = pytest.mark.gpu


def _run(cuda, q16_np, slab_i8, scales, k, id_base=0):
    import torch
    from rag import _native as nat
    nq, d = q16_np.shape
    n = slab_i8.shape[0]
    pd = nat.padded_dim(d, nat.SLAB_I8)
    q = torch.zeros((nq, pd), dtype=torch.float16); q[:, :d] = torch.from_numpy(q16_np)
    s = torch.zeros((n, pd), dtype=torch.int8); s[:, :d] = torch.from_numpy(slab_i8)
    sc, ids = nat.cosine_topk(q.to(cuda), s.to(cuda), n, d, k, slab_type=nat.SLAB_I8,
                              scales=torch.from_numpy(scales).to(cuda), id_base=id_base)
    torch.cuda.synchronize()
    return sc.cpu().numpy(), ids.cpu().numpy()


def _case(n, d, nq, seed=0):
    c = scan_ref.synth_corpus(n, d, seed=1234 + seed)
    q = scan_ref.synth_queries(c, nq, seed=4321 + seed).astype(np.float16)
    c8, sc = scan_ref.quantize_rows_i8(c)
    return q, c8, sc


@pytest.mark.parametrize("n,d,nq,k", [(4096, 768, 8, 10), (5001, 768, 64, 10), (3000, 384, 33, 6), (2000, 256, 1, 3),
                                      (2500, 768, 20, 17), (3000, 1024, 5, 40), (777, 100, 9, 5), (7, 768, 3, 10)])
def test_i8_scan_matches_oracle(cuda, n, d, nq, k):
    q, c8, sc = _case(n, d, nq, seed=n % 5)
    gs, gi = _run(cuda, q, c8, sc, k)
    full = scan_ref.full_scores_f64(scan_ref.dequantized_queries(q), c8, sc)
    check_topk(gs, gi, full, k)
    # against the *unquantised* query the scores still agree to 1e-3 (north_star tolerance)
    exact = scan_ref.full_scores_f64(q, c8, sc)
    kk = min(k, n)
    assert np.abs(np.take_along_axis(exact, gi[:, :kk], 1) - gs[:, :kk]).max() < 1e-3


def test_i8_ties_lower_id_first(cuda):
    q, c8, sc = _case(3000, 768, 4)
    full = scan_ref.full_scores_f64(scan_ref.dequantized_queries(q), c8, sc)
    best = int(full[0].argmax())
    for pos in (3, 1500, 2999):
        c8[pos] = c8[best]; sc[pos] = sc[best]
    gs, gi = _run(cuda, q, c8, sc, 10)
    full = scan_ref.full_scores_f64(scan_ref.dequantized_queries(q), c8, sc)
    check_topk(gs, gi, full, 10)
    dup = sorted({3, 1500, 2999, best})
    assert list(gi[0][:len(dup)]) == dup


def test_i8_recall_vs_exact_fp32(cuda):
    """What int8 compression costs: Recall@10 against the exact fp32 ranking (reported, not 1.0)."""
    c = scan_ref.synth_corpus(20000, 768, seed=2)
    q32 = scan_ref.synth_queries(c, 32, seed=3)
    c8, sc = scan_ref.quantize_rows_i8(c)
    gs, gi = _run(cuda, q32.astype(np.float16), c8, sc, 10)
    rs, ri = scan_ref.cosine_topk_ref(q32, c, 10, accumulate=np.float64)
    rec = np.mean([scan_ref.recall_at_k(gi[r], ri[r]) for r in range(32)])
    assert rec > 0.85
    assert np.abs(gs[:, 0] - rs[:, 0]).max() < 5e-3


def test_store_int8_end_to_end(cuda):
    from rag.indexing import VectorStore
    from rag.chunking import Chunk
    emb = scan_ref.synth_corpus(1200, 384, seed=6)
    chunks = [Chunk(f"text {i}", f"chunk_{i}", 0, 1) for i in range(1200)]
    st = VectorStore({"index_dtype": "int8"})
    st.create_index(chunks, emb)
    assert st.get_stats()["index_dtype"] == "int8"
    q = scan_ref.synth_queries(emb, 4, seed=7)
    c8, sc = scan_ref.quantize_rows_i8(emb)
    for i in range(4):
        got = st.search(q[i], top_k=5)
        full = scan_ref.full_scores_f64(scan_ref.dequantized_queries(q[i:i + 1].astype(np.float16)), c8, sc)[0]
        exp = np.lexsort((np.arange(1200), -full))[:5]
        assert got["ids"][0][0] == f"chunk_{exp[0]}"
        assert len(set(got["ids"][0]) & {f"chunk_{e}" for e in exp}) >= 4   # device-side quantisation may differ by 1 LSB


@pytest.mark.parametrize("n,d,nq,k", [
    (400_000, 768, 64, 10),    # chain mode (long streams), C5's row width
    (300_000, 256, 16, 4),     # chain, 4 slots
    (60_000, 768, 130, 16),    # several query blocks, 16 slots
    (20_000, 512, 64, 10),     # dump mode
    (20_000, 256, 64, 1),      # dump mode with more tiles per stream than k: partial lists wider than k
    (70_000, 768, 40, 3),      #   (found by tools/fuzz_scan.py: the default workspace was sized for the fp16 plan only)
])
def test_i8_tile_best_modes(cuda, n, d, nq, k):
    """scan_i8.hip's tile-best modes + refine_i8_kernel at sizes that select each of them."""
    q, c8, sc = _case(n, d, nq, seed=(n + d) % 3)
    gs, gi = _run(cuda, q, c8, sc, k)
    rs, ri = scan_ref.cosine_topk_ref(scan_ref.dequantized_queries(q), c8, k, scales=sc, accumulate=np.float64)
    assert np.abs(gs - rs).max() < 2e-5
    mism = gi != ri
    assert (np.abs(gs - rs)[mism] < 4e-6).all()
    assert np.mean([scan_ref.recall_at_k(gi[r], ri[r]) for r in range(nq)]) > 0.995


def test_i8_chain_mode_ties(cuda):
    q, c8, sc = _case(400_000, 768, 4, seed=1)
    full0 = (c8.astype(np.float32) @ scan_ref.dequantized_queries(q)[0].astype(np.float32)) * sc
    best = int(full0.argmax())
    planted = sorted({best, 5, 40, 70_001, 250_000, 399_998})
    for pos in planted:
        c8[pos] = c8[best]; sc[pos] = sc[best]
    gs, gi = _run(cuda, q, c8, sc, 10)
    assert list(gi[0][:len(planted)]) == planted
    assert (gs[0][:len(planted)] == gs[0][0]).all()
