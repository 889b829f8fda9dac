"""GPU: the exactness certificate (crs_refine_f32_cert) and the escalation (crs_escalate_exact).

north_star: "identical doc-id top-k sets" vs the reference's fp32 store (/root/reference/rag/indexing.py:114-119,171-176).
The store over-fetches from the fp16 / int8 slab and re-ranks in fp32; these tests hold the PROOF that this is the fp32
top-k of all rows, and the escalation that restores exactness when the proof fails, against the oracle fed the fp32 rows:
  * random corpora: nearly every query is certified by the over-fetch alone, and every certified list equals the oracle's;
  * adversarial near-ties (40 rows within ~1e-5 cosine of each other at the top of a 1 M-row slab, fp16 and int8): the
    over-fetch cannot hold them all, the certificate must refuse, the escalation must return the oracle's fp32 ids;
  * more exact duplicates than the escalation list holds: status 2, and the store's retry with a longer list;
  * the store's DEFAULT config carries the property.
Bar: ids identical to the oracle's, except between rows whose fp64 scores differ by less than fp32 summation order can
resolve (3e-7 at the |score| ~ 0.3 of random corpora; 2e-6 inside the near-duplicate bands, where |score| ~ 1 over 384 / 768
terms: the oracle's sgemm and the kernel's FMA chain + butterfly round differently); scores within 1e-5 of the oracle's
(north_star allows 1e-3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _slab(cuda, rows32, slab_type):
    """fp32 rows (torch, cuda) -> (slab, scales, shadow, row_err) through the product's crs::slab_append."""
    import torch
    from rag import _native as nat
    n, d = rows32.shape
    pd = nat.padded_dim(d, slab_type)
    slab = torch.empty((n, pd), dtype=torch.int8 if slab_type == nat.SLAB_I8 else torch.float16, device=cuda)
    scales = torch.empty(n, dtype=torch.float32, device=cuda) if slab_type == nat.SLAB_I8 else None
    shadow = torch.empty((n, d), dtype=torch.float32, device=cuda)
    row_err = torch.zeros(1, dtype=torch.float32, device=cuda)
    for lo in range(0, n, 250_000):
        nat.slab_append_f32(rows32[lo:lo + 250_000].contiguous(), slab, lo, slab_type, scales=scales, shadow=shadow, row_err=row_err)
    return slab, scales, shadow, row_err


def _search_exact(q32, slab, scales, shadow, row_err, n, d, slab_type, k, k_scan, cap=1024, escalate=True):
    import torch
    from rag import _native as nat
    nq = q32.shape[0]
    q16 = nat.queries_to_f16(q32, slab_type)
    cs, ci = nat.cosine_topk(q16, slab, n, d, k_scan, slab_type=slab_type, scales=scales)
    ws = torch.empty(nat.exact_workspace_bytes(nq, cap), dtype=torch.uint8, device=q32.device)
    s, i, st = nat.refine_f32_cert(q32, q16, shadow, n, 0, ci, cs, k, float(row_err.item()), slab_type, ws, cap)
    st0 = st.clone()
    if escalate:
        nat.escalate_exact(q32, q16, slab, shadow, n, 0, k, s, i, st, ws, cap, scales=scales)
    torch.cuda.synchronize()
    return s.cpu().numpy(), i.cpu().numpy(), st0.cpu().numpy(), st.cpu().numpy()


def _assert_topk(got_s, got_i, q32_h, rows_h, k, what, tol=3e-7):
    """got == the oracle's exact fp32 ranking of the fp32 rows; id differences only between fp64-near-equal rows."""
    from oracle import scan_ref
    rs, ri = scan_ref.cosine_topk_ref(q32_h, rows_h, k)
    assert np.abs(got_s - rs).max() < 1e-5, what
    bad = np.nonzero((got_i != ri).any(axis=1))[0]
    for r in bad:
        ids = np.union1d(got_i[r], ri[r])
        f64 = rows_h[ids].astype(np.float64) @ q32_h[r].astype(np.float64)
        score = dict(zip(ids.tolist(), f64.tolist()))
        kth = sorted(score.values(), reverse=True)[k - 1]
        for a, b in zip(got_i[r], ri[r]):
            if a != b:
                assert abs(score[int(a)] - score[int(b)]) < tol, f"{what}: query {r}: got row {a}, oracle row {b}"
        assert all(score[int(a)] >= kth - tol for a in got_i[r]), f"{what}: query {r} holds a row below the k-th best"


@pytest.mark.parametrize("name,n,d,slab,k_scan", [("f16-384", 1_000_000, 384, "f16", 32), ("f16-768", 400_000, 768, "f16", 32),
                                                  ("f16-100", 200_000, 100, "f16", 16)])
def test_random_corpus_is_certified_and_exact(cuda, name, n, d, slab, k_scan):
    import torch
    from rag import _native as nat
    st_ = nat.SLAB_F16
    g = torch.Generator(device=cuda); g.manual_seed(n % 1009 + d)
    rows = torch.randn((n, d), generator=g, device=cuda)
    sl, sc, shadow, row_err = _slab(cuda, rows, st_)
    nq, k = 64, 10
    q = torch.randn((nq, d), generator=g, device=cuda)
    j = torch.randint(0, n, (nq,), generator=g, device=cuda)
    q[0::2] = shadow[j[0::2]] + 0.1 * q[0::2]
    q = torch.nn.functional.normalize(q, dim=1).contiguous()
    s, i, st0, st1 = _search_exact(q, sl, sc, shadow, row_err, n, d, st_, k, k_scan)
    # the tracked row error: fp16 rounds to 2^-11 relative, so |row16 - row32| is a few 1e-4 and below the analytic bound
    assert 5e-5 < float(row_err.item()) <= nat.exact_row_error_bound(d, st_)
    assert (st0 == 0).mean() >= 0.9, f"{name}: only {(st0 == 0).mean():.2f} of the queries certified at k'={k_scan}"
    assert set(np.unique(st1)) <= {0, 1}
    _assert_topk(s, i, q.cpu().numpy(), shadow.cpu().numpy(), k, name)


@pytest.mark.parametrize("slab,n,d", [("f16", 1_000_000, 384), ("i8", 1_000_000, 768)])
def test_adversarial_near_ties_are_escalated_to_the_exact_ids(cuda, slab, n, d):
    """40 rows within ~1e-5 cosine of each other at the top (near-duplicate chunks): k' = 16 / 32 candidates cannot hold
    them, the fp16 / int8 order among them is noise.  Certificate: must refuse.  Escalation: must return the fp32 ids."""
    import torch
    from rag import _native as nat
    st_ = nat.SLAB_I8 if slab == "i8" else nat.SLAB_F16
    g = torch.Generator(device=cuda); g.manual_seed(77)
    rows = torch.randn((n, d), generator=g, device=cuda)
    rows = torch.nn.functional.normalize(rows, dim=1)
    nq, k, dup = 8, 10, 40
    centres = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=cuda), dim=1)
    where = torch.randperm(n, generator=g, device=cuda)[: nq * dup].view(nq, dup)
    for r in range(nq):
        rows[where[r]] = centres[r] + 1e-3 * torch.randn((dup, d), generator=g, device=cuda)
    sl, sc, shadow, row_err = _slab(cuda, rows, st_)
    q = torch.nn.functional.normalize(centres + 1e-4 * torch.randn((nq, d), generator=g, device=cuda), dim=1).contiguous()
    top = (shadow[where[0]].double() @ q[0].double())
    assert float(top.max() - top.min()) < 1e-4           # the band really is narrower than fp16 / int8 resolution
    for k_scan in (16, 32):
        s, i, st0, st1 = _search_exact(q, sl, sc, shadow, row_err, n, d, st_, k, k_scan)
        assert (st0 == 1).all(), f"{slab} k'={k_scan}: a query with 40 near-ties at the top was certified"
        assert (st1 == 1).all()                             # escalated, no overflow
        _assert_topk(s, i, q.cpu().numpy(), shadow.cpu().numpy(), k, f"{slab} k'={k_scan}", tol=2e-6)
        # and the un-escalated re-rank really is wrong here (the test would be vacuous otherwise)
        s_, i_, _, _ = _search_exact(q, sl, sc, shadow, row_err, n, d, st_, k, k_scan, escalate=False)
        assert (np.sort(i_, 1) != np.sort(i, 1)).any()


def test_small_shards_and_partial_batches(cuda):
    """n_rows <= k': every row was fetched -> certified without a bound; escalation of SOME queries leaves the others alone."""
    import torch
    from rag import _native as nat
    g = torch.Generator(device=cuda); g.manual_seed(3)
    d = 384
    for n in (5, 16, 33):
        rows = torch.randn((n, d), generator=g, device=cuda)
        sl, sc, shadow, row_err = _slab(cuda, rows, nat.SLAB_F16)
        q = torch.nn.functional.normalize(torch.randn((7, d), generator=g, device=cuda), dim=1).contiguous()
        s, i, st0, st1 = _search_exact(q, sl, sc, shadow, row_err, n, d, nat.SLAB_F16, min(10, n), 16)
        if n <= 16:
            assert (st0 == 0).all()
        _assert_topk(s, i, q.cpu().numpy(), shadow.cpu().numpy(), min(10, n), f"n={n}")
    # mixed batch: queries 0..3 sit on 30 near-duplicates, queries 4..19 are random
    n = 200_000
    rows = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=cuda), dim=1)
    centres = torch.nn.functional.normalize(torch.randn((4, d), generator=g, device=cuda), dim=1)
    for r in range(4):
        rows[1000 * (r + 1): 1000 * (r + 1) + 30] = centres[r] + 1e-3 * torch.randn((30, d), generator=g, device=cuda)
    sl, sc, shadow, row_err = _slab(cuda, rows, nat.SLAB_F16)
    q = torch.cat([centres, torch.nn.functional.normalize(torch.randn((16, d), generator=g, device=cuda), dim=1)]).contiguous()
    s, i, st0, st1 = _search_exact(q, sl, sc, shadow, row_err, n, d, nat.SLAB_F16, 10, 16)
    assert (st0[:4] == 1).all() and (st0[4:] == 0).mean() > 0.8
    _assert_topk(s, i, q.cpu().numpy(), shadow.cpu().numpy(), 10, "mixed batch", tol=2e-6)


def test_list_overflow_sets_status_2_and_a_longer_list_resolves_it(cuda):
    import torch
    from rag import _native as nat
    g = torch.Generator(device=cuda); g.manual_seed(9)
    n, d, ndup = 100_000, 384, 3000
    rows = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=cuda), dim=1)
    v = torch.nn.functional.normalize(torch.randn((1, d), generator=g, device=cuda), dim=1)
    pos = torch.randperm(n, generator=g, device=cuda)[:ndup]
    rows[pos] = v                                            # 3000 identical chunks (boilerplate pages)
    sl, sc, shadow, row_err = _slab(cuda, rows, nat.SLAB_F16)
    q = torch.cat([v, torch.nn.functional.normalize(torch.randn((3, d), generator=g, device=cuda), dim=1)]).contiguous()
    s, i, st0, st1 = _search_exact(q, sl, sc, shadow, row_err, n, d, nat.SLAB_F16, 10, 32, cap=1024)
    assert st0[0] == 1 and st1[0] == 2                       # 3000 rows in the band, 1024 slots
    s, i, st0, st1 = _search_exact(q, sl, sc, shadow, row_err, n, d, nat.SLAB_F16, 10, 32, cap=4096)
    assert st1[0] == 1
    assert np.array_equal(i[0], np.sort(pos.cpu().numpy())[:10])      # exact ties: the ten lowest ids
    _assert_topk(s, i, q.cpu().numpy(), shadow.cpu().numpy(), 10, "3000 duplicates")


@pytest.mark.parametrize("dtype", ["fp16", "int8"])
def test_store_default_config_returns_the_fp32_ranking(cuda, dtype):
    """VectorStore with the DEFAULT refine settings (an unmodified reference config.json only sets collection_name /
    persist_directory): fp32 shadow kept, over-fetch, certificate, escalation (fp16; int8 with refine_exact=True) --
    the ids of search_batch are the oracle's fp32 ids, also across a band of near-duplicate chunks."""
    import torch
    from rag.chunking import Chunk
    from rag.indexing import VectorStore
    cfg = {"collection_name": "exact"} if dtype == "fp16" else {"collection_name": "exact8", "index_dtype": "int8", "refine_exact": True}
    store = VectorStore(cfg)
    assert store.refine_fp32 is True and store.refine_overfetch == 24      # (16 of them on shards below 4 M rows: nat.overfetch)
    rng = np.random.default_rng(11)
    n, d = 120_000, 384
    emb = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal(d).astype(np.float32)
    centre /= np.linalg.norm(centre)
    dup = rng.choice(n, 40, replace=False)
    emb[dup] = centre + 1e-3 * rng.standard_normal((40, d)).astype(np.float32)
    chunks = [Chunk(text=f"t{r}", chunk_id=f"chunk_{r}", start_char=0, end_char=1) for r in range(n)]
    for lo in range(0, n, 50_000):
        store.create_index(chunks[lo:lo + 50_000], emb[lo:lo + 50_000])
    q = rng.standard_normal((12, d)).astype(np.float32)
    q[0] = centre
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    res = store.search_batch(q, top_k=10)
    ex = store.last_exactness
    assert ex["queries"] == 12 and ex["escalated"] >= 1 and ex["unproven"] == 0 and ex["certified"] + ex["escalated"] == 12
    rows_h = store.collection.shadow[:n].cpu().numpy()
    got_i = np.array([[int(c.split("_")[1]) for c in row] for row in res["ids"]])
    got_s = np.array([[1.0 - x for x in row] for row in res["distances"]], dtype=np.float32)
    _assert_topk(got_s, got_i, q, rows_h, 10, f"store {dtype}", tol=2e-6)
    assert set(got_i[0]) <= set(dup.tolist())
    # opting out: the plain slab ranking, no shadow, no certificate
    plain = VectorStore({"collection_name": "plain", "refine_fp32": False, "index_dtype": dtype})
    plain.create_index(chunks[:1000], emb[:1000])
    assert plain.collection.shadow is None
    plain.search_batch(q, top_k=5)
    assert plain.last_exactness["queries"] == 0
