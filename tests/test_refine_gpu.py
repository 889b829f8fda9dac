"""GPU: over-fetch + fp32 re-rank (crs_refine_f32) and the one-collective wire layout (crs_merge_topk_wire).

The north-star recall is against the ranking of the UNQUANTISED fp32 rows (what the reference's ChromaDB
collection stores, /root/reference/rag/indexing.py:114-119), not against the quantised slab: these tests
feed the oracle the fp32 rows and demand identical ids from `scan(k'=16) -> refine_f32 -> top 10`, for the
fp16 and the int8 slab, at >= 1M rows.  Tolerance: ids identical, scores within 1e-5 (fp32 dot vs the
oracle's fp32 sgemm; north_star allows 1e-3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(cuda, n, d, slab_type, seed, nq):
    import torch
    from rag import _native as nat
    g = torch.Generator(device=cuda); g.manual_seed(seed)
    pd = nat.padded_dim(d, slab_type)
    slab = torch.empty((n, pd), dtype=torch.int8 if slab_type == nat.SLAB_I8 else torch.float16, device=cuda)
    scales = torch.empty(n, dtype=torch.float32, device=cuda) if slab_type == nat.SLAB_I8 else None
    shadow = torch.empty((n, d), dtype=torch.float32, device=cuda)
    for lo in range(0, n, 250_000):
        m = min(250_000, n - lo)
        nat.slab_append_f32(torch.randn((m, d), generator=g, device=cuda), slab, lo, slab_type, scales=scales, shadow=shadow)
    # queries: half planted near a row, half random
    q = torch.randn((nq, d), generator=g, device=cuda)
    j = torch.randint(0, n, (nq,), generator=g, device=cuda)
    q[0::2] = shadow[j[0::2]] + 0.1 * q[0::2]
    q = torch.nn.functional.normalize(q, dim=1).contiguous()
    return slab, scales, shadow, q


@pytest.mark.parametrize("name,n,d,slab", [
    ("f16-384", 1_250_000, 384, "f16"),
    ("i8-768", 1_000_000, 768, "i8"),
    ("i8-384", 300_000, 384, "i8"),
])
def test_overfetch_plus_refine_equals_exact_fp32_ranking(cuda, name, n, d, slab):
    import torch
    from oracle import scan_ref
    from rag import _native as nat
    st = nat.SLAB_I8 if slab == "i8" else nat.SLAB_F16
    nq, k, k_scan = 64, 10, 16
    sl, sc, shadow, q32 = _build(cuda, n, d, st, seed=len(name) + n % 97, nq=nq)
    q16 = nat.queries_to_f16(q32, st)
    cs, ci = nat.cosine_topk(q16, sl, n, d, k_scan, slab_type=st, scales=sc)
    s, i = nat.refine_f32(q32, shadow, n, 0, ci, k)
    s0, i0 = nat.cosine_topk(q16, sl, n, d, k, slab_type=st, scales=sc)          # the unrefined answer, for contrast
    torch.cuda.synchronize()
    rs, ri = scan_ref.cosine_topk_ref(q32.cpu().numpy(), shadow.cpu().numpy(), k)   # oracle over the fp32 rows
    got_i, got_s = i.cpu().numpy(), s.cpu().numpy()
    assert np.array_equal(got_i, ri), f"{name}: refined ids differ from the exact fp32 ranking in {int((got_i != ri).any(1).sum())} queries"
    assert np.abs(got_s - rs).max() < 1e-5
    rec0 = np.mean([scan_ref.recall_at_k(a, b) for a, b in zip(i0.cpu().numpy(), ri)])
    assert rec0 > (0.80 if slab == "i8" else 0.97)          # the quantised scan alone is close but (int8) not exact
    if slab == "i8":
        assert rec0 < 1.0


def test_refine_f32_small_cases_vs_numpy(cuda):
    import torch
    from rag import _native as nat
    rng = np.random.default_rng(5)
    n, d, nq, k_in, k_out = 500, 100, 7, 16, 5
    shadow = rng.standard_normal((n, d)).astype(np.float32)
    shadow[10] = shadow[11]                                         # an exact tie: lower id first
    q = rng.standard_normal((nq, d)).astype(np.float32)
    cand = rng.integers(0, n, size=(nq, k_in)).astype(np.int64)
    cand[0, :2] = (11, 10)
    cand[1, 3:] = -1                                                # empty slots
    cand[2, :] = -1                                                 # nothing at all
    cand[3, 5] = 100_000                                            # outside the shard: ignored
    id_base = 1000
    s, i = nat.refine_f32(torch.from_numpy(q).to(cuda), torch.from_numpy(shadow).to(cuda), n, id_base,
                          torch.from_numpy(np.where(cand >= 0, cand + id_base, cand)).to(cuda), k_out)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    for r in range(nq):
        ok = [(c, float(q[r].astype(np.float64) @ shadow[c].astype(np.float64))) for c in cand[r] if 0 <= c < n]
        # duplicates among the candidates are kept as separate entries (the scan never produces them)
        ok.sort(key=lambda t: (-t[1], t[0]))
        want = ok[:k_out]
        for slot in range(k_out):
            if slot < len(want):
                assert i[r, slot] == want[slot][0] + id_base or abs(s[r, slot] - want[slot][1]) < 1e-5
                assert abs(s[r, slot] - want[slot][1]) < 1e-4
            else:
                assert i[r, slot] == -1 and s[r, slot] == -np.inf
    assert list(i[0, :2]) == sorted(i[0, :2]) or s[0, 0] > s[0, 1]


@pytest.mark.parametrize("world,nq,k", [(2, 64, 10), (8, 64, 10), (8, 512, 10), (3, 5, 3)])
def test_wire_merge_equals_oracle_merge(cuda, world, nq, k):
    import torch
    from oracle import scan_ref
    from rag import _native as nat
    rng = np.random.default_rng(world * 1000 + nq)
    blocks = [nat.WireBlock(nq, k, cuda, 1) for _ in range(world)]
    sc = np.sort(rng.standard_normal((world, nq, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    ids = rng.integers(0, 1 << 40, size=(world, nq, k)).astype(np.int64)
    sc[1, 0, :] = sc[0, 0, :]                                        # cross-rank ties
    ids[0, 1, k - 1] = -1; sc[0, 1, k - 1] = -np.inf                 # an empty slot
    for w, b in enumerate(blocks):
        b.scores.copy_(torch.from_numpy(sc[w])); b.ids.copy_(torch.from_numpy(ids[w]))
    gathered = torch.cat([b.buf for b in blocks])                     # what all_gather_into_tensor delivers
    assert gathered.numel() == world * blocks[0].nbytes
    s, i = nat.merge_topk_wire(gathered, world, nq, k, k)
    rs, ri = scan_ref.merge_topk_ref(sc, ids, k)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
