"""GPU, BASELINE-size shards: properties that do not need the (slow) CPU oracle.

  * sortedness / uniqueness / id range of every returned list;
  * score check: the returned score of every id equals a direct fp32 dot product of that slab row;
  * split consistency: top-k(full shard) == merge(top-k(first half), top-k(second half))  (scan + id_base + merge);
  * idempotence: the same call twice is bit-identical;
  * planted-neighbour round trip: a row equal to the query itself comes back first with score ~1.
Data is generated on the device with torch (seeded); nothing from oracle/ is involved."""
import pytest

pytestmark = pytest.mark.gpu


def _make(cuda, n, d, nq, slab_type, seed):
    import torch
    from rag import _native as nat
    g = torch.Generator(device=cuda); g.manual_seed(seed)
    pd = nat.padded_dim(d, slab_type)
    slab = torch.empty((n, pd), dtype=torch.int8 if slab_type == nat.SLAB_I8 else torch.float16, device=cuda)
    scales = torch.empty(n, dtype=torch.float32, device=cuda) if slab_type == nat.SLAB_I8 else None
    for lo in range(0, n, 250_000):
        m = min(250_000, n - lo)
        nat.slab_append_f32(torch.randn((m, d), generator=g, device=cuda), slab, lo, slab_type, scales=scales)
    q32 = torch.randn((nq, d), generator=g, device=cuda)
    # plant: row 7*i+3 becomes (a re-quantised copy of) query i
    rows = torch.arange(nq, device=cuda) * 7 + 3
    tmp = torch.empty((nq, pd), dtype=slab.dtype, device=cuda)
    tsc = torch.empty(nq, dtype=torch.float32, device=cuda) if scales is not None else None
    nat.slab_append_f32(q32.contiguous(), tmp, 0, slab_type, scales=tsc)
    slab[rows] = tmp
    if scales is not None:
        scales[rows] = tsc
    return slab, scales, q32, rows


@pytest.mark.parametrize("name,n,d,nq,k,slab", [
    ("c2", 100_000, 384, 64, 10, "f16"),
    ("c4-shard", 1_250_000, 384, 64, 10, "f16"),
    ("c3", 1_000_000, 768, 256, 10, "f16"),
    ("c5-shard", 1_250_000, 768, 64, 10, "i8"),
    ("c4-k40", 1_250_000, 384, 16, 40, "f16"),
    # 16 < k <= 64 on long streams: the 32- / 64-slot chains of the tile-best kernels (k = 40 above: the 64-slot one)
    ("c4-shard-k17", 1_250_000, 384, 64, 17, "f16"),
    ("c4-shard-k32", 1_250_000, 384, 64, 32, "f16"),
    ("c5-shard-k24", 1_250_000, 768, 64, 24, "i8"),
    ("c4-10M-k32", 10_000_000, 384, 64, 32, "f16"),
    # the WHOLE corpora of BASELINE configs[3] / [4] on one GPU (what bench.py's default N = 1 line scans)
    ("c4-10M", 10_000_000, 384, 64, 16, "f16"),
    ("c5-10M", 10_000_000, 768, 64, 16, "i8"),
])
def test_fullsize_properties(cuda, name, n, d, nq, k, slab):
    import torch
    from rag import _native as nat
    st = nat.SLAB_I8 if slab == "i8" else nat.SLAB_F16
    sl, sc, q32, planted = _make(cuda, n, d, nq, st, seed=len(name))
    q16 = nat.queries_to_f16(q32, st)
    s, i = nat.cosine_topk(q16, sl, n, d, k, slab_type=st, scales=sc)
    s2, i2 = nat.cosine_topk(q16, sl, n, d, k, slab_type=st, scales=sc)
    torch.cuda.synchronize()
    assert torch.equal(s, s2) and torch.equal(i, i2)                                   # idempotent
    assert bool(((i >= 0) & (i < n)).all())
    assert bool((s[:, :-1] >= s[:, 1:]).all())                                          # sorted
    tie = s[:, :-1] == s[:, 1:]
    assert bool((i[:, :-1][tie] < i[:, 1:][tie]).all())                                 # ties: lower id first
    assert all(len(set(r)) == k for r in i.tolist())                                    # unique
    # returned scores are the dot products of the returned rows
    rows = sl[i.reshape(-1), :d].float().view(nq, k, d)
    if sc is not None:
        rows = rows * sc[i.reshape(-1)].view(nq, k, 1)
    direct = torch.einsum("qkd,qd->qk", rows, q16[:, :d].float())
    assert float((direct - s).abs().max()) < (2e-3 if slab == "i8" else 2e-5)
    # planted copy of the query comes back first with cosine ~ 1
    assert torch.equal(i[:, 0], planted)
    assert float((1.0 - s[:, 0]).abs().max()) < (2e-2 if slab == "i8" else 2e-3)
    # split consistency through id_base + the cross-shard merge kernel
    h = (n // 2 // 64) * 64 + 17
    sa, ia = nat.cosine_topk(q16, sl[:h], h, d, k, slab_type=st, scales=None if sc is None else sc[:h])
    sb, ib = nat.cosine_topk(q16, sl[h:], n - h, d, k, slab_type=st, scales=None if sc is None else sc[h:], id_base=h)
    ms, mi = nat.merge_topk(torch.stack([sa, sb]), torch.stack([ia, ib]), k)
    torch.cuda.synchronize()
    assert torch.equal(mi, i) and torch.equal(ms, s)
