"""GPU parity: crs_cosine_topk (HIP, through the C ABI) vs oracle/scan_ref.py on seeded inputs."""
import numpy as np
import pytest

from oracle import scan_ref
from topk_check import check_topk

pytestmark = pytest.mark.gpu


def _run(cuda, q16_np, slab_np, k, id_base=0):
    import torch
    from rag import _native as nat
    nq, d = q16_np.shape
    n = slab_np.shape[0]
    pd = nat.padded_dim(d)
    q = torch.zeros((nq, pd), dtype=torch.float16)
    q[:, :d] = torch.from_numpy(q16_np)
    s = torch.zeros((n, pd), dtype=torch.float16)
    s[:, :d] = torch.from_numpy(slab_np)
    q = q.to(cuda); s = s.to(cuda)
    sc, ids = nat.cosine_topk(q, s, n, d, k, id_base=id_base)
    torch.cuda.synchronize()
    return sc.cpu().numpy(), ids.cpu().numpy()


def _case(n, d, nq, seed=0):
    c = scan_ref.synth_corpus(n, d, seed=1234 + seed)
    q = scan_ref.synth_queries(c, nq, seed=4321 + seed)
    return q.astype(np.float16), c.astype(np.float16)


@pytest.mark.parametrize("n,d,nq,k", [
    (4096, 384, 8, 10),      # golden-vector shape G2
    (5000, 768, 64, 10),     # ragged last tile, bge width
    (1000, 384, 1, 3),       # the reference's own call shape: one query, top_k=3
    (3001, 384, 70, 6),      # two query blocks, k = 2*top_k
    (2048, 128, 16, 1),
    (2500, 384, 33, 16),     # k at the edge of the 16-wide selector
    (2500, 384, 20, 17),     # first k on the 64-wide selector
    (6000, 384, 64, 40),     # ablation top_k=20 with rerank -> 40
    (3000, 768, 5, 64),
    (777, 100, 9, 5),        # dim padded 100 -> 128
    (100, 1024, 3, 10),
    (20000, 256, 64, 10),
])
def test_scan_matches_oracle(cuda, n, d, nq, k):
    q, c = _case(n, d, nq, seed=n % 7)
    gs, gi = _run(cuda, q, c, k)
    full = scan_ref.full_scores_f64(q, c)
    check_topk(gs, gi, full, k)
    # and the ids agree with the oracle's own top-k outside near-tie bands
    rs, ri = scan_ref.cosine_topk_ref(q, c, k)
    same = (gi == ri).mean()
    assert same > 0.99, f"only {same:.3f} of ids identical to the oracle"


@pytest.mark.parametrize("n,k", [(1, 10), (5, 10), (16, 16), (33, 40)])
def test_fewer_rows_than_k(cuda, n, k):
    q, c = _case(n, 384, 4)
    gs, gi = _run(cuda, q, c, k)
    check_topk(gs, gi, scan_ref.full_scores_f64(q, c), k)


def test_all_rows_identical_lower_id_first(cuda):
    # every score ties exactly: the tie rule alone decides -> ids 0..k-1 in order
    v = scan_ref.synth_corpus(1, 384, seed=7).astype(np.float16)
    c = np.repeat(v, 3000, axis=0)
    q = scan_ref.synth_corpus(3, 384, seed=8).astype(np.float16)
    for k in (10, 40):
        gs, gi = _run(cuda, q, c, k)
        assert np.array_equal(gi, np.tile(np.arange(k), (3, 1)))
        assert (gs == gs[:, :1]).all()


def test_planted_duplicates_and_ties(cuda):
    q, c = _case(4096, 384, 8)
    c = c.copy()
    # duplicate the best row of query 0 at several later positions, and an early one too
    full = scan_ref.full_scores_f64(q, c)
    best = int(full[0].argmax())
    for pos in (best + 1 if best + 1 < 4096 else 5, 17, 4000, 4095):
        c[pos] = c[best]
    gs, gi = _run(cuda, q, c, 10)
    full = scan_ref.full_scores_f64(q, c)
    check_topk(gs, gi, full, 10)
    rs, ri = scan_ref.cosine_topk_ref(q, c, 10)
    assert np.array_equal(gi[0][:4], ri[0][:4])  # the exact duplicates, in id order


def test_ascending_scores_worst_case_for_threshold(cuda):
    # rows sorted so that scores for query 0 only ever increase: every row passes the running
    # threshold and the compaction path runs constantly
    q, c = _case(6000, 384, 16)
    full = scan_ref.full_scores_f64(q, c)
    c = c[np.argsort(full[0])]
    for k in (10, 33):
        gs, gi = _run(cuda, q, c, k)
        check_topk(gs, gi, scan_ref.full_scores_f64(q, c), k)


def test_id_base_and_unnormalised_scores(cuda):
    q, c = _case(3000, 384, 4)
    gs, gi = _run(cuda, q, c, 7, id_base=1_000_000_007)
    check_topk(gs, gi, scan_ref.full_scores_f64(q, c), 7, id_base=1_000_000_007)


def test_c2_full_size(cuda):
    """BASELINE config #2: 100k x 384 fp16, 64 queries, k=10."""
    q, c = _case(100_000, 384, 64)
    gs, gi = _run(cuda, q, c, 10)
    rs, ri = scan_ref.cosine_topk_ref(q, c, 10, accumulate=np.float64)
    assert np.abs(gs - rs).max() < 2e-5
    mism = gi != ri
    # any id mismatch must be a near-tie in score
    assert (np.abs(gs - rs)[mism] < 4e-6).all()
    for r in range(64):
        assert scan_ref.recall_at_k(gi[r], ri[r]) >= 0.9
    assert np.mean([scan_ref.recall_at_k(gi[r], ri[r]) for r in range(64)]) > 0.995


@pytest.mark.parametrize("n,d,nq,k,kind", [(300_000, 384, 64, 20, "f16"), (300_000, 384, 7, 32, "f16"), (300_000, 768, 64, 32, "i8"),
                                            (300_000, 128, 64, 17, "f16"), (300_000, 384, 33, 40, "f16"), (300_000, 256, 64, 64, "f16"),
                                            (160_000, 768, 64, 40, "f16"), (300_000, 768, 64, 40, "i8"), (300_000, 384, 64, 56, "f16"),
                                            (500_000, 512, 64, 48, "f16")])
def test_long_chain_matches_oracle_and_threshold_kernels(cuda, n, d, nq, k, kind):
    """16 < k <= 64 on streams too long for the dump form: scan_tb / scan_i8 with a 32- / 40- / 48- / 56- / 64-slot chain.  Checked against the
    oracle; test_scan_classic_gpu.py re-runs this module on the threshold kernels (CRS_SCAN_TB=0)."""
    import torch
    from oracle import scan_ref
    from rag import _native as nat
    c = scan_ref.synth_corpus(n, d, seed=31)
    q = scan_ref.synth_queries(c, nq, seed=32)
    st = nat.SLAB_I8 if kind == "i8" else nat.SLAB_F16
    pd = nat.padded_dim(d, st)
    slab = torch.zeros((n, pd), dtype=torch.int8 if kind == "i8" else torch.float16, device=cuda)
    scales = torch.zeros(n, dtype=torch.float32, device=cuda) if kind == "i8" else None
    nat.slab_append_f32(torch.from_numpy(c).to(cuda), slab, 0, st, scales=scales)
    q16 = nat.queries_to_f16(torch.from_numpy(q).to(cuda), st)
    import os
    if os.environ.get("CRS_SCAN_TB", "1") != "0" and os.environ.get("CRS_SCAN_LONG_CHAIN", "1") != "0":
        # chain lengths whose registers fit without scratch (tools/check_resources.py): 64 slots for 256-element fp16 rows, 48 / 56
        # for 384, 48 for 512 / 640, 40 for 768; int8: 48 up to 768
        if k <= 24 and kind == "f16":
            want = ",24>"
        elif k <= 32:
            want = ",32>"
        elif kind == "i8":
            want = ",48>"
        else:
            want = {256: ",64>", 384: ",48>" if k <= 48 else ",56>", 512: ",48>", 640: ",48>", 768: ",40>"}[pd]
        assert want in nat.scan_plan_describe(nq, d, k, n, slab_type=st)                # the long-chain plan
    s, i = nat.cosine_topk(q16, slab, n, d, k, slab_type=st, scales=scales)
    torch.cuda.synchronize()
    if kind == "f16":
        full = scan_ref.full_scores_f64(q16.cpu().numpy(), slab.cpu().numpy())
        check_topk(s.cpu().numpy(), i.cpu().numpy(), full, k)
    else:
        qd = scan_ref.dequantized_queries(q16.cpu().numpy())
        full = scan_ref.full_scores_f64(qd, slab.cpu().numpy(), scales.cpu().numpy())
        got = i.cpu().numpy()
        for r in range(nq):
            assert np.abs(full[r, got[r]] - s[r].cpu().numpy()).max() < 1e-5
            assert full[r, got[r]].min() >= np.sort(full[r])[-k] - 1e-6
