"""GPU parity of the large-batch scan (csrc/scan_wide.hip: 65+ queries per launch, k <= 16) against
oracle/scan_ref.py -- the same acceptance check as test_scan_gpu.py, on the batch sizes that select
each configuration: 4 waves (65..128 queries), 8 waves (129..256), several 256-query blocks (> 256),
plus the tie / duplicate / adversarial-order cases the 64-query kernel is tested on."""
import numpy as np
import pytest

from oracle import scan_ref
from topk_check import check_topk
from test_scan_gpu import _case, _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,nq,k", [
    (3001, 384, 65, 10),     # 4 waves, one mostly idle; ragged last tile
    (5000, 384, 128, 10),    # 4 waves full
    (5000, 384, 129, 6),     # 8 waves, 5th wave holds one query
    (20000, 384, 256, 10),   # 8 waves full: the N-GPU / config-3 batch
    (9000, 384, 300, 10),    # two 256-query blocks, second ragged
    (4000, 384, 512, 16),    # two full blocks, k at the selector's edge
    (2500, 128, 200, 1),
    (2500, 256, 100, 16),
    (3000, 512, 160, 10),
    (777, 100, 90, 5),       # dim padded 100 -> 128
    (40, 384, 256, 10),      # fewer tiles than workgroups
])
def test_wide_scan_matches_oracle(cuda, n, d, nq, k):
    q, c = _case(n, d, nq, seed=(n + nq) % 7)
    gs, gi = _run(cuda, q, c, k)
    check_topk(gs, gi, scan_ref.full_scores_f64(q, c), k)
    rs, ri = scan_ref.cosine_topk_ref(q, c, k)
    assert (gi == ri).mean() > 0.99


@pytest.mark.parametrize("n,k", [(1, 10), (5, 10), (16, 16), (31, 12)])
def test_wide_fewer_rows_than_k(cuda, n, k):
    q, c = _case(n, 384, 130)
    gs, gi = _run(cuda, q, c, k)
    check_topk(gs, gi, scan_ref.full_scores_f64(q, c), k)


def test_wide_all_rows_identical_lower_id_first(cuda):
    v = scan_ref.synth_corpus(1, 384, seed=7).astype(np.float16)
    c = np.repeat(v, 3000, axis=0)
    q = scan_ref.synth_corpus(140, 384, seed=8).astype(np.float16)
    for k in (10, 16):
        gs, gi = _run(cuda, q, c, k)
        assert np.array_equal(gi, np.tile(np.arange(k), (140, 1)))
        assert (gs == gs[:, :1]).all()


def test_wide_planted_duplicates(cuda):
    q, c = _case(4096, 384, 256)
    c = c.copy()
    full = scan_ref.full_scores_f64(q, c)
    for qi in (0, 100, 255):
        best = int(full[qi].argmax())
        for pos in (17 + qi, 4000 - qi, 4095 - qi):
            c[pos] = c[best]
    gs, gi = _run(cuda, q, c, 10)
    full = scan_ref.full_scores_f64(q, c)
    check_topk(gs, gi, full, 10)
    rs, ri = scan_ref.cosine_topk_ref(q, c, 10)
    for qi in (0, 100, 255):
        assert np.array_equal(gi[qi][:4], ri[qi][:4])


def test_wide_ascending_scores_worst_case(cuda):
    # scores of query 0 only ever increase along the stream: every row passes the running threshold
    q, c = _case(6000, 384, 200)
    full = scan_ref.full_scores_f64(q, c)
    c = c[np.argsort(full[0])]
    for k in (10, 16):
        gs, gi = _run(cuda, q, c, k)
        check_topk(gs, gi, scan_ref.full_scores_f64(q, c), k)


def test_wide_agrees_with_narrow_kernel(cuda, monkeypatch):
    """Same inputs through scan.hip (queries in slices of 64) and scan_wide.hip: identical ids and scores."""
    q, c = _case(30000, 384, 256, seed=3)
    gs, gi = _run(cuda, q, c, 10)
    for lo in range(0, 256, 64):
        ns, ni = _run(cuda, q[lo:lo + 64], c, 10)
        assert np.array_equal(ni, gi[lo:lo + 64])
        assert np.array_equal(ns, gs[lo:lo + 64])


# ---- group-best specifics (scan_refine.hip): several top-k rows inside ONE row group / one tile
@pytest.mark.parametrize("nq", [8, 200])
def test_topk_rows_share_a_group(cuda, nq):
    q, c = _case(5000, 384, nq, seed=5)
    c = c.copy()
    full = scan_ref.full_scores_f64(q, c)
    best = int(full[0].argmax())
    # rows 2048..2051 and 2064..2067 sit in one lane's group of tile 64 (32-row tiles); 2056.. in a sibling
    for pos in (2048, 2049, 2050, 2051, 2064, 2065, 2056, 2057):
        c[pos] = c[best]
    c[2052] = (c[best].astype(np.float32) * 0.999).astype(np.float16)
    gs, gi = _run(cuda, q, c, 10)
    full = scan_ref.full_scores_f64(q, c)
    check_topk(gs, gi, full, 10)
    rs, ri = scan_ref.cosine_topk_ref(q, c, 10)
    assert np.array_equal(gi[0], ri[0])


@pytest.mark.parametrize("d,nq", [(768, 16), (1024, 40), (384, 64), (256, 130)])
def test_every_row_of_one_tile_wins(cuda, d, nq):
    """The whole top-16 is one contiguous run of rows (a single tile, all of its groups)."""
    q, c = _case(3000, d, nq, seed=2)
    c = c.copy()
    v = q[0].astype(np.float32)
    for j in range(16):
        c[1600 + j] = (v * (1.0 - 0.001 * j)).astype(np.float16)
    for k in (16, 5):
        gs, gi = _run(cuda, q, c, k)
        check_topk(gs, gi, scan_ref.full_scores_f64(q, c), k)


# ---- rows wider than 512 elements: scan_tb.hip with 8 waves (128 queries / workgroup), or 4 waves x several
# query blocks where a 16-row tile does not split over 512 threads (640, 896); chain mode needs long streams
@pytest.mark.parametrize("n,d,nq,k", [
    (3000, 768, 130, 10),     # 8 waves, two query blocks (XCD-aware grid)
    (2000, 1024, 100, 16),    # 8 waves, k at the selector's edge
    (2500, 640, 70, 6),       # 4 waves x 2 query blocks
    (1800, 896, 200, 3),      # 4 waves x 4 query blocks
    (150_000, 768, 64, 10),   # dump -> chain switch-over region for 16-row tiles (12 tiles / stream)
    (400_000, 384, 64, 10),   # chain mode, 3 workgroups / CU, prefetch depth 2
    (400_000, 384, 64, 16),   # chain mode, 16 slots (single staging set)
    (300_000, 128, 32, 4),    # chain mode, 4 slots, one idle wave pair
])
def test_tile_best_variants(cuda, n, d, nq, k):
    q, c = _case(n, d, nq, seed=(n + d) % 5)
    gs, gi = _run(cuda, q, c, k)
    rs, ri = scan_ref.cosine_topk_ref(q, c, k, accumulate=np.float64)
    assert np.abs(gs - rs).max() < 2e-5
    mism = gi != ri
    assert (np.abs(gs - rs)[mism] < 4e-6).all()          # any id mismatch is a near-tie in score
    assert np.mean([scan_ref.recall_at_k(gi[r], ri[r]) for r in range(nq)]) > 0.995


def test_chain_mode_ties_across_lanes_and_streams(cuda):
    """Long stream (chain mode) where many rows tie exactly: duplicates of one row planted across tiles that
    different lanes / workgroups own -- the final list must be the lowest ids, in order."""
    q, c = _case(400_000, 384, 8, seed=1)
    c = c.copy()
    full0 = c.astype(np.float32) @ q[0].astype(np.float32)
    best = int(full0.argmax())
    planted = sorted({best, 7, 33, 64, 4097, 100_003, 100_035, 250_000, 250_001, 399_999, 123_456, 50})
    for pos in planted:
        c[pos] = c[best]
    gs, gi = _run(cuda, q, c, 10)
    assert np.array_equal(gi[0], np.array(planted[:10]))
    assert (gs[0] == gs[0][0]).all()
    rs, ri = scan_ref.cosine_topk_ref(q, c, 10, accumulate=np.float64)
    assert np.mean([scan_ref.recall_at_k(gi[r], ri[r]) for r in range(8)]) > 0.99


def test_w2_dump_is_bounded_and_the_fallback_is_exact(cuda):
    """The 256-query kernel for 768-element rows dumps every tile's representative: workspace and merge input grow
    with the shard.  Past 256 entries per (query, stream) the planner must hand over to the bounded chain kernels
    (10 M x 768 at 256 queries would be 640 MB of workspace, 312 k merge candidates per query), and both sides of
    that switch -- and the two-level merge that serves dumps of > 8192 candidates -- return the oracle's answer."""
    import os
    import torch
    from rag import _native as nat
    if os.environ.get("CRS_SCAN_TB") == "0":
        pytest.skip("this run forces the threshold kernels (tests/test_scan_classic_gpu.py)")
    assert "scan_w2" in nat.scan_plan_describe(256, 768, 10, 1_000_000)                 # C3: the dump form
    assert "scan_w2" in nat.scan_plan_describe(256, 768, 10, 2_000_000)                 # 62500 tiles / 256 streams = 245
    plan_big = nat.scan_plan_describe(256, 768, 10, 10_000_000)
    assert "scan_w2" not in plan_big, plan_big
    assert nat.scan_workspace_bytes(256, 768, 10, 10_000_000) < 64 << 20
    assert nat.scan_workspace_bytes(1024, 768, 10, 10_000_000) < 256 << 20
    n, d, nq, k = 2_200_000, 768, 256, 10                                                # 68750 tiles / 256 = 269 > 256
    assert "scan_w2" not in nat.scan_plan_describe(nq, d, k, n)
    g = torch.Generator(device=cuda); g.manual_seed(5)
    slab = torch.empty((n, d), dtype=torch.float16, device=cuda)
    for lo in range(0, n, 250_000):
        m = min(250_000, n - lo)
        nat.slab_append_f32(torch.randn((m, d), generator=g, device=cuda), slab, lo, nat.SLAB_F16)
    q16 = nat.queries_to_f16(torch.randn((nq, d), generator=g, device=cuda))
    for rows in (n, 1_000_000):                                                          # chain fallback; dump + two-level merge
        s, i = nat.cosine_topk(q16, slab, rows, d, k)
        torch.cuda.synchronize()
        rs, ri = scan_ref.cosine_topk_ref(q16.cpu().numpy(), slab[:rows].cpu().numpy(), k)
        assert (i.cpu().numpy() == ri).mean() > 0.999
        assert np.abs(s.cpu().numpy() - rs).max() < 2e-5
