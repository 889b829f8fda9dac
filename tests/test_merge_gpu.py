"""GPU parity: crs_merge_topk (cross-shard merge of all-gathered partial lists) vs oracle."""
import numpy as np
import pytest

from oracle import scan_ref

pytestmark = pytest.mark.gpu


def _merge(cuda, s, i, k):
    import torch
    from rag import _native as nat
    gs, gi = nat.merge_topk(torch.from_numpy(s).to(cuda), torch.from_numpy(i).to(cuda), k)
    torch.cuda.synchronize()
    return gs.cpu().numpy(), gi.cpu().numpy()


@pytest.mark.parametrize("g,nq,kin,kout", [(8, 64, 10, 10), (2, 5, 6, 6), (8, 512, 10, 10), (4, 3, 40, 40),
                                           (3, 7, 10, 4), (1, 9, 64, 64), (8, 2, 3, 10), (600, 4, 10, 10)])
def test_merge_matches_oracle(cuda, g, nq, kin, kout):
    rng = np.random.default_rng(g * 1000 + nq)
    s = rng.standard_normal((g, nq, kin)).astype(np.float32)
    i = rng.permutation(g * nq * kin).reshape(g, nq, kin).astype(np.int64)
    # some empty slots and some exact score ties across shards
    empty = rng.random((g, nq, kin)) < 0.15
    i[empty] = -1
    s[empty] = -np.inf
    if g > 1:
        s[1, :, 0] = s[0, :, 0]
    gs, gi = _merge(cuda, s, i, kout)
    rs, ri = scan_ref.merge_topk_ref(s, i, kout)
    assert np.array_equal(gi, ri)
    assert np.array_equal(gs, rs)


def test_merge_all_tied_takes_fallback_path(cuda):
    # 8 shards x 512 queries... every score identical: more than 1024 tied candidates per query
    g, nq, kin = 64, 3, 40
    s = np.full((g, nq, kin), 0.5, dtype=np.float32)
    i = np.arange(g * nq * kin, dtype=np.int64)[::-1].copy().reshape(g, nq, kin)
    gs, gi = _merge(cuda, s, i, 40)
    rs, ri = scan_ref.merge_topk_ref(s, i, 40)
    assert np.array_equal(gi, ri) and np.array_equal(gs, rs)


def test_merge_all_empty(cuda):
    s = np.full((4, 2, 5), -np.inf, dtype=np.float32)
    i = np.full((4, 2, 5), -1, dtype=np.int64)
    gs, gi = _merge(cuda, s, i, 5)
    assert (gi == -1).all() and np.isneginf(gs).all()
