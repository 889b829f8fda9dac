"""GPU: the throughput engine's batching logic (rag/_engine.py) -- encode groups, role lanes, the token-batch iterator.

One encoder forward may serve G consecutive batches (their token blocks are slices of one block).  Whatever the grouping, the
lanes or the number of batches a call brings, ``search_token_batches`` must yield, IN INPUT ORDER, what an ungrouped engine with
one stream per batch yields for the same tokens: same rows (embeddings differ in a last bit when the forward's GEMMs see another
token count or run their small-LDS forms, so near-ties among random rows may swap: >= 97 % identical ids -- a mix-up of
batches or slices would leave ~0 % --, scores within 1e-4), a short last batch cut to its real queries, an
incomplete last group searched only for its real batches.  The reference has no batched entry point
(/root/reference/evaluation/retrieval/benchmark.py:241-247 loops over single queries); the per-query result is the bar."""
import numpy as np
import pytest

from oracle import encoder_ref as er

pytestmark = pytest.mark.gpu

ROWS, DIM, QB, SEQ, K = 40_000, 384, 16, 16, 10


@pytest.fixture(scope="module")
def world(cuda):
    import torch
    from rag import _native as nat
    from rag._encoder import HipEncoder, ModelShape
    from rag._engine import ShardView
    cfg = er.MINILM_L6
    enc = HipEncoder(ModelShape(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.ln_eps, cfg.pooling,
                                cfg.max_seq), er.make_weights(cfg, seed=3), device=cuda)
    g = torch.Generator(device=cuda).manual_seed(11)
    emb = torch.randn((ROWS, DIM), device=cuda, generator=g)
    emb = emb / emb.norm(dim=1, keepdim=True)
    pd = nat.padded_dim(DIM)
    slab = torch.zeros((ROWS, pd), dtype=torch.float16, device=cuda)
    shadow = torch.empty((ROWS, DIM), dtype=torch.float32, device=cuda)
    err = torch.zeros(1, dtype=torch.float32, device=cuda)
    nat.slab_append_f32(emb, slab, 0, nat.SLAB_F16, shadow=shadow, row_err=err)
    view = ShardView(slab, None, shadow, ROWS, DIM, nat.SLAB_F16, 0, float(err.item()))
    return enc, view


def _batches(n_batches, last, seed):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(n_batches):
        m = last if b == n_batches - 1 else QB
        ids = rng.integers(1000, 30000, size=(m, SEQ)).astype(np.int32)
        lens = rng.integers(4, SEQ + 1, size=m).astype(np.int32)
        ids[:, 0] = 101
        out.append((ids, lens))
    return out


def _run(engine, batches):
    res = list(engine.search_token_batches(iter(batches)))
    assert len(res) == len(batches)
    for (s, r, st), (ids, _) in zip(res, batches):
        assert s.shape == (ids.shape[0], K) and r.shape == (ids.shape[0], K) and st.shape == (ids.shape[0],)
    return res


def test_grouped_engine_matches_ungrouped_in_input_order(world):
    from rag._engine import RetrievalEngine
    enc, view = world
    plain = RetrievalEngine(enc, view, QB, SEQ, K, lanes="batch", encode_group=1, n_ctx=2)
    grouped = RetrievalEngine(enc, view, QB, SEQ, K, lanes="split", encode_group=4, n_ctx=8, enc_lanes=1, search_lanes=2)
    assert grouped.enc_group == 4 and plain.enc_group == 1 and grouped.describe_lanes().startswith("1 encoder + 2 search")
    for n_batches, last in ((1, QB), (1, 5), (3, QB), (4, 7), (5, QB), (9, 3), (19, QB), (8, QB), (21, 11)):
        batches = _batches(n_batches, last, seed=100 + n_batches)
        a, b = _run(plain, batches), _run(grouped, batches)
        same, total, worst = 0, 0, 0.0
        for (sa, ra, _), (sb, rb, _) in zip(a, b):
            same += int((ra == rb).sum())
            total += ra.size
            worst = max(worst, float(np.abs(sa - sb).max()))
        assert same / total >= 0.97 and worst < 1e-4, (n_batches, last, same / total, worst)
    # the same engine again with another call size: nothing left over from the previous call's incomplete group
    batches = _batches(6, 9, seed=7)
    a, b = _run(plain, batches), _run(grouped, batches)
    assert sum(int((x[1] == y[1]).sum()) for x, y in zip(a, b)) >= 0.97 * sum(x[1].size for x in a)


def test_group_members_need_their_groups_first_buffer_set(world):
    from rag import _native as nat
    from rag._engine import RetrievalEngine
    enc, view = world
    eng = RetrievalEngine(enc, view, QB, SEQ, K, lanes="split", encode_group=2, n_ctx=4, graphs=False)
    with pytest.raises(nat.NativeError):
        eng.submit(1)               # buffer set 1 belongs to the group of buffer set 0, whose forward has not been issued
    eng.submit(0)
    eng.submit(1)
    eng.wait(1)


def test_cu_masked_stream_runs_the_same_search(world):
    """crs_stream_create_cu_masked (include/crs_hip.h): a stream confined to 32 CUs runs the library's kernels with the same
    results (the encoder-placement A/B of DESIGN section 4 used it for the encoder lanes); a range past the device is refused."""
    import torch
    from rag import _native as nat
    enc, view = world
    st = nat.cu_masked_stream(0, 32, view.slab.device)
    g = torch.Generator(device=view.slab.device).manual_seed(5)
    q = torch.randn((8, DIM), device=view.slab.device, generator=g)
    q16 = nat.queries_to_f16(q / q.norm(dim=1, keepdim=True))
    s0, i0 = nat.cosine_topk(q16, view.slab, view.n, view.dim, K)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        s1, i1 = nat.cosine_topk(q16, view.slab, view.n, view.dim, K)
    st.synchronize()
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    with pytest.raises(nat.NativeError):
        nat.cu_masked_stream(250, 64, view.slab.device)


def test_no_batches_and_an_int8_shard(world, cuda):
    """An empty iterator yields nothing; an int8 shard (scales, empirical exactness) behaves the same under groups."""
    import torch
    from rag import _native as nat
    from rag._engine import RetrievalEngine, ShardView
    enc, view = world
    eng = RetrievalEngine(enc, view, QB, SEQ, K, lanes="split", encode_group=4, n_ctx=8)
    assert list(eng.search_token_batches(iter([]))) == []
    n = 30_000
    pd = nat.padded_dim(DIM, nat.SLAB_I8)
    slab = torch.zeros((n, pd), dtype=torch.int8, device=cuda)
    scales = torch.empty(n, dtype=torch.float32, device=cuda)
    shadow = torch.empty((n, DIM), dtype=torch.float32, device=cuda)
    err = torch.zeros(1, dtype=torch.float32, device=cuda)
    nat.slab_append_f32(view.shadow[:n].contiguous(), slab, 0, nat.SLAB_I8, scales=scales, shadow=shadow, row_err=err)
    v8 = ShardView(slab, scales, shadow, n, DIM, nat.SLAB_I8, 0, float(err.item()))
    plain = RetrievalEngine(enc, v8, QB, SEQ, K, lanes="batch", encode_group=1, n_ctx=2)
    grouped = RetrievalEngine(enc, v8, QB, SEQ, K, lanes="split", encode_group=4, n_ctx=8, enc_lanes=2, search_lanes=1)
    batches = _batches(7, 6, seed=9)
    a, b = _run(plain, batches), _run(grouped, batches)
    same = sum(int((x[1] == y[1]).sum()) for x, y in zip(a, b))
    assert same >= 0.97 * sum(x[1].size for x in a)
    assert max(float(np.abs(x[0] - y[0]).max()) for x, y in zip(a, b)) < 1e-4
