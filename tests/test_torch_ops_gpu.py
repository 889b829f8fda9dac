"""GPU: the PyTorch-ROCm custom ops (torch.ops.crs.*, csrc/torch_ops.cpp) called DIRECTLY -- not through the
rag/ wrappers -- against the oracle.  These are the ops north_star names as the Python->HIP boundary."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cosine_topk_op_matches_oracle(cuda):
    import torch
    from oracle import scan_ref
    from rag import _native as nat
    ops = nat.ops()
    n, d, nq, k = 20_000, 384, 33, 10
    c = scan_ref.synth_corpus(n, d, seed=11)
    q = scan_ref.synth_queries(c, nq, seed=12)
    slab = torch.zeros((n, 384), dtype=torch.float16, device=cuda)
    ops.slab_append(torch.from_numpy(c).to(cuda), slab, None, None, 0)
    q16 = torch.empty((nq, 384), dtype=torch.float16, device=cuda)
    ops.queries_to_f16(torch.from_numpy(q).to(cuda), q16, 0)
    s, i = torch.ops.crs.cosine_topk(q16, slab, None, n, d, k, 0)
    torch.cuda.synchronize()
    rs, ri = scan_ref.cosine_topk_ref(q16.cpu().numpy(), slab.cpu().numpy(), k)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.abs(s.cpu().numpy() - rs).max() < 2e-5
    # int8 slab through the same op
    slab8 = torch.zeros((n, 512), dtype=torch.int8, device=cuda)
    sc8 = torch.zeros(n, dtype=torch.float32, device=cuda)
    ops.slab_append(torch.from_numpy(c).to(cuda), slab8, sc8, None, 0)
    q16b = torch.empty((nq, 512), dtype=torch.float16, device=cuda)
    ops.queries_to_f16(torch.from_numpy(q).to(cuda), q16b, 1)
    s8, i8 = torch.ops.crs.cosine_topk(q16b, slab8, sc8, n, d, k, 7)
    qd = scan_ref.dequantized_queries(q16b.cpu().numpy())
    full = scan_ref.full_scores_f64(qd, slab8.cpu().numpy(), sc8.cpu().numpy())
    got = i8.cpu().numpy() - 7
    for r in range(nq):
        assert np.abs(full[r, got[r]] - s8[r].cpu().numpy()).max() < 1e-5
        assert full[r, got[r]].min() >= np.sort(full[r])[-k] - 1e-6


def test_merge_and_refine_ops(cuda):
    import torch
    from oracle import scan_ref
    rng = np.random.default_rng(3)
    sc = np.sort(rng.standard_normal((4, 9, 6)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    ids = rng.integers(0, 1000, size=(4, 9, 6)).astype(np.int64)
    s, i = torch.ops.crs.merge_topk(torch.from_numpy(sc).to(cuda), torch.from_numpy(ids).to(cuda), 5)
    rs, ri = scan_ref.merge_topk_ref(sc, ids, 5)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    shadow = rng.standard_normal((300, 64)).astype(np.float32)
    q = rng.standard_normal((5, 64)).astype(np.float32)
    cand = np.stack([rng.permutation(300)[:12] for _ in range(5)]).astype(np.int64)
    s2, i2 = torch.ops.crs.refine_f32(torch.from_numpy(q).to(cuda), torch.from_numpy(shadow).to(cuda), 300, 0,
                                     torch.from_numpy(cand).to(cuda), 4)
    for r in range(5):
        sc_r = (shadow[cand[r]].astype(np.float64) @ q[r].astype(np.float64))
        order = np.lexsort((cand[r], -sc_r))[:4]
        assert list(i2[r].cpu().numpy()) == list(cand[r][order])
        assert np.abs(s2[r].cpu().numpy() - sc_r[order]).max() < 1e-5


def test_encoder_forward_op_matches_oracle(cuda):
    import torch
    from oracle import encoder_ref as er
    from rag import _native as nat
    from rag._encoder import HipEncoder, ModelShape
    cfg = er.TINY if hasattr(er, "TINY") else er.MINILM_L6
    w = er.make_weights(cfg, seed=21)
    enc = HipEncoder(ModelShape(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.ln_eps,
                                cfg.pooling, cfg.max_seq), w, device=cuda)
    ids, mask = er.synth_tokens(cfg, 6, 20, seed=22)
    ids_d = torch.from_numpy(ids.astype(np.int32)).to(cuda)
    lens_d = torch.from_numpy(mask.sum(1).astype(np.int32)).to(cuda)
    out = torch.empty((6, cfg.hidden), dtype=torch.float32, device=cuda)
    ws = torch.empty(enc.workspace_bytes(6, 20), dtype=torch.uint8, device=cuda)
    torch.ops.crs.encoder_forward(ids_d, lens_d, enc._wlist, enc._desc_list, float(cfg.ln_eps), ws, out, None, 0, True, None)
    ref = er.encode_ref(ids, mask, w, cfg)
    cos = (out.cpu().numpy() * ref).sum(1)
    assert cos.min() > 1 - 2e-4


def test_ops_raise_on_bad_arguments(cuda):
    import torch
    from rag import _native as nat
    nat.ops()
    slab = torch.zeros((10, 384), dtype=torch.float16, device=cuda)
    q16 = torch.zeros((2, 256), dtype=torch.float16, device=cuda)           # wrong row length
    with pytest.raises(RuntimeError):
        torch.ops.crs.cosine_topk(q16, slab, None, 10, 384, 3, 0)
    with pytest.raises(RuntimeError):
        torch.ops.crs.cosine_topk(torch.zeros((2, 384), dtype=torch.float16, device=cuda), slab, None, 10, 384, 65, 0)   # k > CRS_MAX_K
