"""The chain-mode tile-best scan hands most of its tiles out through a device-wide counter (csrc/scan_tb.hip, capi.hip run_scan:
CRS_TB_DYN / CRS_TB_DYN_G / CRS_TB_DYN_MIN).  Which workgroup multiplies which tile then differs from run to run; the lists must not:
  * bit-identical scores and ids against the static stride (CRS_TB_DYN=0) on the same inputs, workspace poisoned beforehand
    (the library zeroes its counter itself, in stream order);
  * the oracle's acceptance check (oracle/scan_ref.py) on the dynamic result;
  * the same when the call is replayed from a hipGraph, and for ragged row counts, short query blocks, ties and duplicates.
CRS_TB_DYN_MIN=4 switches the dynamic schedule on for the short streams a test can afford (default: >= 96 tiles per stream)."""
import os

import numpy as np
import pytest

from oracle import scan_ref
from topk_check import check_topk
from test_scan_gpu import _case

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _dev(cuda, q16_np, slab_np):
    import torch
    from rag import _native as nat
    nq, d = q16_np.shape
    n = slab_np.shape[0]
    pd = nat.padded_dim(d)
    q = torch.zeros((nq, pd), dtype=torch.float16)
    q[:, :d] = torch.from_numpy(q16_np)
    s = torch.zeros((n, pd), dtype=torch.float16)
    s[:, :d] = torch.from_numpy(slab_np)
    return q.to(cuda), s.to(cuda)


def _search(q, s, n, d, k, poison=0xAB):
    import torch
    from rag import _native as nat
    ws = torch.full((nat.scan_workspace_bytes(q.shape[0], d, k, n),), poison, dtype=torch.uint8, device=q.device)
    sc, ids = nat.cosine_topk(q, s, n, d, k, workspace=ws)
    torch.cuda.synchronize()
    return sc.cpu().numpy(), ids.cpu().numpy(), int(ws[:4].view(torch.int32)[0])


@pytest.mark.parametrize("n,d,nq,k,pct,gran", [
    (400_000, 384, 64, 10, 85, 8),     # the default schedule
    (400_001, 384, 64, 24, 85, 8),     # ragged last tile, the C4 over-fetch length
    (300_000, 384, 7, 10, 100, 1),     # every tile but the first two rounds drawn, one tile per ticket; short query block
    (300_000, 384, 64, 16, 50, 4),
    (250_000, 768, 64, 10, 85, 16),    # bge width
    (600_000, 128, 64, 10, 85, 8),     # three workgroups per CU
    (350_000, 256, 33, 32, 20, 2),
])
def test_dynamic_schedule_gives_the_static_lists(cuda, n, d, nq, k, pct, gran):
    from rag import _native as nat
    q_np, c_np = _case(n, d, nq, seed=(n + k) % 5)
    q, s = _dev(cuda, q_np, c_np)
    plan = nat.scan_plan_describe(nq, d, k, n)
    with _Env(CRS_TB_DYN=0):
        ss, si, _ = _search(q, s, n, d, k)
    with _Env(CRS_TB_DYN=pct, CRS_TB_DYN_G=gran, CRS_TB_DYN_MIN=4):
        ds, di, ticket = _search(q, s, n, d, k)
        ds2, di2, _ = _search(q, s, n, d, k, poison=0xFF)
    assert np.array_equal(si, di) and np.array_equal(ss, ds), plan
    assert np.array_equal(di, di2) and np.array_equal(ds, ds2)
    if "scan_tb_kernel" in plan and ",0>" not in plan and "qblocks=1" in plan:
        assert 0 < ticket < n, (plan, ticket)          # the counter was zeroed and used (a static launch leaves the poison)
    check_topk(ds, di, scan_ref.full_scores_f64(q_np, c_np), k)


def test_dynamic_schedule_ties_and_duplicates(cuda):
    n, d, nq, k = 300_000, 384, 64, 10
    q_np, c_np = _case(n, d, nq, seed=3)
    c_np = c_np.copy()
    full = scan_ref.full_scores_f64(q_np, c_np)
    for qi in (0, 31, 63):
        best = int(full[qi].argmax())
        for pos in (5 + qi, 150_000 + qi, n - 1 - qi):     # copies of the best row in the static part, the dynamic part, the last tile
            c_np[pos] = c_np[best]
    q, s = _dev(cuda, q_np, c_np)
    with _Env(CRS_TB_DYN=0):
        ss, si, _ = _search(q, s, n, d, k)
    with _Env(CRS_TB_DYN=85, CRS_TB_DYN_G=8, CRS_TB_DYN_MIN=4):
        for _ in range(3):
            ds, di, _ = _search(q, s, n, d, k)
            assert np.array_equal(si, di) and np.array_equal(ss, ds)
    check_topk(ds, di, scan_ref.full_scores_f64(q_np, c_np), k)


def test_dynamic_schedule_inside_a_graph(cuda):
    import torch
    from rag import _native as nat
    n, d, nq, k = 400_000, 384, 64, 24
    q_np, c_np = _case(n, d, nq, seed=1)
    q, s = _dev(cuda, q_np, c_np)
    with _Env(CRS_TB_DYN=0):
        ss, si, _ = _search(q, s, n, d, k)
    with _Env(CRS_TB_DYN=85, CRS_TB_DYN_G=8, CRS_TB_DYN_MIN=4):
        ws = torch.empty(nat.scan_workspace_bytes(nq, d, k, n), dtype=torch.uint8, device=q.device)
        out_s = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            nat.cosine_topk(q, s, n, d, k, workspace=ws, out_scores=out_s, out_ids=out_i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            nat.cosine_topk(q, s, n, d, k, workspace=ws, out_scores=out_s, out_ids=out_i)
        for _ in range(3):
            ws.fill_(0xCD)
            out_s.zero_()
            out_i.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(out_i.cpu().numpy(), si) and np.array_equal(out_s.cpu().numpy(), ss)


@pytest.mark.parametrize("n,d,nq,k,pct,gran", [
    (300_000, 768, 64, 16, 85, 8),     # C5's kernel (16-slot chain) with the default schedule
    (300_001, 768, 17, 10, 100, 1),
    (250_000, 384, 64, 32, 50, 4),     # padded to 512-element rows, 32-slot chain
])
def test_dynamic_schedule_int8(cuda, n, d, nq, k, pct, gran):
    import torch
    from rag import _native as nat
    c = scan_ref.synth_corpus(n, d, seed=77 + k)
    q_np = scan_ref.synth_queries(c, nq, seed=78).astype(np.float16)
    c8, sc = scan_ref.quantize_rows_i8(c)
    pd = nat.padded_dim(d, nat.SLAB_I8)
    q = torch.zeros((nq, pd), dtype=torch.float16)
    q[:, :d] = torch.from_numpy(q_np)
    s = torch.zeros((n, pd), dtype=torch.int8)
    s[:, :d] = torch.from_numpy(c8)
    q, s, scales = q.to(cuda), s.to(cuda), torch.from_numpy(sc).to(cuda)

    def search(poison):
        ws = torch.full((nat.scan_workspace_bytes(nq, d, k, n),), poison, dtype=torch.uint8, device=q.device)
        a, b = nat.cosine_topk(q, s, n, d, k, slab_type=nat.SLAB_I8, scales=scales, workspace=ws)
        torch.cuda.synchronize()
        return a.cpu().numpy(), b.cpu().numpy()

    with _Env(CRS_TB_DYN=0):
        ss, si = search(0xAB)
    with _Env(CRS_TB_DYN=pct, CRS_TB_DYN_G=gran, CRS_TB_DYN_MIN=4):
        ds, di = search(0xAB)
        ds2, di2 = search(0xFF)
    assert np.array_equal(si, di) and np.array_equal(ss, ds)
    assert np.array_equal(di, di2) and np.array_equal(ds, ds2)
    check_topk(ds, di, scan_ref.full_scores_f64(scan_ref.dequantized_queries(q_np), c8, sc), k)
