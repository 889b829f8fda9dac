#!/usr/bin/env python3
"""Randomised parity sweep of crs_cosine_topk against the oracle over (rows, dim, queries, k, slab type): exercises
every kernel family make_plan can pick (dump / chain / wide / wide_ks / 8-wave / int8 / threshold), ragged tiles,
several query blocks, planted duplicates.   python tools/fuzz_scan.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import scan_ref
from rag import _native as nat
from topk_check import check_topk

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
seen = {}
for ci in range(cases):
    i8 = rng.random() < 0.3
    d = int(rng.choice([100, 128, 256, 384, 512, 768, 1024] if i8 else [100, 128, 256, 384, 512, 640, 768, 896, 1024]))
    n = int(rng.choice([1, 7, 33, 100, 1000, 3001, 20000, 70000, 300000], p=[.05, .05, .05, .1, .2, .2, .15, .1, .1]))
    if n * d > 120_000_000: n = 120_000_000 // d
    nq = int(rng.choice([1, 5, 64, 65, 100, 128, 129, 256, 300, 520]))
    k = int(rng.choice([1, 3, 6, 10, 16, 17, 40, 64]))
    c = scan_ref.synth_corpus(n, d, seed=int(rng.integers(1 << 30)))
    q = scan_ref.synth_queries(c, nq, seed=int(rng.integers(1 << 30))).astype(np.float16)
    if n > 40 and rng.random() < 0.5:   # exact duplicates of a strong row, spread over the corpus
        full = c.astype(np.float32) @ q[0].astype(np.float32)
        b = int(full.argmax())
        for pos in rng.integers(0, n, size=4): c[pos] = c[b]
    st = nat.SLAB_I8 if i8 else nat.SLAB_F16
    pd = nat.padded_dim(d, st)
    qd = torch.zeros((nq, pd), dtype=torch.float16); qd[:, :d] = torch.from_numpy(q)
    if i8:
        c8, sc = scan_ref.quantize_rows_i8(c)
        s = torch.zeros((n, pd), dtype=torch.int8); s[:, :d] = torch.from_numpy(c8)
        scales = torch.from_numpy(sc).to(dev)
        full64 = scan_ref.full_scores_f64(scan_ref.dequantized_queries(q), c8, sc)
    else:
        c16 = c.astype(np.float16)
        s = torch.zeros((n, pd), dtype=torch.float16); s[:, :d] = torch.from_numpy(c16)
        scales = None
        full64 = scan_ref.full_scores_f64(q, c16)
    plan = nat.scan_plan_describe(nq, d, k, n, st)
    print("   %3d n=%7d d=%4d nq=%3d k=%2d %s %s ..." % (ci, n, d, nq, k, "i8 " if i8 else "f16", plan), flush=True)
    gs, gi = nat.cosine_topk(qd.to(dev), s.to(dev), n, d, k, slab_type=st, scales=scales)
    torch.cuda.synchronize()
    try:
        check_topk(gs.cpu().numpy(), gi.cpu().numpy(), full64, k)
    except AssertionError as e:
        print(f"FAIL case {ci}: n={n} d={d} nq={nq} k={k} i8={i8} plan={plan}: {e}")
        sys.exit(1)
    fam = plan.split("<")[0] + ("/dump" if ",0>" in plan.split(" ")[0] else "")
    seen[fam] = seen.get(fam, 0) + 1
    print(f"ok {ci:3d} n={n:7d} d={d:4d} nq={nq:3d} k={k:2d} {'i8 ' if i8 else 'f16'} {plan}", flush=True)
print("all cases passed; kernel families:", seen)
