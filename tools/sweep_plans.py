#!/usr/bin/env python3
"""Sweep of the search's plan and time over (row length, slab type, query batch, k) at a fixed corpus size, to spot cliffs
between kernel families.   python tools/sweep_plans.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
import torch
from rag import _native as nat
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for d, st in [(384, nat.SLAB_F16), (768, nat.SLAB_F16), (128, nat.SLAB_F16), (1024, nat.SLAB_F16), (768, nat.SLAB_I8), (384, nat.SLAB_I8)]:
    pd = nat.padded_dim(d, st)
    slab = torch.empty((n, pd), dtype=torch.int8 if st == nat.SLAB_I8 else torch.float16, device=dev)
    scales = torch.ones(n, dtype=torch.float32, device=dev) if st == nat.SLAB_I8 else None
    for lo in range(0, n, 500_000):
        hi = min(n, lo + 500_000)
        nat.slab_append_f32(torch.nn.functional.normalize(torch.randn((hi - lo, d), device=dev), dim=1), slab, lo, st, scales=scales)
    elem = 1 if st == nat.SLAB_I8 else 2
    for nq in (1, 16, 64, 65, 128, 256, 512):
        q16 = nat.queries_to_f16(torch.nn.functional.normalize(torch.randn((nq, d), device=dev), dim=1), st)
        for k in (6, 16, 32, 40):
            total, scan = nat.time_cosine_topk(q16, slab, n, d, k, 5, slab_type=st, scales=scales)
            gf = 2.0 * nq * n * pd / scan / 1e9
            print(f"d={d:4d} {'i8 ' if st == nat.SLAB_I8 else 'f16'} nq={nq:3d} k={k:2d}: scan {scan:7.3f} ms {n * pd * elem / scan / 1e9:5.2f} TB/s {gf:6.0f} TFLOP/s  "
                  f"search {total:7.3f} ms  {nat.scan_plan_describe(nq, d, k, n, slab_type=st)[:62]}", flush=True)
    del slab
