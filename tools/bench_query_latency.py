#!/usr/bin/env python3
"""Single-query latency of the product path pieces: encoder forward (graph replay vs eager) and search.
   python tools/bench_query_latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
import numpy as np, torch
from oracle import encoder_ref as er
from rag._encoder import HipEncoder, ModelShape
from rag import _native as nat
cfg = er.MINILM_L6
dev = torch.device("cuda:0")
enc = HipEncoder(ModelShape(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.ln_eps, cfg.pooling, cfg.max_seq),
                 er.make_weights(cfg, seed=1), device=dev)
ids, mask = er.synth_tokens(cfg, 1, 16, seed=2)
lens = mask.sum(1).astype(np.int32)
ws = torch.empty(enc.workspace_bytes(1, 16), dtype=torch.uint8, device=dev)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f"encoder forward, 1 x 16 tokens: {timeit(lambda: enc.forward(ids, lens, workspace=ws)):.0f} us (38 launches; before the QKV+attention fusion 44 launches took 266 us, and a hipGraph replay of those measured 254: the C++ launch loop is not the bound)")
n, d = 100_000, 384
slab = torch.zeros((n, nat.padded_dim(d)), dtype=torch.float16, device=dev)
x = torch.randn((n, d), device=dev); nat.slab_append_f32(x, slab, 0, nat.SLAB_F16)
q = enc.forward(ids, lens)
def search():
    q16 = nat.queries_to_f16(q); return nat.cosine_topk(q16, slab, n, d, 6)
print(f"search 1 query x {n} rows, k=6: {timeit(search):.0f} us")
