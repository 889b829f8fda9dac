#!/usr/bin/env python3
"""Does the query encoder make progress WHILE a scan kernel occupies every CU?  Stream A runs a C4 scan (10M x 384,
~1.3 ms); stream B starts an encoder forward (64 x 16 tokens) right behind the scan's launch.  Reports when the
encoder finished relative to the scan.  Run with and without CRS_PANEL_KC=128 CRS_ENC_QKVATTN=0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
from rag import _native as nat
from rag._encoder import HipEncoder, ModelShape
from rag.embedding import _KNOWN, synthetic_weights
dev = torch.device("cuda:0")
rows, dim = int(os.environ.get("ROWS", 10_000_000)), 384
slab = torch.empty((rows, dim), dtype=torch.float16, device=dev)
for lo in range(0, rows, 250_000):
    nat.slab_append_f32(torch.randn((min(250_000, rows - lo), dim), device=dev), slab, lo, nat.SLAB_F16)
shape = ModelShape(ln_eps=1e-12, **_KNOWN["all-minilm-l6-v2"])
enc = HipEncoder(shape, synthetic_weights(shape, seed=7), device=dev)
ids = torch.randint(1000, 30000, (64, 16), dtype=torch.int32, device=dev)
lens = torch.full((64,), 16, dtype=torch.int32, device=dev)
q16 = nat.queries_to_f16(torch.randn((64, dim), device=dev))
ws = torch.empty(nat.scan_workspace_bytes(64, dim, 16, rows), dtype=torch.uint8, device=dev)
os_, oi_ = torch.empty((64, 16), device=dev), torch.empty((64, 16), dtype=torch.int64, device=dev)
out = torch.empty((64, dim), device=dev); ews = torch.empty(enc.workspace_bytes(64, 16), dtype=torch.uint8, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def scan(): nat.cosine_topk(q16, slab, rows, dim, 16, workspace=ws, out_scores=os_, out_ids=oi_)
def encode(): enc.forward(ids, lens, out=out, workspace=ews)
for _ in range(3): scan(); encode()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=sb): encode()
torch.cuda.synchronize()
res = []
for rep in range(5):
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    with torch.cuda.stream(sa):
        sa.wait_event(e0); scan(); ea.record()
    with torch.cuda.stream(sb):
        sb.wait_event(e0); g.replay(); eb.record()
    torch.cuda.synchronize()
    res.append((e0.elapsed_time(ea), e0.elapsed_time(eb)))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); g.replay(); t1.record(); torch.cuda.synchronize()
print("env", {k: v for k, v in os.environ.items() if k.startswith("CRS_")}, "| encoder alone %.3f ms" % t0.elapsed_time(t1))
for a, b in res: print("  scan done at %.3f ms, encoder (other stream) done at %.3f ms" % (a, b))
