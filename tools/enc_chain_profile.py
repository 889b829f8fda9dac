#!/usr/bin/env python3
"""Query-regime encoder chain, ONE stream: graph-replayed forward latency for (arch, batch x 16 tokens).  Under
rocprofv3 --kernel-trace --stats the per-kernel table shows the isolated kernel durations of the chain.
   python tools/enc_chain_profile.py minilm 64 ; python tools/enc_chain_profile.py bge 64 ; ... bge 256"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from rag._encoder import HipEncoder, ModelShape
from rag.embedding import _KNOWN, synthetic_weights
arch = sys.argv[1] if len(sys.argv) > 1 else "minilm"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
seq = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
shape = ModelShape(ln_eps=1e-12, **_KNOWN["all-minilm-l6-v2" if arch == "minilm" else "bge-base-en-v1.5"])
enc = HipEncoder(shape, synthetic_weights(shape, seed=7), device=dev)
ids = torch.randint(1000, 30000, (batch, seq), dtype=torch.int32, device=dev)
lens = torch.full((batch,), seq, dtype=torch.int32, device=dev)
out = torch.empty((batch, shape.hidden), device=dev); ws = torch.empty(enc.workspace_bytes(batch, seq), dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3): enc.forward(ids, lens, out=out, workspace=ws)
st.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st): enc.forward(ids, lens, out=out, workspace=ws)
torch.cuda.synchronize()
n = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(st):
    e0.record()
    for _ in range(n): g.replay()
    e1.record()
torch.cuda.synchronize()
print(f"{arch} {batch} x {seq} tokens: {e0.elapsed_time(e1) / n * 1e3:.1f} us per forward (graph replay, back to back on one stream)")
