#!/usr/bin/env python3
"""Dynamic tile tickets of scan_tb.hip (CRS_TB_DYN=<percent>): same lists as the static schedule, the counter's final value,
and the launch time per setting.   python3 tools/tb_dyn_check.py [rows] [dim]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "compressed-rag-suite_amd"))
import torch
nat = importlib.import_module("rag._native")
nat.require_gpu()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
k = 24
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
slab = torch.randn((rows, dim), device=dev, generator=g, dtype=torch.float32)
slab = (slab / slab.norm(dim=1, keepdim=True)).half()
q = torch.randn((64, dim), device=dev, generator=g, dtype=torch.float32)
q16 = (q / q.norm(dim=1, keepdim=True)).half()
ws = torch.empty(nat.scan_workspace_bytes(64, dim, k, rows), dtype=torch.uint8, device=dev)
print(nat.scan_plan_describe(64, dim, k, rows), flush=True)
ref = None
for pct, gran in ((0, 1), (10, 1), (10, 4), (20, 4), (20, 8), (50, 8), (0, 1)):
    os.environ["CRS_TB_DYN"] = str(pct)
    os.environ["CRS_TB_DYN_G"] = str(gran)
    ws.fill_(0xAB)   # poison: the library must zero its counter itself
    s, i = nat.cosine_topk(q16, slab, rows, dim, k, workspace=ws)
    torch.cuda.synchronize()
    ticket = int(ws[:4].view(torch.int32)[0])
    if ref is None:
        ref = (s.clone(), i.clone())
    same = bool((i == ref[1]).all()) and bool((s == ref[0]).all())
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        nat.cosine_topk(q16, slab, rows, dim, k, workspace=ws, out_scores=s, out_ids=i)
    t1.record(); torch.cuda.synchronize()
    print(f"dyn {pct:3d} % x{gran}: ticket word after the call {ticket:#x}  identical to static: {same}  {t0.elapsed_time(t1) / 20 * 1e3:.1f} us per call", flush=True)
    if not same:
        bad = (i != ref[1]).nonzero()
        print("   first differences:", bad[:5].tolist(), i[bad[0, 0]].tolist()[:8], ref[1][bad[0, 0]].tolist()[:8])
# the same call replayed from a hipGraph (the engine's form): is the counter's zeroing part of the graph?
os.environ["CRS_TB_DYN"] = "20"
st = torch.cuda.Stream()
s2 = torch.empty_like(ref[0]); i2 = torch.empty_like(ref[1])
with torch.cuda.stream(st):
    for _ in range(2):
        nat.cosine_topk(q16, slab, rows, dim, k, workspace=ws, out_scores=s2, out_ids=i2)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=st, capture_error_mode="thread_local"):
    nat.cosine_topk(q16, slab, rows, dim, k, workspace=ws, out_scores=s2, out_ids=i2)
for rep in range(3):
    ws[:16].fill_(0xAB)
    s2.zero_(); i2.zero_()
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    print(f"graph replay {rep}: ticket word {int(ws[:4].view(torch.int32)[0]):#x}  identical to static: {bool((i2 == ref[1]).all()) and bool((s2 == ref[0]).all())}", flush=True)
