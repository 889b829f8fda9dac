#!/usr/bin/env python3
"""crs_gemm_f16 throughput on the encoder's shapes (+ a 4096^3 reference point), with a correctness check against torch.
   CRS_GEMM_BIG=0|1|2 selects the dispatch (0: 128 x 128 kernel / row-streaming only, 1: + 256-row tiles where the tiled kernel
   was used, 2: 256-row tiles wherever they apply)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
import torch
from rag._encoder import gemm_f16
dev = torch.device("cuda:0")
shapes = [(4096, 4096, 4096, 0),
          (65536, 1152, 384, 0), (65536, 1536, 384, 1), (65536, 384, 1536, 2), (65536, 384, 384, 2),          # MiniLM, 256 x 256 tokens
          (32768, 2304, 768, 0), (32768, 3072, 768, 1), (32768, 768, 3072, 2), (32768, 768, 768, 2),           # bge-base, 64 x 512 tokens
          (4096, 2304, 768, 0), (4096, 3072, 768, 1), (4096, 768, 3072, 2)]
if len(sys.argv) > 4: shapes = [tuple(int(x) for x in sys.argv[i:i + 4]) for i in range(1, len(sys.argv) - 3, 4)]
elif len(sys.argv) > 1: shapes = [x for x in shapes if x[0] == int(sys.argv[1]) and x[1] > 384]
print("CRS_GEMM_BIG =", os.environ.get("CRS_GEMM_BIG", "(default 1)"))
for m, n, k, mode in shapes:
    g = torch.Generator(device=dev); g.manual_seed(m + n + k)
    a = (torch.randn((m, k), device=dev, generator=g) * 0.5).half(); w = (torch.randn((n, k), device=dev, generator=g) * 0.05).half()
    b = torch.randn(n, device=dev, generator=g); r = torch.randn((m, n), device=dev, generator=g) if mode == 2 else None
    out = gemm_f16(a, w, b, r, mode)
    rows = torch.randint(0, m, (64,), device=dev)
    ref = a[rows].float() @ w.float().T + b
    if mode == 1: ref = torch.nn.functional.gelu(ref)
    if mode == 2: ref = ref + r[rows]
    err = (out[rows].float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    for _ in range(3): gemm_f16(a, w, b, r, mode)
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 10
    for _ in range(it): gemm_f16(a, w, b, r, mode)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    print(f"M={m:6d} N={n:5d} K={k:5d} mode={mode}: {dt*1e6:8.1f} us  {2*m*n*k/dt/1e12:7.1f} TFLOP/s   rel err {err:.1e}" + ("  <-- WRONG" if err > (2e-3 if mode != 2 else 2e-4) else ""))
