#!/usr/bin/env python3
"""crs_gemm_f16 throughput on the encoder's shapes (and a 4096^3 reference point)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
import torch
from rag._encoder import gemm_f16
dev = torch.device("cuda:0")
shapes = [(4096, 4096, 4096, 0), (65536, 1152, 384, 0), (65536, 1536, 384, 1), (65536, 384, 1536, 2), (65536, 384, 384, 2),
          (32768, 2304, 768, 0), (32768, 3072, 768, 1), (32768, 768, 3072, 2), (4096, 2304, 768, 0), (4096, 3072, 768, 1)]
for m, n, k, mode in shapes:
    a = (torch.randn((m, k), device=dev) * 0.5).half(); w = (torch.randn((n, k), device=dev) * 0.05).half()
    b = torch.randn(n, device=dev); r = torch.randn((m, n), device=dev) if mode == 2 else None
    for _ in range(3): gemm_f16(a, w, b, r, mode)
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 10
    for _ in range(it): gemm_f16(a, w, b, r, mode)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    print(f"M={m} N={n} K={k} mode={mode}: {dt*1e6:8.1f} us  {2*m*n*k/dt/1e12:7.1f} TFLOP/s")
