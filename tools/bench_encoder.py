#!/usr/bin/env python3
"""Encoder throughput (index-build side of the path): tokens/s and achieved TFLOP/s vs the MFMA peak.
   python tools/bench_encoder.py [minilm|bge] [batch] [seq]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
import numpy as np, torch
from oracle import encoder_ref as er
from rag._encoder import HipEncoder, ModelShape

name = sys.argv[1] if len(sys.argv) > 1 else "minilm"
cfg = er.MINILM_L6 if name == "minilm" else er.BGE_BASE
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else cfg.max_seq
dev = torch.device("cuda:0")
enc = HipEncoder(ModelShape(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.ln_eps, cfg.pooling, cfg.max_seq),
                 er.make_weights(cfg, seed=1), device=dev)
ids, mask = er.synth_tokens(cfg, B, S, seed=2, ragged=False)
ids_d = torch.from_numpy(ids).to(dev); lens_d = torch.from_numpy(mask.sum(1).astype(np.int32)).to(dev)
for _ in range(3): enc.forward(ids_d, lens_d)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): enc.forward(ids_d, lens_d)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
T = B * S
H, F, L = cfg.hidden, cfg.ffn, cfg.layers
flops = T * L * (2 * (4 * H * H + 2 * H * F)) + L * 4 * S * H * T
print(f"{name} B={B} S={S}: {dt*1e3:.3f} ms/batch  {T/dt/1e6:.2f} Mtok/s  {B/dt:.0f} chunks/s  {flops/dt/1e12:.1f} TFLOP/s ({flops/dt/2.5e15*100:.1f}% of 2.5 PF)")
