#!/usr/bin/env python3
"""Randomised parity sweep of the HIP encoder against the fp32 oracle over (model, batch, seq, ragged lengths): crosses
every dispatch threshold of enc_capi.hip (fused QKV+attention at 16/32/64 tokens, panel vs tiled vs streaming GEMMs at
512 / 2048 / 4096 tokens, the index-build projection+LayerNorm kernel).   python tools/fuzz_encoder.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
import numpy as np, torch
from oracle import encoder_ref as er
from rag._encoder import HipEncoder, ModelShape

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
encs = {}
def get(cfg, name):
    if name not in encs:
        w = er.make_weights(cfg, seed=5)
        encs[name] = (HipEncoder(ModelShape(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.ln_eps,
                                            cfg.pooling, cfg.max_seq), w, device=dev), w)
    return encs[name]
targets = [100, 500, 520, 1000, 2000, 2100, 4000, 4200, 6000]
for ci in range(cases):
    name = str(rng.choice(["minilm", "minilm", "bge", "tiny"]))
    cfg = {"minilm": er.MINILM_L6, "bge": er.BGE_BASE, "tiny": er.TINY}[name]
    seq = int(rng.choice([1, 7, 16, 24, 32, 64, 100, 128, 200]))
    seq = min(seq, cfg.max_seq)
    tokens = int(rng.choice(targets))
    if name == "bge": tokens = min(tokens, 2200)      # keep the CPU oracle quick
    batch = max(1, tokens // seq)
    enc, w = get(cfg, name)
    ids, mask = er.synth_tokens(cfg, batch, seq, seed=int(rng.integers(1 << 30)))
    lens = rng.integers(1, seq + 1, size=batch).astype(np.int32)
    lens[0] = seq
    mask = (np.arange(seq)[None, :] < lens[:, None]).astype(np.int32)
    got = enc.forward(ids, lens).cpu().numpy()
    ref = er.encode_ref(ids, mask, w, cfg)
    cos = (got * ref).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(ref, axis=1))
    err = np.abs(got - ref).max()
    ok = cos.min() > 1 - 2e-4 and err < 3e-3
    print(f"{'ok' if ok else 'FAIL'} {ci:3d} {name:6s} batch={batch:4d} seq={seq:3d} tokens={batch*seq:5d}  min cos {cos.min():.7f}  max |d| {err:.2e}", flush=True)
    if not ok: sys.exit(1)
print("all cases passed")
