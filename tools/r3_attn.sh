#!/bin/bash
cd "$(dirname "$0")/.."
python3 - <<'P'
import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "compressed-rag-suite_amd"); sys.path.insert(0, ".")
import torch
from oracle import encoder_ref as er
from rag._encoder import HipEncoder, ModelShape
cfg = er.MINILM_L6
dev = torch.device("cuda:0")
enc = HipEncoder(ModelShape(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.ln_eps, cfg.pooling, cfg.max_seq), er.make_weights(cfg, seed=31), device=dev)
ids, mask = er.synth_tokens(cfg, 4, 20, seed=5)
lens = mask.sum(1).astype(np.int32)
a = enc.forward(torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)).cpu().numpy()
wide = np.zeros((4, 70), dtype=np.int32); wide[:, :20] = ids
b = enc.forward(torch.from_numpy(wide).to(dev), torch.from_numpy(lens).to(dev)).cpu().numpy()
ref = er.encode(er.make_weights(cfg, seed=31), cfg, ids, mask) if hasattr(er, "encode") else None
print("padding test: max |a - b| =", np.abs(a - b).max(), " (x32 =", os.environ.get("CRS_ATTN_X32", "1"), ")")
P
CRS_ATTN_X32=0 python3 - <<'P'
import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "compressed-rag-suite_amd"); sys.path.insert(0, ".")
import torch
from oracle import encoder_ref as er
from rag._encoder import HipEncoder, ModelShape
cfg = er.MINILM_L6
dev = torch.device("cuda:0")
enc = HipEncoder(ModelShape(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.ln_eps, cfg.pooling, cfg.max_seq), er.make_weights(cfg, seed=31), device=dev)
ids, mask = er.synth_tokens(cfg, 4, 20, seed=5)
lens = mask.sum(1).astype(np.int32)
a = enc.forward(torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)).cpu().numpy()
wide = np.zeros((4, 70), dtype=np.int32); wide[:, :20] = ids
b = enc.forward(torch.from_numpy(wide).to(dev), torch.from_numpy(lens).to(dev)).cpu().numpy()
print("padding test: max |a - b| =", np.abs(a - b).max(), " (x32 = 0)")
P
for q in 2 4; do for w in enc-bge enc-minilm; do
CRS_ATTN_QT=$q python3 bench.py --workload $w --steps 20 --warmup 3 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  qt=$q $w', d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'])"
done; done
