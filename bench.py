#!/usr/bin/env python3
"""bench.py -- queries/sec of the embed -> index -> retrieve hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N > 1: launched by
torch.distributed.run, one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

One query BATCH through the retrieve path, everything resident in HBM:
  query encoder forward (token ids [Qb, 16] -> fp32 sentence embeddings + the scan's fp16 query block)
  -> exact cosine scan of this rank's slab shard, over-fetching k' = 16 candidates per query
  -> per-workgroup list merge + tile refine -> fp32 re-rank of the k' candidates against the fp32 shadow
     of the shard (the ranking the reference's fp32 ChromaDB collection gives), best k = 10 written
     straight into this rank's wire block
  -> (N > 1) ONE RCCL all-gather of the wire blocks + k-way merge on every rank.
A STEP is S such batches in flight (S = --streams, each with its own buffers), so the timed region is steady state
whatever --steps is; queries per step = S x Qb.  --lanes: every batch wholly on its own HIP stream ("batch"), or encoder
forwards on encoder lane(s) and searches on search lane(s), tied by events ("split": C4-class scans; see main()).

Workloads (BASELINE.json configs; --workload):
  c4  10M x 384 fp16 corpus, global 64-query batches, k=10      (default: the configuration the metric is quoted on)
  c5  10M x 768 int8 (+fp32 row scales) corpus, fp16 queries, 64-query batches
  c3  1M x 768 fp16, 256-query batches
  c2  100k x 384 fp16, 64-query batches (Infinity-Cache resident: a latency case, not an HBM measurement)
  enc-minilm / enc-bge   the index-build side: encoder forward over full-length chunks (tokens/s, MFMA fraction)
Scaling is STRONG by default: the corpus is fixed and split into N contiguous
row shards (10M / N rows per GPU), the global query batch is fixed and every rank scans its shard for all of it.
The queries are encoded replicated on every rank (N <= 2: one collective per batch) or in shards of Qb / N per rank
followed by an all-gather of the embeddings (N >= 4: two collectives; --encode).  --scaling weak keeps the round-1
behaviour (fixed 1.25M-row shard and Qb queries PER RANK; queries are all-gathered first).

The encoder has the architecture BASELINE.json names for the workload (all-MiniLM-L6-v2 for the 384-d
configs, bge-base-en-v1.5 for the 768-d ones) with seeded random weights and synthetic token ids (no
checkpoints offline).  Recall@10 is measured against the exact ranking of the UNQUANTISED fp32 rows
(fp64 accumulation), not against the quantised slab.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# kernel arguments in device memory: ~15 % off the latency of an eagerly launched small kernel on this stack
# (one 16-token query through the encoder: 232 vs 280 us); must be set before the HIP runtime starts
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))

WORKLOADS = {
    #       corpus rows   dim  Qb   k  slab  encoder   rows/GPU in weak mode
    "c2": (100_000, 384, 64, 10, "f16", "minilm", 100_000),
    "c3": (1_000_000, 768, 256, 10, "f16", "bge", 1_000_000),
    "c4": (10_000_000, 384, 64, 10, "f16", "minilm", 1_250_000),
    "c5": (10_000_000, 768, 64, 10, "i8", "bge", 1_250_000),
}
ENC_WORKLOADS = {"enc-minilm": ("minilm", 256, 256), "enc-bge": ("bge", 64, 512)}   # (arch, chunks per batch, tokens per chunk)
QUERY_TOKENS = 16
K_SCAN = 16               # candidates the scan over-fetches for the fp32 re-rank
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TF = {"f16": 2500.0, "i8": 5000.0}


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def synth_rows(torch, n, dim, seed, device, chunk=250_000):
    """Seeded Gaussian rows, generated on the device in chunks (fp32; the slab append normalises)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        yield lo, torch.randn((m, dim), generator=g, device=device, dtype=torch.float32)


def exact_topk_f64(torch, q32, shadow, rows, k, id_base, block=200_000):
    """Ground truth: top-k of <q, row> over the UNQUANTISED fp32 rows with fp64 accumulation
    (score desc, id asc) -- the ranking an exact fp32 store such as the reference's returns."""
    nq = q32.shape[0]
    best_s = torch.full((nq, k), float("-inf"), dtype=torch.float64, device=q32.device)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=q32.device)
    q64 = q32.double()
    for lo in range(0, rows, block):
        hi = min(rows, lo + block)
        sc = q64 @ shadow[lo:hi].double().T
        ts, ti = sc.topk(min(k, hi - lo), dim=1)
        cat_s = torch.cat([best_s, ts], 1)
        cat_i = torch.cat([best_i, ti + lo + id_base], 1)
        # order: score desc, id asc (stable sort by id first, then by score)
        o = torch.argsort(cat_i, dim=1, stable=True)
        cat_s, cat_i = torch.gather(cat_s, 1, o), torch.gather(cat_i, 1, o)
        o = torch.argsort(cat_s, dim=1, descending=True, stable=True)[:, :k]
        best_s, best_i = torch.gather(cat_s, 1, o), torch.gather(cat_i, 1, o)
    return best_s, best_i


def recall_rows(got_i, ref_i):
    """Recall@k per query: |got ∩ ref| / |ref| (reference evaluation/retrieval/retrieval_metrics.py:49-58)."""
    hits = (got_i.unsqueeze(2) == ref_i.unsqueeze(1)) & (ref_i.unsqueeze(1) >= 0)
    return hits.any(dim=1).float().sum(1) / (ref_i >= 0).float().sum(1).clamp(min=1)


def bench_encoder(args, torch, nat, dev, rank, world, dist):
    """--workload enc-*: the index-build side of the path (EmbeddingModel.embed_chunks, reference
    rag/embedding.py:75-87): full-length synthetic chunks through the encoder, tokens/s and MFMA fraction."""
    import numpy as np
    from rag._encoder import HipEncoder, ModelShape
    from rag.embedding import _KNOWN, synthetic_weights
    arch_name, batch, seq = ENC_WORKLOADS[args.workload]
    arch = _KNOWN["all-minilm-l6-v2" if arch_name == "minilm" else "bge-base-en-v1.5"]
    shape = ModelShape(ln_eps=1e-12, **arch)
    seq = min(seq, shape.max_seq)
    enc = HipEncoder(shape, synthetic_weights(shape, seed=7), device=dev)
    rng = np.random.default_rng(4321 + rank)
    ids_h = rng.integers(1000, shape.vocab_size, size=(batch, seq)).astype(np.int32)
    ids_h[:, 0], ids_h[:, -1] = 101, 102
    ids_d = torch.from_numpy(ids_h).to(dev)
    lens_d = torch.full((batch,), seq, dtype=torch.int32, device=dev)
    # F forwards in flight (--enc-inflight), each on its own stream with its own output / workspace: a step is F batches
    nfl = max(1, args.enc_inflight)
    outs = [torch.empty((batch, shape.hidden), dtype=torch.float32, device=dev) for _ in range(nfl)]
    wss = [torch.empty(enc.workspace_bytes(batch, seq), dtype=torch.uint8, device=dev) for _ in range(nfl)]
    sts = [torch.cuda.Stream(device=dev) for _ in range(nfl)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        for o_, w_, s_ in zip(outs, wss, sts):
            with torch.cuda.stream(s_):
                enc.forward(ids_d, lens_d, out=o_, workspace=w_)

    torch.cuda.synchronize()
    for _ in range(max(1, args.warmup)):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    batch *= nfl
    tokens = batch * seq
    h, f, nl = shape.hidden, shape.ffn, shape.layers
    flops = tokens * nl * (2 * (4 * h * h + 2 * h * f)) + nl * 4 * seq * h * tokens   # projections + QK^T + PV
    ms = dt / args.steps * 1e3
    tf = flops / (ms * 1e-3) / 1e12
    if rank == 0:
        print(json.dumps({
            "metric": "index-build encoder tokens/sec (EmbeddingModel.embed_chunks forward, full-length chunks)",
            "value": round(tokens * world * args.steps / dt, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 x f16 -> f32", "data": "synthetic",
            "config": {"workload": args.workload, "encoder": ("all-MiniLM-L6-v2" if arch_name == "minilm" else "bge-base-en-v1.5")
                       + " shape, seeded random weights", "chunks_per_step_per_gpu": batch, "tokens_per_chunk": seq,
                       "batches_in_flight": nfl},
            "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TF["f16"], "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_PEAK_TF["f16"], 4), "traffic": None,
                         "kernel": "whole forward (see profiles/r02_enc_*_kernel_stats.csv for the per-kernel split)",
                         "algorithmic_flops": int(flops)},
            "cpu_baseline": None}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS) + sorted(ENC_WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"))
    ap.add_argument("--no-refine", action="store_true",
                    help="time the plain fp16/int8 scan (k' = k, no fp32 shadow re-rank); recall vs fp32 is then < 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--scan-only", action="store_true", help="diagnostic: skip the encoder (NOT the metric)")
    ap.add_argument("--streams", type=int, default=8,
                    help="query batches in flight, one HIP stream each; a step is one batch on every stream")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--enc-inflight", type=int, default=2,
                    help="enc-* workloads: forwards in flight, one stream each (a step = that many batches; 2 = what EmbeddingModel.embed does)")
    ap.add_argument("--lanes", default="auto", choices=("auto", "split", "batch"),
                    help="stream layout of the in-flight batches: 'batch' = every batch wholly on its own stream; 'split' = encoder "
                         "forwards on encoder lane(s), searches on search lane(s), tied by events (with the encoder's kernels sized to "
                         "sit beside the scan's workgroups: CRS_PANEL_KC=128 CRS_ENC_QKVATTN=0); auto = split for MiniLM-class encoders "
                         "over scans of >= 512 MB per batch (C4), else batch")
    ap.add_argument("--enc-lanes", type=int, default=0, help="--lanes split: streams that run encoder forwards (0: 2)")
    ap.add_argument("--search-lanes", type=int, default=0, help="--lanes split: streams that run searches (0: 1)")
    ap.add_argument("--dist-single", action="store_true",
                    help="diagnostic: run the N > 1 step (process group, collectives, merge) with ONE rank -- RCCL on one card")
    ap.add_argument("--queries", type=int, default=0, help="diagnostic: override the query batch size")
    ap.add_argument("--rows", type=int, default=0, help="diagnostic: override the corpus rows")
    ap.add_argument("--encode", default="auto", choices=("auto", "replicated", "sharded"),
                    help="strong scaling, N > 1: every rank encodes the whole query batch (one collective per batch), or each rank "
                         "encodes Qb / N queries and the embeddings are all-gathered first (two collectives; the encoder's kernels "
                         "then cover 1/N of the CUs, so the chains of the in-flight batches run side by side).  auto = sharded "
                         "from 4 GPUs on (when N divides the batch), replicated below")
    ap.add_argument("--proxy-encode-shard", type=int, default=0,
                    help="diagnostic, N = 1: encode only Qb / W queries per batch and tile them W times in place of the all-gather "
                         "(what one rank of a W-GPU step with --encode sharded executes, minus the collectives)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from rag import _native as nat
    from rag import _shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    nat.require_gpu()
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # multi: the distributed step (collectives + merge).  N > 1 always; --dist-single runs the same path on ONE rank
    # (process group of world size 1): the RCCL calls, dtypes and graph / stream interplay of the N > 1 step, on one card
    multi = world > 1 or args.dist_single
    dist = None
    if multi:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29577"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CRS_DIST_BACKEND", "nccl")   # "gloo": functional rehearsal of N > 1 on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.workload in ENC_WORKLOADS:
        bench_encoder(args, torch, nat, dev, rank, world, dist)
        if multi:
            dist.destroy_process_group()
        return

    corpus_rows, dim, qb, k, slab_kind, enc_name, weak_rows = WORKLOADS[args.workload]
    if args.queries > 0:
        qb = args.queries
    if args.rows > 0:
        corpus_rows = weak_rows = args.rows
    strong = args.scaling == "strong"
    if strong:
        lo_row, hi_row = _shard.shard_slice(corpus_rows, world, rank)
        rows, id_base, nq_all = hi_row - lo_row, lo_row, qb
    else:
        rows, id_base, nq_all = weak_rows, rank * weak_rows, qb * world
        corpus_rows = weak_rows * world
    # auto: sharded from 4 GPUs on.  Measured on one GPU as what a rank executes minus the collectives (tools/ab_shard_enc.sh,
    # --proxy-encode-shard): per batch 0.313 -> 0.261 ms at the 8-GPU shard size, 0.466 -> 0.414 at 4, 0.775 -> 0.749 at 2 --
    # at 2 GPUs the gain is less than a second collective is expected to cost
    want_shard = args.encode == "sharded" or (args.encode == "auto" and world >= 4)
    shard_w = world if (strong and multi and want_shard and qb % world == 0) else 1
    if not multi and args.proxy_encode_shard > 1 and qb % args.proxy_encode_shard == 0:
        shard_w = args.proxy_encode_shard
    q_loc = qb // shard_w                      # queries THIS rank encodes per batch
    gather_q = (multi and (not strong or shard_w > 1)) or (not multi and shard_w > 1)
    refine = not args.no_refine
    k_scan = max(k, K_SCAN) if refine else k
    slab_type = nat.SLAB_I8 if slab_kind == "i8" else nat.SLAB_F16
    pd = nat.padded_dim(dim, slab_type)
    # Lane layout (see the comment at the batch loop).  'split' only pays when the scan is long against the encoder chain
    # AND the encoder's kernels can run beside the scan's workgroups (2 x 48 KB of a CU's 160 KB of LDS are taken): the
    # MiniLM-class query encoder has such a form (K walked in 128-column chunks, QKV projection and attention as separate
    # launches: <= 48 KB each); measured per 64-query batch, same call, one lane per batch -> split: C4 1.40 - 1.42 ->
    # 1.31 - 1.39 ms, one rank of a 4 / 8-GPU step 0.470 -> 0.462 / 0.265 -> 0.252 ms; bge-base (C5 / C3) and C2 lose
    # with it (tools/ab_lanes*.sh).
    scan_bytes = rows * pd * (1 if slab_type == nat.SLAB_I8 else 2)
    pipelined = args.lanes == "split" or (args.lanes == "auto" and enc_name == "minilm" and scan_bytes >= (512 << 20))
    if pipelined:
        os.environ.setdefault("CRS_PANEL_KC", "128")
        os.environ.setdefault("CRS_ENC_QKVATTN", "0")

    # ---- index build (untimed): synthetic embeddings -> slab shard (+ fp32 shadow) in HBM through the product path
    slab = torch.empty((rows, pd), dtype=torch.int8 if slab_type == nat.SLAB_I8 else torch.float16, device=dev)
    scales = torch.empty(rows, dtype=torch.float32, device=dev) if slab_type == nat.SLAB_I8 else None
    shadow = torch.empty((rows, dim), dtype=torch.float32, device=dev)   # unquantised rows: refine operand AND ground truth
    t_build = time.perf_counter()
    for lo, x in synth_rows(torch, rows, dim, 1234 + rank, dev):
        nat.slab_append_f32(x, slab, lo, slab_type, scales=scales, shadow=shadow)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    # ---- query encoder (architecture per BASELINE config, seeded random weights) + synthetic token ids.
    # Everything in the timed path comes from the product package; oracle/ is imported further down,
    # inside the cpu_baseline leg only.
    from rag._encoder import HipEncoder, ModelShape
    from rag.embedding import _KNOWN, synthetic_weights
    arch = _KNOWN["all-minilm-l6-v2" if enc_name == "minilm" else "bge-base-en-v1.5"]
    shape = ModelShape(ln_eps=1e-12, **arch)
    enc_w = synthetic_weights(shape, seed=7)
    enc = HipEncoder(shape, enc_w, device=dev)
    rng = np.random.default_rng(4321 + (0 if strong else rank))      # strong: every rank holds the SAME global batch
    ids_h = rng.integers(1000, shape.vocab_size, size=(qb, QUERY_TOKENS)).astype(np.int32)
    ids_h[:, 0], ids_h[:, -1] = 101, 102                      # [CLS] ... [SEP], no padding
    mask_h = np.ones((qb, QUERY_TOKENS), dtype=np.int32)
    ids_full = torch.from_numpy(ids_h).to(dev)
    lens_full = torch.from_numpy(mask_h.sum(1).astype(np.int32)).to(dev)
    q32 = enc.forward(ids_full, lens_full).clone()            # fp32 unit rows [qb, dim] (planting, diagnostics)
    enc_lo = (rank if multi else 0) * q_loc if shard_w > 1 else 0
    ids_d, lens_d = ids_full[enc_lo:enc_lo + q_loc].contiguous(), lens_full[enc_lo:enc_lo + q_loc].contiguous()
    # plant a near neighbour of every even query (50 % planted); strong: query 2p lives on rank p % world
    g = torch.Generator(device=dev); g.manual_seed(99 + rank)
    mine = [p for p in range(0, qb, 2) if (not strong) or ((p // 2) % world == rank)]
    if mine and rows > len(mine):
        j = torch.randperm(rows, generator=g, device=dev)[: len(mine)]
        planted = q32[mine] + 0.1 * torch.randn((len(mine), dim), generator=g, device=dev)
        tmp = torch.empty((len(mine), pd), dtype=slab.dtype, device=dev)
        tmp_sc = torch.empty(len(mine), dtype=torch.float32, device=dev) if scales is not None else None
        tmp_sh = torch.empty((len(mine), dim), dtype=torch.float32, device=dev)
        nat.slab_append_f32(planted.contiguous(), tmp, 0, slab_type, scales=tmp_sc, shadow=tmp_sh)
        slab[j] = tmp
        shadow[j] = tmp_sh
        if scales is not None:
            scales[j] = tmp_sc

    class Ctx:
        """Buffers of one in-flight query batch (a batch touches nothing outside its Ctx + read-only state)."""
        def __init__(self):
            self.q_out = torch.empty((q_loc, dim), dtype=torch.float32, device=dev)
            self.q16 = torch.empty((q_loc, pd), dtype=torch.float16, device=dev)
            self.enc_ws = torch.empty(enc.workspace_bytes(q_loc, QUERY_TOKENS), dtype=torch.uint8, device=dev)
            self.ws = torch.empty(nat.scan_workspace_bytes(nq_all, dim, k_scan, rows), dtype=torch.uint8, device=dev)
            self.cand_s = torch.empty((nq_all, k_scan), dtype=torch.float32, device=dev)
            self.cand_i = torch.empty((nq_all, k_scan), dtype=torch.int64, device=dev)
            self.wire = nat.WireBlock(nq_all, k, dev, world, gather=multi)     # this rank's (ids | scores) block + the gathered blocks
            self.graphs = None
            if multi:
                self.fin_s = torch.empty((nq_all, k), dtype=torch.float32, device=dev)
                self.fin_i = torch.empty((nq_all, k), dtype=torch.int64, device=dev)
            if gather_q:
                self.q_all32 = torch.empty((nq_all, dim), dtype=torch.float32, device=dev)
                self.q_all16 = torch.empty((nq_all, pd), dtype=torch.float16, device=dev)

    # A batch = device segments with the collectives between them; every segment reads and writes fixed buffers
    # of its Ctx, so each is captured once into a hipGraph and replayed.
    def seg_encode(c):      # token ids -> fp32 embeddings + the scan's fp16 query block
        if args.scan_only:
            c.q_out.copy_(q32[enc_lo:enc_lo + q_loc])
            nat.queries_to_f16(c.q_out, slab_type, out=c.q16)
        else:
            enc.forward(ids_d, lens_d, out=c.q_out, workspace=c.enc_ws, q16_out=c.q16, slab_type=slab_type)

    def seg_search(c, do_refine=refine):       # all queries of the batch x this rank's shard -> this rank's wire block
        qa32 = c.q_all32 if gather_q else c.q_out
        if gather_q:
            if not multi:      # --proxy-encode-shard: the local queries tiled in place of the all-gather
                c.q_all32.view(shard_w, q_loc, dim).copy_(c.q_out.unsqueeze(0).expand(shard_w, q_loc, dim))
            nat.queries_to_f16(qa32, slab_type, out=c.q_all16)
        qa16 = c.q_all16 if gather_q else c.q16
        if do_refine:
            nat.cosine_topk(qa16, slab, rows, dim, k_scan, slab_type=slab_type, scales=scales, id_base=id_base,
                            workspace=c.ws, out_scores=c.cand_s, out_ids=c.cand_i)
            nat.refine_f32(qa32, shadow, rows, id_base, c.cand_i, k, out_scores=c.wire.scores, out_ids=c.wire.ids)
        else:
            nat.cosine_topk(qa16, slab, rows, dim, k, slab_type=slab_type, scales=scales, id_base=id_base,
                            workspace=c.ws, out_scores=c.wire.scores, out_ids=c.wire.ids)

    def seg_merge(c):       # N > 1: the gathered wire blocks -> global top-k
        nat.merge_topk_wire(c.wire.gathered, world, nq_all, k, k, out_scores=c.fin_s, out_ids=c.fin_i)

    # segments of a batch, in order, each with the collective that follows it (N > 1) and the lane it runs on
    # ("E": encoder lane, "S": search lane)
    segs = [seg_encode, seg_search] + ([seg_merge] if multi else [])
    lanes = ["E", "S", "S"][: len(segs)]
    exchanges = [None] * len(segs)
    if multi and gather_q:  # queries encoded in shards (or weak scaling: per-rank queries): embeddings gathered first
        exchanges[0] = lambda c: dist.all_gather_into_tensor(c.q_all32, c.q_out)
    if multi:               # the ONE exchange of a sharded search: every rank's wire block
        exchanges[1] = lambda c: dist.all_gather_into_tensor(c.wire.gathered, c.wire.buf)
    n_exchanges = sum(1 for e in exchanges if e is not None)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # Throughput mode: S batches in flight, each with its own buffers (Ctx).  The GPU runs a process's streams on four
    # hardware queues; a query-encoder forward is a chain of 38 dependent launches of a few microseconds each (latency,
    # not work), a scan is one kernel that wants every byte of HBM bandwidth.  Round 1 / early round 2 gave every batch
    # its own stream -- encoder chain, scan, merge, refine in one lane -- so a lane spent a third of its time inside an
    # encoder chain and the scans of the other lanes did not always cover it (C4 batch 1.38 - 1.41 ms for a 1.28 ms scan).
    # Now the lanes have ROLES: encoder forwards of upcoming batches run on the encoder lane(s), searches alternate
    # between the search lanes (one's merge / refine tail under the other's scan), tied together by events per Ctx;
    # the same kernels, the same work per batch, nothing skipped.  --lanes batch restores one lane per batch.
    # For N > 1 the RCCL collectives between the segments are launched eagerly in batch order (every rank issues them in
    # the same order; they are serialised on the process group's own stream).
    n_ctx = max(1, args.streams)
    use_graph = not args.no_graph
    ctxs = [Ctx() for _ in range(n_ctx)]
    if pipelined:
        # measured (tools/ab_lanes.sh): two encoder lanes feed ONE search lane at every shard size (one encoder lane starves
        # scans of <= 2.5 M rows: 0.41 against 0.25 ms per batch at 1.25 M).  On the whole 10 M-row corpus a second search
        # lane (one scan's merge / refine tail under the next scan) is worth another 1 - 2 %, but then consecutive scans
        # overlap and a per-kernel duration -- the roofline's denominator, the rocprof kernel average -- stops meaning
        # "one scan"; with one search lane the scans run one after another (1.30 ms in the trace against 1.27 - 1.28 isolated).
        n_enc = args.enc_lanes if args.enc_lanes > 0 else 2
        n_srch = args.search_lanes if args.search_lanes > 0 else 1
        enc_lanes = [torch.cuda.Stream(device=dev) for _ in range(n_enc)]
        srch_lanes = [torch.cuda.Stream(device=dev) for _ in range(n_srch)]
    else:
        n_enc = n_srch = n_ctx
        enc_lanes = srch_lanes = [torch.cuda.Stream(device=dev) for _ in range(n_ctx)]
    for c in ctxs:
        c.ev_enc, c.ev_done = torch.cuda.Event(), torch.cuda.Event()
    issued = [0]

    def batch(c):
        """Issue one batch: encode on an encoder lane, search (+ exchange + merge) on a search lane."""
        b = issued[0]
        issued[0] += 1
        lane = {"E": enc_lanes[b % n_enc], "S": srch_lanes[b % n_srch]}
        prev = None
        for j, seg in enumerate(segs):
            st = lane[lanes[j]]
            with torch.cuda.stream(st):
                if j == 0:
                    st.wait_event(c.ev_done)          # the previous batch that used this Ctx is through (no-op before its first use)
                elif st is not prev:
                    st.wait_event(c.ev_enc)           # lane change: the encoder lane's output (and its collective) is complete
                if c.graphs is not None:
                    c.graphs[j].replay()
                else:
                    seg(c)
                if exchanges[j] is not None:
                    exchanges[j](c)
                if j == 0:
                    c.ev_enc.record(st)
                if j == len(segs) - 1:
                    c.ev_done.record(st)
            prev = st
        return (c.fin_s, c.fin_i) if multi else (c.wire.scores, c.wire.ids)

    torch.cuda.synchronize()
    for c in ctxs:
        for _ in range(2):
            batch(c)
        torch.cuda.synchronize()
        if use_graph:
            # thread_local: with N > 1 the process group's watchdog thread polls events while we capture;
            # only this thread's calls belong to the capture.  If a capture fails anyway, run eagerly.
            try:
                gl = []
                for j, seg in enumerate(segs):
                    st = (enc_lanes if lanes[j] == "E" else srch_lanes)[0]
                    g_ = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_, stream=st, capture_error_mode="thread_local"):
                        seg(c)
                    gl.append(g_)
                c.graphs = gl
            except Exception as exc:   # noqa: BLE001 -- report and keep going without graphs
                print(f"[bench] hipGraph capture failed on rank {rank} ({exc!r}); launching eagerly", file=sys.stderr, flush=True)
                use_graph = False
                for cc in ctxs:
                    cc.graphs = None
                torch.cuda.synchronize()   # (no break: every rank must still run the same warm-up collectives)
    sync()

    def run(n_steps):
        for _ in range(n_steps):
            for c in ctxs:
                batch(c)

    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    q_per_step = nq_all * n_ctx
    ms_step = dt / args.steps * 1e3
    qps = q_per_step * args.steps / dt

    # ---- correctness of the timed path (untimed): its final lists against the exact ranking of the UNQUANTISED
    # fp32 rows of every shard (fp64 accumulation) -- catches quantisation loss as well as any mix-up of query
    # order, id bases, wire layout or merge.
    def gathered_queries():
        if multi and gather_q:
            qa = torch.empty((nq_all, dim), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(qa, ctxs[0].q_out.contiguous())
            return qa
        return ctxs[0].q_all32 if gather_q else ctxs[0].q_out

    fin_s, fin_i = batch(ctxs[0])           # (issues on its lanes)
    torch.cuda.synchronize()
    fin_s, fin_i = fin_s.clone(), fin_i.clone()
    q_truth = gathered_queries()
    gt_s, gt_i = exact_topk_f64(torch, q_truth, shadow, rows, k, id_base)
    if multi:
        all_s = torch.empty((world * nq_all, k), dtype=torch.float64, device=dev)
        all_i = torch.empty((world * nq_all, k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_s, gt_s.contiguous()); dist.all_gather_into_tensor(all_i, gt_i.contiguous())
        all_s = all_s.view(world, nq_all, k).permute(1, 0, 2).reshape(nq_all, world * k)
        all_i = all_i.view(world, nq_all, k).permute(1, 0, 2).reshape(nq_all, world * k)   # rank-major = id-ascending
        o = torch.argsort(all_s, dim=1, descending=True, stable=True)[:, :k]
        gt_s, gt_i = torch.gather(all_s, 1, o), torch.gather(all_i, 1, o)
    rec_timed = recall_rows(fin_i, gt_i)
    recall_timed = float(rec_timed.mean().item())
    ids_identical = float((fin_i == gt_i).all(dim=1).float().mean().item())
    score_err = float((fin_s.double() - gt_s).abs().max().item())
    # the other mode, for the record (one untimed eager batch): plain fp16/int8 scan with k' = k, or the refined one
    c0 = ctxs[0]
    with torch.cuda.stream(srch_lanes[0]):
        seg_encode(c0)
        if multi and gather_q:
            dist.all_gather_into_tensor(c0.q_all32, c0.q_out)
        if refine:
            seg_search(c0, do_refine=False)
        else:
            c0.cand_s = torch.empty((nq_all, max(k, K_SCAN)), dtype=torch.float32, device=dev)
            c0.cand_i = torch.empty((nq_all, max(k, K_SCAN)), dtype=torch.int64, device=dev)
            c0.ws = torch.empty(nat.scan_workspace_bytes(nq_all, dim, max(k, K_SCAN), rows), dtype=torch.uint8, device=dev)
            qa16 = c0.q_all16 if gather_q else c0.q16
            qa32 = c0.q_all32 if gather_q else c0.q_out
            if gather_q:
                if not multi:
                    c0.q_all32.view(shard_w, q_loc, dim).copy_(c0.q_out.unsqueeze(0).expand(shard_w, q_loc, dim))
                nat.queries_to_f16(qa32, slab_type, out=c0.q_all16)
            nat.cosine_topk(qa16, slab, rows, dim, max(k, K_SCAN), slab_type=slab_type, scales=scales, id_base=id_base,
                            workspace=c0.ws, out_scores=c0.cand_s, out_ids=c0.cand_i)
            nat.refine_f32(qa32, shadow, rows, id_base, c0.cand_i, k, out_scores=c0.wire.scores, out_ids=c0.wire.ids)
        if multi:
            dist.all_gather_into_tensor(c0.wire.gathered, c0.wire.buf)
            seg_merge(c0)
        oth_s, oth_i = ((c0.fin_s, c0.fin_i) if multi else (c0.wire.scores, c0.wire.ids))
    torch.cuda.synchronize()
    recall_other = float(recall_rows(oth_i, gt_i).mean().item())
    err_other = float((oth_s.double() - gt_s).abs().max().item())
    tol = 1e-5 if refine else (5e-3 if slab_type == nat.SLAB_I8 else 1e-3)
    check_ok = bool(score_err < tol and recall_timed >= (0.999 if refine else 0.8))
    recall_report = {"timed_path": round(recall_timed, 5),
                     "timed_mode": (f"{slab_kind} scan k'={k_scan} + fp32 shadow re-rank" if refine else f"{slab_kind} scan only"),
                     "queries_with_identical_ordered_ids": round(ids_identical, 5), "max_abs_score_err_vs_fp64": score_err,
                     ("scan_only_no_refine" if refine else "with_fp32_refine"): round(recall_other, 5),
                     ("scan_only_max_abs_score_err" if refine else "with_fp32_refine_max_abs_score_err"): err_other}

    # ---- roofline of the dominant kernel (the scan), hipEvent-timed on the launch stream; the re-rank beside it
    qa16 = ctxs[0].q_all16 if gather_q else ctxs[0].q16
    ms_total, ms_scan = nat.time_cosine_topk(qa16, slab, rows, dim, k_scan, max(10, min(args.steps, 50)),
                                             slab_type=slab_type, scales=scales)
    ms_refine = None
    if refine:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        qa32 = ctxs[0].q_all32 if gather_q else ctxs[0].q_out
        e0.record()
        for _ in range(50):
            nat.refine_f32(qa32, shadow, rows, id_base, ctxs[0].cand_i, k, out_scores=ctxs[0].wire.scores, out_ids=ctxs[0].wire.ids)
        e1.record(); e1.synchronize()
        ms_refine = e0.elapsed_time(e1) / 50
    elem = 1 if slab_type == nat.SLAB_I8 else 2
    alg_bytes = rows * pd * elem + (rows * 4 if slab_type == nat.SLAB_I8 else 0) + nq_all * pd * 2 + nq_all * k_scan * 8
    achieved = alg_bytes / (ms_scan * 1e-3) / 1e9
    # HBM traffic per launch comes from the committed PMC passes of this same command (bench.py cannot
    # read hardware counters itself); null when that workload has not been profiled yet
    traffic, traffic_src = None, None
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
            with open(path) as fh:
                pm = json.load(fh)
            key = f"{args.workload}-n{world}" if f"{args.workload}-n{world}" in pm else None
            if key and pm[key].get("rows") == rows:
                traffic = pm[key]["hbm_read_bytes_per_launch"]
                traffic_src = os.path.relpath(path, ROOT)
                break
    except Exception:
        pass
    # Which roof bounds the launch: flop per algorithmic byte (= queries per launch for fp16, 2x that for int8)
    # against the ridge of the dense MFMA peak over the HBM peak (MI355X_MICROARCH.md: 2.5 PF fp16 / 5 PF int8, 8 TB/s).
    alg_flops = 2.0 * nq_all * rows * pd
    mfma_peak_tf = MFMA_PEAK_TF[slab_kind]
    achieved_tf = alg_flops / (ms_scan * 1e-3) / 1e12
    hbm_frac, mfma_frac = achieved / HBM_PEAK_GBS, achieved_tf / mfma_peak_tf
    mfma_bound = (alg_flops / alg_bytes) > (mfma_peak_tf * 1e12) / (HBM_PEAK_GBS * 1e9)
    roofline = {"bound": "mfma" if mfma_bound else "hbm",
                "achieved": round(achieved_tf if mfma_bound else achieved, 1),
                "peak": mfma_peak_tf if mfma_bound else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": round(mfma_frac if mfma_bound else hbm_frac, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": nat.scan_plan_describe(nq_all, dim, k_scan, rows, slab_type),
                "kernel_ms": round(ms_scan, 5), "scan_merge_refine_ms": round(ms_total, 5),
                "fp32_rerank_ms": round(ms_refine, 5) if ms_refine is not None else None,
                "fp32_shadow_bytes": int(rows * dim * 4) if refine else 0,
                "algorithmic_bytes": int(alg_bytes), "algorithmic_flops": int(alg_flops),
                "hbm_frac": round(hbm_frac, 4), "mfma_frac": round(mfma_frac, 4)}

    # ---- CPU baseline (rank 0, N=1 only): the oracle on a bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import encoder_ref as er, scan_ref   # the CPU restatement: only ever the baseline / checker
        cfg = er.EncoderConfig(shape.vocab_size, shape.hidden, shape.layers, shape.heads, shape.ffn, shape.max_pos, 2,
                               shape.ln_eps, shape.max_seq, shape.pooling)
        sample_rows = min(rows, 200_000)
        rows_h = shadow[:sample_rows].cpu().numpy()          # the reference stores fp32 (rag/indexing.py:114-119)
        q_h = q32.cpu().numpy()
        # the HIP path on the same sample prefix against the oracle (exactness of the kernels themselves)
        cs_, ci_ = nat.cosine_topk(ctxs[0].q16, slab, sample_rows, dim, k_scan, slab_type=slab_type, scales=scales)
        gs_, gi_ = nat.refine_f32(q32, shadow, sample_rows, 0, ci_, k)
        rs, ri = scan_ref.cosine_topk_ref(q_h, rows_h, k)
        gi_h = gi_.cpu().numpy()
        recall_oracle = float(np.mean([scan_ref.recall_at_k(gi_h[r], ri[r]) for r in range(qb)]))
        max_err = float(np.abs(gs_.cpu().numpy() - rs).max())
        n_done, t_enc, t_scan = 0, 0.0, 0.0
        while (t_enc + t_scan) < args.cpu_seconds:
            ta = time.perf_counter()
            er.encode_ref(ids_h, mask_h, enc_w, cfg)
            tb = time.perf_counter()
            scan_ref.cosine_topk_ref(q_h, rows_h, k)
            tc = time.perf_counter()
            t_enc += tb - ta; t_scan += tc - tb
            n_done += 1
        t_cpu = t_enc + t_scan
        # queries/s over the FULL corpus: the scan part of the sample's time scales by rows/sample_rows
        cpu_qps = qb * n_done / (t_enc + t_scan * (rows / sample_rows))
        cpu = {"value": round(cpu_qps, 2), "unit": "queries/s", "cores": int(torch.get_num_threads()),
               "cpu_model": cpu_model(), "kind": "port",
               "sample": f"oracle encoder_ref.encode_ref ({qb}x{QUERY_TOKENS} tokens, torch fp32) + scan_ref.cosine_topk_ref "
                         f"(numpy fp32 sgemm + exact top-k over the fp32 rows) on {sample_rows} of {rows} rows; {n_done} passes in "
                         f"{t_cpu:.1f}s (encoder {t_enc:.1f}s, scan {t_scan:.1f}s), scan time scaled by rows/sample",
               "recall_at_10_gpu_vs_oracle_on_sample": recall_oracle, "max_abs_score_err_on_sample": max_err}

    if rank == 0:
        line = {
            "metric": "queries/sec over N-vector corpus (exact cosine top-k; Recall@10 vs the exact fp32 ranking)",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 5), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f16 x f16 -> f32" if slab_type == nat.SLAB_F16 else "i8 x i16(f16 query) -> i32 -> f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "corpus_rows": corpus_rows, "rows_per_gpu": rows, "dim": dim,
                       "queries_per_batch": nq_all, "batches_per_step": n_ctx, "queries_per_step": q_per_step,
                       "ms_per_batch": round(ms_step / n_ctx, 5), "lanes": (f"{n_enc} encoder + {n_srch} search (encoder kernels <= 48 KB of LDS)" if pipelined else "one per batch"), "top_k": k, "k_scan": k_scan,
                       "slab": slab_kind, "refine_fp32": refine, "encoder_in_step": not args.scan_only,
                       "encoder": ("all-MiniLM-L6-v2" if enc_name == "minilm" else "bge-base-en-v1.5") + " shape, seeded random weights",
                       "query_tokens": QUERY_TOKENS, "hip_graph": use_graph,
                       "collectives_per_batch": n_exchanges, "dist_single_rank": bool(multi and world == 1),
                       "query_encode": ("replicated" if not gather_q else ("per-rank queries (weak scaling)" if not strong else
                                        f"sharded: {q_loc} of {qb} queries per rank" + (" [single-GPU proxy: tiled instead of gathered]" if not multi else ""))),
                       "recall_at_10_vs_fp32": recall_report, "check_ok": check_ok,
                       "index_build_s_per_gpu": round(t_build, 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if multi:
        ok_t = torch.tensor([1 if check_ok else 0], device=dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        check_ok = bool(ok_t.item())
        dist.destroy_process_group()
    if not check_ok:
        print(f"[bench] rank {rank}: result check FAILED (recall {recall_timed}, max score err {score_err})", file=sys.stderr, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
