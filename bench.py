#!/usr/bin/env python3
"""bench.py -- queries/sec of the embed -> index -> retrieve hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N > 1: launched by
torch.distributed.run, one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

One query BATCH through the retrieve path, everything resident in HBM:
  query encoder forward (token ids [Qb, 16] -> fp32 sentence embeddings + the scan's fp16 query block)
  -> exact cosine scan of this rank's slab shard, over-fetching k' = 24 candidates per query
  -> per-workgroup list merge + tile refine -> fp32 re-rank of the k' candidates against the fp32 shadow
     of the shard (the ranking the reference's fp32 ChromaDB collection gives) WITH a per-query exactness
     certificate (csrc/exact.hip), unproven queries escalated on the device inside the same graph (fp16
     slabs; int8 stays empirical), best k = 10 written straight into this rank's wire block
  -> (N > 1) ONE RCCL all-gather of the wire blocks + k-way merge on every rank.
A STEP is S such batches in flight (S = --streams, each with its own buffers and its OWN 64 queries), so the timed region is
steady state whatever --steps is; queries per step = S x Qb.  The batch, its lanes and its hipGraphs are the library's
(rag/_engine.py RetrievalEngine -- the object ContextRetriever.retrieve_batch drives); bench.py only feeds it token ids.  --lanes: every batch wholly on its own HIP stream ("batch"), or encoder
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
checkpoints offline).  Recall@10 is measured, untimed, on >= 8192 distinct queries run through the same engine, against the
exact ranking of the UNQUANTISED fp32 rows (fp64 accumulation), not against the quantised slab; the line carries the sample
size, the certified fraction and the number of escalated queries.  --through-pipeline times the plugin surface instead
(texts -> RAGPipeline.retrieve_batch -> lists of dicts).
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
K_SCAN = 24               # candidates the scan over-fetches for the fp32 re-rank (VectorStore's default refine_overfetch)
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


def bench_pipeline(args, torch, nat, dev):
    """--through-pipeline: the PLUGIN SURFACE timed the way the reference's harness times it
    (/root/reference/evaluation/retrieval/benchmark.py:241-247: perf_counter around rag_pipeline.retrieve(question)), at
    BASELINE batch sizes through the additive batched entry: texts -> RAGPipeline.retrieve_batch -> lists of dicts.
    Host tokenisation, the result D2H, the sidecar lookup and ContextRetriever's scoring / rerank / MMR are all inside
    the timed call.  Two retrieval configs: (rerank, diversity) off, and the reference's defaults on (config.json:20-25)."""
    import logging
    import numpy as np
    logging.disable(logging.WARNING)
    from rag import RAGPipeline
    from rag.chunking import Chunk
    rows, dim, qb, k, slab_kind, enc_name, _ = WORKLOADS[args.workload]
    if enc_name != "minilm" or slab_kind != "f16":
        sys.exit("--through-pipeline runs the MiniLM fp16 workloads (c2, c4)")
    if args.pipeline_rows > 0:
        rows = args.pipeline_rows
    rng = np.random.default_rng(0)
    words = ("retrieval augmented generation language model quantization weights perplexity attention embedding cosine similarity "
             "vector index chunk context answer question compression memory latency throughput accuracy benchmark kernel").split()

    class Stub:
        def generate(self, prompt, **kw):
            return "n/a"

    results = []
    for label, rcfg in (("rerank off, diversity off", {"top_k": k, "similarity_threshold": 0.0, "rerank": False, "diversity_penalty": 0.0}),
                        ("rerank on, diversity_penalty 0.1 (reference defaults)", {"top_k": k, "similarity_threshold": 0.0, "rerank": True, "diversity_penalty": 0.1})):
        cfg = {"embedding": {"model_name": "synthetic:minilm", "device": "cuda", "batch_size": 64, "normalize": True},
               "retrieval": dict(rcfg, batch_queries=qb), "vector_store": {"collection_name": "pipe"}}
        p = RAGPipeline(cfg)
        p.setup(Stub())
        # index: synthetic unit embeddings straight into the store's slab (encoding 10 M chunk texts is the enc-* workload),
        # with a ~40-word document per row in the host sidecars
        t_index = time.perf_counter()
        g = torch.Generator(device=dev); g.manual_seed(1234)
        base_docs = [" ".join(rng.choice(words, size=40)) for _ in range(4096)]
        for lo in range(0, rows, 250_000):
            m = min(250_000, rows - lo)
            emb = torch.randn((m, dim), generator=g, device=dev)
            chunks = [Chunk(text=base_docs[(lo + r) & 4095] + f" {lo + r}", chunk_id=f"chunk_{lo + r}", start_char=0, end_char=1) for r in range(m)]
            p.vector_store.create_index(chunks, emb)
        torch.cuda.synchronize()
        t_index = time.perf_counter() - t_index
        n_q = qb * (args.streams if args.streams > 0 else 8)     # queries per retrieve_batch call: eight engine batches
        queries = [" ".join(rng.choice(words, size=int(rng.integers(5, 12)))) for _ in range(n_q)]
        for _ in range(max(1, args.warmup)):
            p.retrieve_batch(queries)
        lat = []
        for _ in range(args.steps):
            t0 = time.perf_counter()
            out = p.retrieve_batch(queries)
            lat.append(time.perf_counter() - t0)
        # the same queries one at a time through the unchanged per-query entry (what the reference's loop does)
        t0 = time.perf_counter()
        for q in queries[:64]:
            p.retrieve(q)
        single_ms = (time.perf_counter() - t0) / 64 * 1e3
        # host / device split of one call: tokenisation, engine (device + readback), post-processing incl. the MMR encoder pass
        r = p.retriever
        t0 = time.perf_counter(); toks = p.embedding_model.tokenize(queries); t_tok = time.perf_counter() - t0
        t0 = time.perf_counter(); hits = [h for piece in r._search_many(queries, k * 2 if r.rerank else k) for h in piece]; t_search = time.perf_counter() - t0
        med = sorted(lat)[len(lat) // 2]
        results.append({"retrieval_config": label, "queries_per_call": n_q, "queries_per_batch": qb, "corpus_rows": rows,
                        "queries_per_s": round(n_q / med, 1), "ms_per_call_median": round(med * 1e3, 3),
                        "ms_per_call_min": round(min(lat) * 1e3, 3), "single_query_retrieve_ms": round(single_ms, 3),
                        "split_ms": {"tokenize": round(t_tok * 1e3, 3), "tokenize_plus_engine_search_and_readback": round(t_search * 1e3, 3),
                                     "post_processing_incl_mmr_encoder_pass": round(max(med - t_search, 0.0) * 1e3, 3)},
                        "chunks_returned_first_query": len(out[0]), "exactness": dict(p.vector_store.last_exactness),
                        "index_s": round(t_index, 2)})
        del p
        torch.cuda.empty_cache()
    best = results[0]
    print(json.dumps({
        "metric": "queries/sec through the plugin surface (texts -> RAGPipeline.retrieve_batch -> lists of dicts)",
        "value": best["queries_per_s"], "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": best["ms_per_call_median"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16 x f16 -> f32", "data": "synthetic",
        "config": {"workload": args.workload + " through RAGPipeline", "timing": "time.perf_counter around the call (reference: evaluation/retrieval/benchmark.py:241-247)",
                   "encoder": "all-MiniLM-L6-v2 shape, seeded random weights, hash tokenizer", "results": results},
        "roofline": None, "cpu_baseline": None}), flush=True)


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
    ap.add_argument("--streams", type=int, default=0,
                    help="query batches (buffer sets) in flight; a step is one batch from every buffer set (0: the engine's rule, 8 or 24)")
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
    ap.add_argument("--enc-small-lds", default="auto", choices=("auto", "on", "off"),
                    help="the encoder's <= 48 KB-of-LDS kernel forms (auto: with role lanes only)")
    ap.add_argument("--encode-group", type=int, default=0,
                    help="query batches served by ONE encoder forward (0: the engine's rule -- 2 for >= 2048-token batches of a bge-class model, else 1)")
    ap.add_argument("--search-fuse", type=int, default=0,
                    help="search segments replayed as ONE hipGraph (0: the engine's rule -- 4 inside an encode group on one rank, else 1)")
    ap.add_argument("--k-scan", type=int, default=0,
                    help="candidates the scan over-fetches for the fp32 re-rank (0: the library's rule, rag/_native.py overfetch: 24 on fp16 shards of >= 4 M rows, else 16)")
    ap.add_argument("--exact", default="auto", choices=("auto", "on", "off"),
                    help="in-stream escalation of queries whose list the certificate could not prove (auto: fp16 slabs on, int8 empirical)")
    ap.add_argument("--recall-queries", type=int, default=8192,
                    help="distinct queries the (untimed) recall / certificate check runs through the timed path's engine")
    ap.add_argument("--through-pipeline", action="store_true",
                    help="time the PLUGIN SURFACE instead: texts -> ContextRetriever.retrieve_batch -> list of dicts "
                         "(workloads c2 / c4), perf_counter around the call as evaluation/retrieval/benchmark.py:241-247 does")
    ap.add_argument("--pipeline-rows", type=int, default=0, help="--through-pipeline: override the corpus rows")
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

    if args.through_pipeline:
        bench_pipeline(args, torch, nat, dev)
        return

    corpus_rows, dim, qb, k, slab_kind, enc_name, weak_rows = WORKLOADS[args.workload]
    if args.queries > 0:
        qb = args.queries
    if args.rows > 0:
        corpus_rows = weak_rows = args.rows
    strong = args.scaling == "strong"
    if strong:
        lo_row, hi_row = _shard.shard_slice(corpus_rows, world, rank)
        rows, id_base = hi_row - lo_row, lo_row
    else:
        rows, id_base = weak_rows, rank * weak_rows
        corpus_rows = weak_rows * world
    # auto: sharded from 4 GPUs on.  Measured on one GPU as what a rank executes minus the collectives (tools/ab_shard_enc.sh,
    # --proxy-encode-shard): per batch 0.313 -> 0.261 ms at the 8-GPU shard size, 0.466 -> 0.414 at 4, 0.775 -> 0.749 at 2 --
    # at 2 GPUs the gain is less than a second collective is expected to cost.  PROVISIONAL: the proxy excludes the collective
    # this adds; no N > 1 RCCL run exists yet
    want_shard = args.encode == "sharded" or (args.encode == "auto" and world >= 4)
    refine = not args.no_refine
    slab_type = nat.SLAB_I8 if slab_kind == "i8" else nat.SLAB_F16
    pd = nat.padded_dim(dim, slab_type)

    # ---- index build (untimed): synthetic embeddings -> slab shard (+ fp32 shadow) in HBM through the product path
    slab = torch.empty((rows, pd), dtype=torch.int8 if slab_type == nat.SLAB_I8 else torch.float16, device=dev)
    scales = torch.empty(rows, dtype=torch.float32, device=dev) if slab_type == nat.SLAB_I8 else None
    shadow = torch.empty((rows, dim), dtype=torch.float32, device=dev)   # unquantised rows: refine operand AND ground truth
    row_err = torch.zeros(1, dtype=torch.float32, device=dev)            # tracked |stored row - fp32 row|_2 maximum
    t_build = time.perf_counter()
    for lo, x in synth_rows(torch, rows, dim, 1234 + rank, dev):
        nat.slab_append_f32(x, slab, lo, slab_type, scales=scales, shadow=shadow, row_err=row_err)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    # ---- query encoder (architecture per BASELINE config, seeded random weights) + synthetic token ids.
    # Everything in the timed path comes from the product package (rag/_engine.py drives it); oracle/ is imported
    # further down, inside the cpu_baseline leg only.
    from rag._encoder import HipEncoder, ModelShape
    from rag._engine import RetrievalEngine, ShardView
    from rag.embedding import _KNOWN, synthetic_weights
    arch = _KNOWN["all-minilm-l6-v2" if enc_name == "minilm" else "bge-base-en-v1.5"]
    shape = ModelShape(ln_eps=1e-12, **arch)
    enc_w = synthetic_weights(shape, seed=7)
    enc = HipEncoder(shape, enc_w, device=dev)
    # buffer sets in flight: the engine's rule (8; 16 / 24 with encode groups) unless --streams says otherwise
    n_ctx = RetrievalEngine.plan_layout(shape.hidden, rows * pd * (1 if slab_type == nat.SLAB_I8 else 2), encode=not args.scan_only, multi=multi,
                                        lanes=args.lanes, encode_group=args.encode_group if args.encode_group > 0 else "auto",
                                        n_ctx=args.streams, enc_lanes=args.enc_lanes, search_lanes=args.search_lanes,
                                        batch_tokens=(qb // (world if (strong and multi and want_shard) else 1)) * QUERY_TOKENS)["n_ctx"]
    # Query set: R distinct queries (>= --recall-queries, a whole number of batches; the first n_ctx batches are the ones the
    # timed loop keeps in flight).  strong: every rank holds the SAME global batches; weak: per-rank queries.
    n_batches = max(n_ctx, -(-max(args.recall_queries, 1) // qb)) if strong else n_ctx
    rng = np.random.default_rng(4321 + (0 if strong else rank))
    ids_all = rng.integers(1000, shape.vocab_size, size=(n_batches, qb, QUERY_TOKENS)).astype(np.int32)
    ids_all[:, :, 0], ids_all[:, :, -1] = 101, 102                      # [CLS] ... [SEP], no padding
    lens_h = np.full(qb, QUERY_TOKENS, dtype=np.int32)
    ids_dev_all = torch.from_numpy(ids_all).to(dev)
    lens_dev = torch.from_numpy(lens_h).to(dev)
    q32_all = torch.empty((n_batches, qb, dim), dtype=torch.float32, device=dev)   # fp32 unit rows (planting, ground truth)
    for b in range(n_batches):
        q32_all[b] = enc.forward(ids_dev_all[b], lens_dev)
    # plant a near neighbour of every even query (50 % planted); strong: planted query p lives on rank p % world
    g = torch.Generator(device=dev); g.manual_seed(99 + rank)
    flat_q = q32_all.view(-1, dim)
    mine = [p for p in range(0, flat_q.shape[0], 2) if (not strong) or ((p // 2) % world == rank)]
    if mine and rows > len(mine):
        j = torch.randperm(rows, generator=g, device=dev)[: len(mine)]
        planted = flat_q[mine] + 0.1 * torch.randn((len(mine), dim), generator=g, device=dev)
        tmp = torch.empty((len(mine), pd), dtype=slab.dtype, device=dev)
        tmp_sc = torch.empty(len(mine), dtype=torch.float32, device=dev) if scales is not None else None
        tmp_sh = torch.empty((len(mine), dim), dtype=torch.float32, device=dev)
        nat.slab_append_f32(planted.contiguous(), tmp, 0, slab_type, scales=tmp_sc, shadow=tmp_sh, row_err=row_err)
        slab[j] = tmp
        shadow[j] = tmp_sh
        if scales is not None:
            scales[j] = tmp_sc
    view = ShardView(slab, scales, shadow, rows, dim, slab_type, id_base, float(row_err.item()))

    def make_engine(do_refine, exact):
        return RetrievalEngine(enc if not args.scan_only else None, view if do_refine else ShardView(slab, scales, None, rows, dim, slab_type, id_base),
                               qb, QUERY_TOKENS, k, k_scan=K_SCAN, k_scan_exact=args.k_scan, refine=do_refine, exact=exact, n_ctx=n_ctx, lanes=args.lanes,
                               enc_lanes=args.enc_lanes, search_lanes=args.search_lanes, graphs=not args.no_graph,
                               dist=dist if multi else None, world=world, rank=rank, queries_per_rank=not strong,
                               encode_shard=(world if (strong and multi and want_shard) else 1),
                               proxy_encode_shard=args.proxy_encode_shard, encode=not args.scan_only,
                               enc_small_lds={"auto": "auto", "on": True, "off": False}[args.enc_small_lds],
                               encode_group=args.encode_group if args.encode_group > 0 else "auto",
                               search_fuse=args.search_fuse if args.search_fuse > 0 else "auto")

    exact_mode = {"auto": "auto", "on": True, "off": False}[args.exact]
    eng = make_engine(refine, exact_mode)
    nq_all, q_loc, enc_lo, gather_q, k_scan = eng.nq_all, eng.q_loc, eng.enc_lo, eng.gather_q, eng.k_scan

    def load_batch(i, b):          # tokens (or, --scan-only, embeddings) of query batch b into buffer set i
        if args.scan_only:
            eng.ctxs[i].q_out.copy_(q32_all[b, enc_lo:enc_lo + q_loc])
        else:
            eng.set_tokens(i, ids_dev_all[b, enc_lo:enc_lo + q_loc], lens_dev[:q_loc])

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: n_ctx DISTINCT query batches in flight, one step = one batch from every buffer set ----------
    for i in range(n_ctx):
        load_batch(i, i)
    eng.warm_up()
    sync()
    for _ in range(args.warmup):
        eng.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    sync()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    q_per_step = nq_all * n_ctx
    ms_step = dt / args.steps * 1e3
    qps = q_per_step * args.steps / dt
    timed_status = torch.stack([eng.outputs(i)[2] for i in range(n_ctx)]).clone()     # certificate outcome of the in-flight batches

    # ---- correctness of the timed path (untimed): EVERY query of the set through the same engine (same graphs, fresh token
    # ids per batch), final lists against the exact ranking of the UNQUANTISED fp32 rows of every shard (fp64 accumulation)
    # -- catches quantisation loss as well as any mix-up of query order, id bases, wire layout or merge.
    def run_all(engine):
        fs, fi, fst, fq = [], [], [], []
        for b0 in range(0, n_batches, n_ctx):
            nb = min(n_ctx, n_batches - b0)
            for i in range(nb):
                if args.scan_only:
                    engine.ctxs[i].q_out.copy_(q32_all[b0 + i, enc_lo:enc_lo + q_loc])
                else:
                    engine.set_tokens(i, ids_dev_all[b0 + i, enc_lo:enc_lo + q_loc], lens_dev[:q_loc])
            for i in range(nb):      # (after ALL token blocks are in place: one encoder forward may serve several batches)
                engine.submit(i)
            torch.cuda.synchronize()
            for i in range(nb):
                s_, i_, st_ = engine.outputs(i)
                c_ = engine.ctxs[i]
                fs.append(s_.clone()); fi.append(i_.clone()); fst.append(st_.clone())
                fq.append((c_.q_all32 if engine.gather_q else c_.q_out).clone())     # the fp32 queries this batch searched with
        return torch.cat(fs), torch.cat(fi), torch.cat(fst), torch.cat(fq)

    # ground-truth queries = the embeddings the engine itself produced and searched with (its encoder runs the small-LDS
    # kernel forms under role lanes: same arithmetic, different K-chunking, so the last bits differ from a plain forward)
    fin_s, fin_i, fin_st, q_truth = run_all(eng)
    nq_total = q_truth.shape[0]
    gt_s = torch.empty((nq_total, k), dtype=torch.float64, device=dev)
    gt_i = torch.empty((nq_total, k), dtype=torch.int64, device=dev)
    for lo in range(0, nq_total, 1024):
        gt_s[lo:lo + 1024], gt_i[lo:lo + 1024] = exact_topk_f64(torch, q_truth[lo:lo + 1024], shadow, rows, k, id_base)
    if multi:
        all_s = torch.empty((world * nq_total, k), dtype=torch.float64, device=dev)
        all_i = torch.empty((world * nq_total, k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_s, gt_s.contiguous()); dist.all_gather_into_tensor(all_i, gt_i.contiguous())
        all_s = all_s.view(world, nq_total, k).permute(1, 0, 2).reshape(nq_total, world * k)
        all_i = all_i.view(world, nq_total, k).permute(1, 0, 2).reshape(nq_total, world * k)   # rank-major = id-ascending
        o = torch.argsort(all_s, dim=1, descending=True, stable=True)[:, :k]
        gt_s, gt_i = torch.gather(all_s, 1, o), torch.gather(all_i, 1, o)
    rec_timed = recall_rows(fin_i, gt_i)
    recall_timed = float(rec_timed.mean().item())
    # "identical id sets" can only be asked up to what fp32 resolves: the product (like the reference's fp32 store) ranks by
    # fp32 dot products, this ground truth accumulates in fp64, and two rows within ~1e-7 of each other around rank k may
    # swap.  A returned list is accepted as exact when every returned row (that this rank owns) scores, in fp64, at least
    # the true k-th score minus 3e-7; anything else is a real miss.
    own = (fin_i >= id_base) & (fin_i < id_base + rows)
    loc = (fin_i - id_base).clamp(0, rows - 1)
    s64 = torch.empty(fin_i.shape, dtype=torch.float64, device=dev)
    for lo in range(0, nq_total, 2048):
        s64[lo:lo + 2048] = (shadow[loc[lo:lo + 2048]].double() * q_truth[lo:lo + 2048].double().unsqueeze(1)).sum(-1)
    kth64 = gt_s[:, k - 1:k]
    exact_ok = ((~own) | (fin_i < 0) | (s64 >= kth64 - 3e-7)).all(dim=1) & ((fin_i >= 0).sum(1) == (gt_i >= 0).sum(1))
    if multi:
        eo = exact_ok.to(torch.int32)
        dist.all_reduce(eo, op=dist.ReduceOp.MIN)
        exact_ok = eo.bool()
    exact_frac = float(exact_ok.float().mean().item())
    ids_identical = float((fin_i == gt_i).all(dim=1).float().mean().item())
    sets_identical = float((rec_timed == 1.0).float().mean().item())
    score_err = float((fin_s.double() - gt_s).abs().max().item())
    # certificate outcome over the whole set (this rank's shard; N > 1: summed over the ranks)
    cert_counts = torch.tensor([(fin_st == 0).sum(), (fin_st == 1).sum(), (fin_st == 2).sum(), fin_st.numel()], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(cert_counts)
    cert_counts = cert_counts.tolist()
    # the other mode, for the record (untimed, same queries): the plain fp16 / int8 scan with k' = k, or the refined one
    eng_other = make_engine(not refine, exact_mode)
    eng_other.use_graph = False
    eng_other.warm_up()
    oth_s, oth_i, _, _ = run_all(eng_other)
    torch.cuda.synchronize()
    recall_other = float(recall_rows(oth_i, gt_i).mean().item())
    err_other = float((oth_s.double() - gt_s).abs().max().item())
    del eng_other
    tol = 1e-5 if refine else (5e-3 if slab_type == nat.SLAB_I8 else 1e-3)
    # with escalation on (fp16), every list is proven or made exact: demand identical id SETS for every query, not a mean
    guaranteed = refine and eng.exact
    check_ok = bool(score_err < tol and recall_timed >= (0.999 if refine else 0.8) and (not guaranteed or exact_frac == 1.0))
    recall_report = {"timed_path": round(recall_timed, 6), "queries_checked": int(nq_total),
                     "timed_mode": (f"{slab_kind} scan k'={k_scan} + fp32 shadow re-rank + exactness certificate"
                                    + (" + in-stream escalation of unproven queries" if eng.exact else " (unproven queries NOT escalated: empirical)")
                                    if refine else f"{slab_kind} scan only"),
                     "queries_with_identical_id_sets": round(sets_identical, 6),
                     "queries_exact_up_to_fp32_resolution": round(exact_frac, 6),
                     "queries_with_identical_ordered_ids": round(ids_identical, 6), "max_abs_score_err_vs_fp64": score_err,
                     "certified_frac": round(cert_counts[0] / max(cert_counts[3], 1), 6) if refine else None,
                     "escalated": int(cert_counts[1]) if (refine and eng.exact) else 0,
                     "unproven": (int(cert_counts[2]) if eng.exact else int(cert_counts[1] + cert_counts[2])) if refine else None,
                     "escalated_in_timed_batches": int((timed_status == 1).sum().item()) if (refine and eng.exact) else 0,
                     ("scan_only_no_refine" if refine else "with_fp32_refine"): round(recall_other, 6),
                     ("scan_only_max_abs_score_err" if refine else "with_fp32_refine_max_abs_score_err"): err_other}

    # ---- roofline of the dominant kernel (the scan), hipEvent-timed on the launch stream; the re-rank beside it
    c0 = eng.ctxs[0]
    qa16 = c0.q_all16 if gather_q else c0.q16
    ms_total, ms_scan = nat.time_cosine_topk(qa16, slab, rows, dim, k_scan, max(10, min(args.steps, 50)),
                                             slab_type=slab_type, scales=scales)
    ms_refine = None
    if refine:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            # (the segment's tail as the timed path runs it: certificate + the escalation launches)
            qa32 = c0.q_all32 if gather_q else c0.q_out
            nat.refine_f32_cert(qa32, qa16, shadow, rows, id_base, c0.cand_i, c0.cand_s, k, view.row_err_max, slab_type, c0.exact_ws,
                                eng.exact_cap, out_scores=c0.wire.scores, out_ids=c0.wire.ids, status=c0.status)
            if eng.exact:
                nat.escalate_exact(qa32, qa16, slab, shadow, rows, id_base, k, c0.wire.scores, c0.wire.ids, c0.status, c0.exact_ws,
                                   eng.exact_cap, scales=scales)
        e1.record(); e1.synchronize()
        ms_refine = e0.elapsed_time(e1) / 50
    # in-run figure: the search segment (scan + merge + tile refine + certificate + escalation launches) as it runs in the timed
    # mix, event-timed on the search lane -- what a rocprof trace of this command averages to, beside the isolated kernel_ms
    ms_search_in_run = None
    if not multi:
        for i in range(n_ctx):
            load_batch(i, i)
        ms_search_in_run = eng.measure_search_segment_ms()
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
                "kernel_ms": round(ms_scan, 5), "kernel_ms_is": "scan kernel alone, back-to-back launches, hipEvents (crs_time_cosine_topk)",
                "scan_merge_refine_ms": round(ms_total, 5),
                "search_segment_ms_in_run": round(ms_search_in_run, 5) if ms_search_in_run is not None else None,
                "fp32_rerank_cert_ms": round(ms_refine, 5) if ms_refine is not None else None,
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
        q_h = q32_all[0].cpu().numpy()
        mask_h = np.ones((qb, QUERY_TOKENS), dtype=np.int32)
        # the HIP path on the same sample prefix against the oracle (exactness of the kernels themselves)
        q16s = nat.queries_to_f16(q32_all[0], slab_type)
        cs_, ci_ = nat.cosine_topk(q16s, slab, sample_rows, dim, k_scan, slab_type=slab_type, scales=scales)
        gs_, gi_ = nat.refine_f32(q32_all[0], shadow, sample_rows, 0, ci_, k)
        rs, ri = scan_ref.cosine_topk_ref(q_h, rows_h, k)
        gi_h = gi_.cpu().numpy()
        recall_oracle = float(np.mean([scan_ref.recall_at_k(gi_h[r], ri[r]) for r in range(qb)]))
        max_err = float(np.abs(gs_.cpu().numpy() - rs).max())
        n_done, t_enc, tm = 0, 0.0, {}
        while (t_enc + tm.get("gemm", 0.0) + tm.get("select", 0.0)) < args.cpu_seconds:
            ta = time.perf_counter()
            er.encode_ref(ids_all[0], mask_h, enc_w, cfg)
            t_enc += time.perf_counter() - ta
            scan_ref.cosine_topk_ref(q_h, rows_h, k, timing=tm)
            n_done += 1
        t_gemm, t_sel = tm["gemm"], tm["select"]
        t_cpu = t_enc + t_gemm + t_sel
        # queries/s over the FULL corpus: the scan part of the sample's time scales by rows/sample_rows
        cpu_qps = qb * n_done / (t_enc + (t_gemm + t_sel) * (rows / sample_rows))
        gemm_gflops = 2.0 * qb * sample_rows * dim * n_done / max(t_gemm, 1e-9) / 1e9
        cpu = {"value": round(cpu_qps, 2), "unit": "queries/s", "cores": int(torch.get_num_threads()),
               "cpu_model": cpu_model(), "kind": "port",
               "sample": f"oracle encoder_ref.encode_ref ({qb}x{QUERY_TOKENS} tokens, torch fp32) + scan_ref.cosine_topk_ref "
                         f"(numpy fp32 sgemm per 65536-row block + partition-based exact top-k) on {sample_rows} of {rows} fp32 rows; "
                         f"{n_done} passes in {t_cpu:.1f}s, scan time scaled by rows/sample",
               "seconds": {"encoder": round(t_enc, 3), "sgemm": round(t_gemm, 3), "selection": round(t_sel, 3)},
               "sgemm_gflops": round(gemm_gflops, 1),
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
                       "distinct_queries_in_flight": q_per_step,
                       "ms_per_batch": round(ms_step / n_ctx, 5), "lanes": eng.describe_lanes(), "batches_per_encoder_forward": eng.enc_group, "searches_per_graph": eng.search_fuse, "top_k": k, "k_scan": k_scan,
                       "slab": slab_kind, "refine_fp32": refine, "exact_escalation": bool(eng.exact), "encoder_in_step": not args.scan_only,
                       "engine": "rag._engine.RetrievalEngine (the object ContextRetriever.retrieve_batch drives)",
                       "encoder": ("all-MiniLM-L6-v2" if enc_name == "minilm" else "bge-base-en-v1.5") + " shape, seeded random weights",
                       "query_tokens": QUERY_TOKENS, "hip_graph": eng.use_graph,
                       "collectives_per_batch": eng.collectives_per_batch, "dist_single_rank": bool(multi and world == 1),
                       "query_encode": ("replicated" if not gather_q else ("per-rank queries (weak scaling)" if not strong else
                                        f"sharded: {q_loc} of {qb} queries per rank" + (" [single-GPU proxy: tiled instead of gathered]" if not multi else ""))),
                       "recall_at_10_vs_fp32": recall_report, "check_ok": check_ok,
                       "row_err_max_tracked": view.row_err_max,
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
