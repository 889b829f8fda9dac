#!/usr/bin/env python3
"""bench.py -- queries/sec of the embed -> index -> retrieve hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N > 1: launched by
torch.distributed.run, one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

A "step" is one query batch through the retrieve path with everything resident in HBM:
  query encoder forward (token ids [Qb, 16] -> fp32 sentence embeddings; MFMA GEMMs + attention)
  -> normalise + fp16 cast -> exact cosine scan of this rank's slab shard + in-kernel top-k
  -> per-workgroup list merge -> (N > 1) RCCL all-gather of the per-shard top-k + final merge.
The encoder has the architecture BASELINE.json names for the workload (all-MiniLM-L6-v2 for the
384-d configs, bge-base-en-v1.5 for the 768-d ones) with seeded random weights and synthetic token
ids (no checkpoints offline).

Workloads (BASELINE.json configs; --workload):
  c2  100k x 384 fp16 slab per GPU, 64 queries per rank per step, k=10   (default; configs[1])
  c3  1M x 768 fp16, 256 queries
  c4  10M x 384 fp16 sharded: 1.25M rows per GPU (the 8-GPU shard size), 64 queries per rank
  c5  as c4 with an int8 768-d slab
Scaling is weak: the per-GPU shard and the per-rank query batch are fixed; the corpus and the
global query batch grow with N (every rank scans its shard for all N*Qb queries of the step).
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
    #        rows/GPU   dim  Qb   k  slab  encoder
    "c2": (100_000, 384, 64, 10, "f16", "minilm"),
    "c3": (1_000_000, 768, 256, 10, "f16", "bge"),
    "c4": (1_250_000, 384, 64, 10, "f16", "minilm"),
    "c5": (1_250_000, 768, 64, 10, "i8", "bge"),
}
QUERY_TOKENS = 16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def synth_shard(torch, n, dim, seed, device, chunk=250_000):
    """Seeded unit-normalised Gaussian rows, generated on the device in chunks (fp32)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        x = torch.randn((m, dim), generator=g, device=device, dtype=torch.float32)
        yield lo, x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--scan-only", action="store_true", help="diagnostic: skip the encoder (NOT the metric)")
    ap.add_argument("--streams", type=int, default=32,
                    help="independent query batches in flight on separate HIP streams (1 GPU runs only)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--extra-launches", type=int, default=0,
                    help="diagnostic: this many extra tiny kernels per step (what does one more launch cost?)")
    ap.add_argument("--queries", type=int, default=0, help="diagnostic: override the per-rank query batch size")
    args = ap.parse_args()

    import numpy as np
    import torch
    from rag import _native as nat

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
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CRS_DIST_BACKEND", "nccl")   # "gloo": functional rehearsal of N > 1 on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    rows, dim, qb, k, slab_kind, enc_name = WORKLOADS[args.workload]
    if args.queries > 0:
        qb = args.queries
    slab_type = nat.SLAB_I8 if slab_kind == "i8" else nat.SLAB_F16
    pd = nat.padded_dim(dim, slab_type)
    id_base = rank * rows

    # ---- index build (untimed): synthetic embeddings -> slab shard in HBM through the product path
    slab = torch.empty((rows, pd), dtype=torch.int8 if slab_type == nat.SLAB_I8 else torch.float16, device=dev)
    scales = torch.empty(rows, dtype=torch.float32, device=dev) if slab_type == nat.SLAB_I8 else None
    t_build = time.perf_counter()
    for lo, x in synth_shard(torch, rows, dim, 1234 + rank, dev):
        nat.slab_append_f32(x, slab, lo, slab_type, scales=scales)
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
    rng = np.random.default_rng(4321 + rank)
    ids_h = rng.integers(1000, shape.vocab_size, size=(qb, QUERY_TOKENS)).astype(np.int32)
    ids_h[:, 0], ids_h[:, -1] = 101, 102                      # [CLS] ... [SEP], no padding
    mask_h = np.ones((qb, QUERY_TOKENS), dtype=np.int32)
    ids_d = torch.from_numpy(ids_h).to(dev)
    lens_d = torch.from_numpy(mask_h.sum(1).astype(np.int32)).to(dev)
    q32 = enc.forward(ids_d, lens_d).clone()
    # plant a near neighbour of every even query into this rank's shard (50 % planted)
    g = torch.Generator(device=dev); g.manual_seed(99 + rank)
    j = torch.randperm(rows, generator=g, device=dev)[: qb // 2]
    planted = q32[0::2] + 0.1 * torch.randn((qb // 2, dim), generator=g, device=dev)
    tmp = torch.empty((qb // 2, pd), dtype=slab.dtype, device=dev)
    tmp_sc = torch.empty(qb // 2, dtype=torch.float32, device=dev) if scales is not None else None
    nat.slab_append_f32(planted.contiguous(), tmp, 0, slab_type, scales=tmp_sc)
    slab[j] = tmp
    if scales is not None:
        scales[j] = tmp_sc
    nq_all = qb * world

    class Ctx:
        """Buffers of one in-flight query batch (a step touches nothing outside its Ctx + read-only state)."""
        def __init__(self):
            self.q_out = torch.empty((qb, dim), dtype=torch.float32, device=dev)
            self.q16 = torch.empty((qb, pd), dtype=torch.float16, device=dev)
            self.dummy16 = torch.empty((qb, pd), dtype=torch.float16, device=dev)
            self.enc_ws = torch.empty(enc.workspace_bytes(qb, QUERY_TOKENS), dtype=torch.uint8, device=dev)
            self.ws = torch.empty(nat.scan_workspace_bytes(nq_all, dim, k, rows), dtype=torch.uint8, device=dev)
            self.out_s = torch.empty((nq_all, k), dtype=torch.float32, device=dev)
            self.out_i = torch.empty((nq_all, k), dtype=torch.int64, device=dev)
            self.graphs = None
            if world > 1:
                self.q_all = torch.empty((nq_all, pd), dtype=torch.float16, device=dev)
                self.gs = torch.empty((world * nq_all, k), dtype=torch.float32, device=dev)
                self.gi = torch.empty((world * nq_all, k), dtype=torch.int64, device=dev)
                self.fin_s = torch.empty((nq_all, k), dtype=torch.float32, device=dev)
                self.fin_i = torch.empty((nq_all, k), dtype=torch.int64, device=dev)

    # The step is three device segments with the two exchanges between them; every segment reads and
    # writes fixed buffers of its Ctx, so each can be captured once into a hipGraph and replayed.
    def seg_encode(c):      # token ids -> fp16 queries of this rank
        if args.scan_only:
            nat.queries_to_f16(q32, slab_type, out=c.q16)
        else:   # pooled embeddings leave the encoder as fp32 and as the scan's fp16 query block
            enc.forward(ids_d, lens_d, out=c.q_out, workspace=c.enc_ws, q16_out=c.q16, slab_type=slab_type)
        for _ in range(args.extra_launches):
            nat.queries_to_f16(q32, slab_type, out=c.dummy16)

    def seg_scan(c):        # all queries of the step x this rank's shard -> per-shard top-k
        nat.cosine_topk(c.q_all if world > 1 else c.q16, slab, rows, dim, k, slab_type=slab_type, scales=scales,
                        id_base=id_base, workspace=c.ws, out_scores=c.out_s, out_ids=c.out_i)

    def seg_merge(c):       # N > 1: the gathered per-shard lists -> global top-k
        nat.merge_topk(c.gs.view(world, nq_all, k), c.gi.view(world, nq_all, k), k, out_scores=c.fin_s, out_ids=c.fin_i)

    segs = (seg_encode, seg_scan) + ((seg_merge,) if world > 1 else ())

    def step(c):
        run_seg = (lambda j: c.graphs[j].replay()) if c.graphs is not None else (lambda j: segs[j](c))
        run_seg(0)
        if world > 1:
            dist.all_gather_into_tensor(c.q_all, c.q16)
        run_seg(1)
        if world > 1:
            dist.all_gather_into_tensor(c.gs, c.out_s)
            dist.all_gather_into_tensor(c.gi, c.out_i)
            run_seg(2)
            return c.fin_s, c.fin_i
        return c.out_s, c.out_i

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Throughput mode: S independent batches in flight, each on its own stream with its own buffers.  The
    # device segments of a step (~45 small launches) are captured once per batch into hipGraphs and
    # replayed; for N > 1 the RCCL collectives between the segments are launched eagerly (those of
    # different batches are serialised on the process group's own stream; every rank issues them in the
    # same order).
    n_streams = max(1, args.streams)
    use_graph = not args.no_graph
    ctxs = [Ctx() for _ in range(n_streams)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    torch.cuda.synchronize()
    for c, st in zip(ctxs, streams):
        with torch.cuda.stream(st):
            for _ in range(3):
                step(c)
        st.synchronize()
        if use_graph:
            # thread_local: with N > 1 the process group's watchdog thread polls events while we capture;
            # only this thread's calls belong to the capture.  If a capture fails anyway, run eagerly.
            try:
                gl = []
                for seg in segs:
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

    def run(n):
        for it in range(n):
            sidx = it % n_streams
            with torch.cuda.stream(streams[sidx]):
                step(ctxs[sidx])

    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    dt = time.perf_counter() - t0
    res_i = ctxs[0].out_i
    # functional check of the whole exchange (untimed): the final lists against a plain torch matmul + topk
    # over every rank's shard, gathered and re-sorted -- catches any mix-up of query order, id bases,
    # all-gather layout or merge
    with torch.cuda.stream(streams[0]):
        fin_s, fin_i = step(ctxs[0])
    streams[0].synchronize()
    q_chk = (ctxs[0].q16 if world == 1 else ctxs[0].q_all).float()[:, :dim]
    ref_s = torch.full((nq_all, k), float("-inf"), device=dev)
    ref_i = torch.full((nq_all, k), -1, dtype=torch.int64, device=dev)
    for lo in range(0, rows, 250_000):
        blk = slab[lo:lo + 250_000, :dim].float()
        if scales is not None:
            blk = blk * scales[lo:lo + 250_000, None]
        sc_blk = q_chk @ blk.T
        ts, ti = sc_blk.topk(min(k, sc_blk.shape[1]), dim=1)
        cat_s, cat_i = torch.cat([ref_s, ts], 1), torch.cat([ref_i, ti + lo + id_base], 1)
        ref_s, pos = cat_s.topk(k, dim=1)
        ref_i = torch.gather(cat_i, 1, pos)
    if world > 1:
        all_s = torch.empty((world * nq_all, k), device=dev); all_i = torch.empty((world * nq_all, k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_s, ref_s.contiguous()); dist.all_gather_into_tensor(all_i, ref_i.contiguous())
        all_s = all_s.view(world, nq_all, k).permute(1, 0, 2).reshape(nq_all, world * k)
        all_i = all_i.view(world, nq_all, k).permute(1, 0, 2).reshape(nq_all, world * k)
        ref_s, pos = all_s.topk(k, dim=1)
        ref_i = torch.gather(all_i, 1, pos)
    tol = 2e-3 if slab_type == nat.SLAB_I8 else 2e-5     # int8: the kernel searches with 16-bit fixed-point queries
    score_err = float((fin_s - ref_s).abs().max().item())
    id_match = float((fin_i == ref_i).float().mean().item())
    exchange_ok = bool(score_err < tol and id_match > 0.98)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = dt / args.steps * 1e3
    qps = nq_all * args.steps / dt

    # ---- roofline of the dominant kernel (the scan), hipEvent-timed on the launch stream
    q16_all = nat.queries_to_f16(q32, slab_type) if world == 1 else ctxs[0].q_all
    ms_total, ms_scan = nat.time_cosine_topk(q16_all, slab, rows, dim, k, max(20, min(args.steps, 200)),
                                             slab_type=slab_type, scales=scales)
    elem = 1 if slab_type == nat.SLAB_I8 else 2
    alg_bytes = rows * pd * elem + (rows * 4 if slab_type == nat.SLAB_I8 else 0) + nq_all * pd * 2 + nq_all * k * 8
    achieved = alg_bytes / (ms_scan * 1e-3) / 1e9
    # HBM traffic per launch comes from the committed PMC passes of this same command (bench.py cannot
    # read hardware counters itself); null when that workload has not been profiled yet
    traffic, traffic_src = None, None
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
            with open(path) as fh:
                pm = json.load(fh)
            if args.workload in pm and world == 1:
                traffic = pm[args.workload]["hbm_read_bytes_per_launch"]
                traffic_src = os.path.relpath(path, ROOT)
                break
    except Exception:
        pass
    # Which roof bounds the launch: flop per algorithmic byte (= queries per launch for fp16, 2x that for int8)
    # against the ridge of the dense MFMA peak over the HBM peak (MI355X_MICROARCH.md: 2.5 PF fp16 / 5 PF int8, 8 TB/s).
    # One rank of an N-GPU step scans its shard for the queries of ALL ranks, so from ~313 queries on the scan is
    # matrix-bound and is priced against the MFMA peak; both fractions are reported either way.
    alg_flops = 2.0 * nq_all * rows * pd
    mfma_peak_tf = 5000.0 if slab_type == nat.SLAB_I8 else 2500.0
    achieved_tf = alg_flops / (ms_scan * 1e-3) / 1e12
    hbm_frac, mfma_frac = achieved / HBM_PEAK_GBS, achieved_tf / mfma_peak_tf
    mfma_bound = (alg_flops / alg_bytes) > (mfma_peak_tf * 1e12) / (HBM_PEAK_GBS * 1e9)
    roofline = {"bound": "mfma" if mfma_bound else "hbm",
                "achieved": round(achieved_tf if mfma_bound else achieved, 1),
                "peak": mfma_peak_tf if mfma_bound else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": round(mfma_frac if mfma_bound else hbm_frac, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": nat.scan_plan_describe(nq_all, dim, k, rows, slab_type),
                "kernel_ms": round(ms_scan, 5), "scan_plus_merge_ms": round(ms_total, 5),
                "algorithmic_bytes": int(alg_bytes), "algorithmic_flops": int(alg_flops),
                "hbm_frac": round(hbm_frac, 4), "mfma_frac": round(mfma_frac, 4)}

    # ---- correctness of the timed result + CPU baseline (rank 0, N=1 only): the oracle, same workload
    cpu = None
    recall = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import encoder_ref as er, scan_ref   # the CPU restatement: only ever the baseline / checker
        cfg = er.EncoderConfig(shape.vocab_size, shape.hidden, shape.layers, shape.heads, shape.ffn, shape.max_pos, 2,
                               shape.ln_eps, shape.max_seq, shape.pooling)
        sample_rows = min(rows, 200_000)
        slab_h = slab[:sample_rows, :dim].cpu().numpy()
        sc_h = scales[:sample_rows].cpu().numpy() if scales is not None else None
        q_h = nat.queries_to_f16(q32, slab_type)[:, :dim].cpu().numpy()
        # check the GPU's answer on the sample prefix against the oracle (exactness of the timed path)
        gs_, gi_ = nat.cosine_topk(nat.queries_to_f16(q32, slab_type), slab, sample_rows, dim, k, slab_type=slab_type,
                                   scales=scales)
        rs, ri = scan_ref.cosine_topk_ref(q_h, slab_h, k, scales=sc_h)
        gi_h = gi_.cpu().numpy()
        recall = float(np.mean([scan_ref.recall_at_k(gi_h[r], ri[r]) for r in range(qb)]))
        max_err = float(np.abs(gs_.cpu().numpy() - rs).max())
        # time the oracle on the bounded sample: encoder (fp32 torch CPU restatement) + exact scan
        n_done, t_enc, t_scan = 0, 0.0, 0.0
        t_start = time.perf_counter()
        while (t_enc + t_scan) < args.cpu_seconds:
            ta = time.perf_counter()
            er.encode_ref(ids_h, mask_h, enc_w, cfg)
            tb = time.perf_counter()
            scan_ref.cosine_topk_ref(q_h, slab_h, k, scales=sc_h)
            tc = time.perf_counter()
            t_enc += tb - ta; t_scan += tc - tb
            n_done += 1
        t_cpu = t_enc + t_scan
        # queries/s over the FULL shard: the scan part of the sample's time scales by rows/sample_rows
        cpu_qps = qb * n_done / (t_enc + t_scan * (rows / sample_rows))
        cpu = {"value": round(cpu_qps, 1), "unit": "queries/s", "cores": int(torch.get_num_threads()),
               "kind": "port",
               "sample": f"oracle encoder_ref.encode_ref ({qb}x{QUERY_TOKENS} tokens, torch fp32) + scan_ref.cosine_topk_ref "
                         f"(numpy sgemm + exact top-k) over {sample_rows} of {rows} rows; {n_done} passes in "
                         f"{t_cpu:.1f}s (encoder {t_enc:.1f}s, scan {t_scan:.1f}s), scan time scaled by rows",
               "recall_at_10_gpu_vs_oracle": recall, "max_abs_score_err": max_err}

    if rank == 0:
        line = {
            "metric": "queries/sec over N-vector corpus (exact cosine top-k, Recall@10 vs oracle)",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 x f16 -> f32" if slab_type == nat.SLAB_F16 else "i8(f16) x f16 -> f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "rows_per_gpu": rows, "corpus_rows": rows * world, "dim": dim,
                       "queries_per_rank_per_step": qb, "queries_per_step": nq_all, "top_k": k,
                       "slab": slab_kind, "encoder_in_step": not args.scan_only,
                       "encoder": ("all-MiniLM-L6-v2" if enc_name == "minilm" else "bge-base-en-v1.5") + " shape, seeded random weights",
                       "query_tokens": QUERY_TOKENS, "streams_in_flight": n_streams, "hip_graph": use_graph, "exchange_check": {"ok": exchange_ok, "max_score_err": score_err, "id_match": round(id_match, 4)},
                       "index_build_s_per_gpu": round(t_build, 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
