#!/usr/bin/env python3
"""bench.py -- queries/sec of the embed -> index -> retrieve hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N > 1: launched by
torch.distributed.run, one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

A "step" is one query batch through the retrieve path with everything resident in HBM:
  [query encoder forward (token ids -> fp32 embeddings), once the encoder backend is built]
  -> normalise + fp16 cast -> exact cosine scan of this rank's slab shard + in-kernel top-k
  -> per-workgroup list merge -> (N > 1) RCCL all-gather of the per-shard top-k + final merge.

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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))

WORKLOADS = {
    #        rows/GPU   dim  Qb   k  slab
    "c2": (100_000, 384, 64, 10, "f16"),
    "c3": (1_000_000, 768, 256, 10, "f16"),
    "c4": (1_250_000, 384, 64, 10, "f16"),
    "c5": (1_250_000, 768, 64, 10, "i8"),
}
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    rows, dim, qb, k, slab_kind = WORKLOADS[args.workload]
    slab_type = nat.SLAB_I8 if slab_kind == "i8" else nat.SLAB_F16
    pd = nat.padded_dim(dim)
    id_base = rank * rows

    # ---- index build (untimed): synthetic embeddings -> slab shard in HBM through the product path
    slab = torch.empty((rows, pd), dtype=torch.int8 if slab_type == nat.SLAB_I8 else torch.float16, device=dev)
    scales = torch.empty(rows, dtype=torch.float32, device=dev) if slab_type == nat.SLAB_I8 else None
    t_build = time.perf_counter()
    for lo, x in synth_shard(torch, rows, dim, 1234 + rank, dev):
        nat.slab_append_f32(x, slab, lo, slab_type, scales=scales)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    # ---- queries for this rank: half planted near shard rows, half random (fp32, resident)
    g = torch.Generator(device=dev); g.manual_seed(4321 + rank)
    q32 = torch.randn((qb, dim), generator=g, device=dev, dtype=torch.float32)
    j = torch.randint(0, rows, (qb,), generator=g, device=dev)
    planted = (torch.arange(qb, device=dev) % 2 == 0)
    base = slab[j].float()[:, :dim]
    if slab_type == nat.SLAB_I8:
        base = base * scales[j][:, None]
    q32 = torch.where(planted[:, None], base + 0.1 * q32, q32).contiguous()

    nq_all = qb * world
    ws = torch.empty(nat.scan_workspace_bytes(nq_all, dim, k, rows), dtype=torch.uint8, device=dev)
    out_s = torch.empty((nq_all, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq_all, k), dtype=torch.int64, device=dev)
    if world > 1:
        q_all = torch.empty((nq_all, pd), dtype=torch.float16, device=dev)
        gs = torch.empty((world, nq_all, k), dtype=torch.float32, device=dev)
        gi = torch.empty((world, nq_all, k), dtype=torch.int64, device=dev)

    def step():
        q16 = nat.queries_to_f16(q32)
        if world > 1:
            dist.all_gather_into_tensor(q_all, q16)
            q16 = q_all
        s, i = nat.cosine_topk(q16, slab, rows, dim, k, slab_type=slab_type, scales=scales, id_base=id_base,
                               workspace=ws, out_scores=out_s, out_ids=out_i)
        if world > 1:
            dist.all_gather_into_tensor(gs, s)
            dist.all_gather_into_tensor(gi, i)
            s, i = nat.merge_topk(gs, gi, k)
        return s, i

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res_s, res_i = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = dt / args.steps * 1e3
    qps = nq_all * args.steps / dt

    # ---- roofline of the dominant kernel (the scan), hipEvent-timed on the launch stream
    q16_all = nat.queries_to_f16(q32) if world == 1 else q_all
    ms_total, ms_scan = nat.time_cosine_topk(q16_all, slab, rows, dim, k, max(20, min(args.steps, 200)),
                                             slab_type=slab_type, scales=scales)
    elem = 1 if slab_type == nat.SLAB_I8 else 2
    alg_bytes = rows * pd * elem + (rows * 4 if slab_type == nat.SLAB_I8 else 0) + nq_all * pd * 2 + nq_all * k * 8
    achieved = alg_bytes / (ms_scan * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "scan_f16_kernel" if slab_type == nat.SLAB_F16 else "scan_i8_kernel",
                "kernel_ms": round(ms_scan, 5), "scan_plus_merge_ms": round(ms_total, 5),
                "algorithmic_bytes": int(alg_bytes)}

    # ---- correctness of the timed result + CPU baseline (rank 0, N=1 only): the oracle, same workload
    cpu = None
    recall = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import scan_ref
        sample_rows = min(rows, 200_000)
        slab_h = slab[:sample_rows, :dim].cpu().numpy()
        sc_h = scales[:sample_rows].cpu().numpy() if scales is not None else None
        q_h = nat.queries_to_f16(q32)[:, :dim].cpu().numpy()
        # check the GPU's answer on the sample prefix against the oracle (exactness of the timed path)
        gs_, gi_ = nat.cosine_topk(nat.queries_to_f16(q32), slab, sample_rows, dim, k, slab_type=slab_type,
                                   scales=scales)
        rs, ri = scan_ref.cosine_topk_ref(q_h, slab_h, k, scales=sc_h)
        gi_h = gi_.cpu().numpy()
        recall = float(np.mean([scan_ref.recall_at_k(gi_h[r], ri[r]) for r in range(qb)]))
        max_err = float(np.abs(gs_.cpu().numpy() - rs).max())
        # time the oracle on the bounded sample
        n_done, t_cpu = 0, 0.0
        t_start = time.perf_counter()
        while t_cpu < args.cpu_seconds:
            scan_ref.cosine_topk_ref(q_h, slab_h, k, scales=sc_h)
            n_done += 1
            t_cpu = time.perf_counter() - t_start
        # queries/s over the FULL shard: scale the sample's time by rows/sample_rows
        cpu_qps = (qb * n_done / t_cpu) * (sample_rows / rows)
        cpu = {"value": round(cpu_qps, 1), "unit": "queries/s", "cores": int(torch.get_num_threads()),
               "kind": "port",
               "sample": f"oracle/scan_ref.cosine_topk_ref (numpy sgemm + exact top-k), {qb} queries x "
                         f"{sample_rows} of {rows} rows, {n_done} passes in {t_cpu:.1f}s, scaled by rows",
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
                       "slab": slab_kind, "encoder_in_step": False,
                       "index_build_s_per_gpu": round(t_build, 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
