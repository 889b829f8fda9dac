"""Row sharding of the slab + the one exchange step of a sharded search.

The reference is single-process; this is the new multi-GPU layer of section 8(e):
contiguous row shards, each shard scanned where it lives, ONE all-gather of the per-shard results and a k-way
merge on every rank.  Two drivers share it:
  * SPMD (one process per GPU, torch.distributed over RCCL): ``VectorStore({"sharded": True})``, bench.py;
  * one process driving N devices (so RAGPipeline stays one object): ``VectorStore({"num_gpus": N})``.
Ids on the wire are GLOBAL sidecar rows (the row of the ids / documents / metadatas lists, identical on every
rank), so the merged order (score desc, row asc) is the single-GPU order whatever the sharding.
The wire block layout is the C ABI's (include/crs_hip.h: [ids int64 [nq, k] | scores fp32 [nq, k] | pad to 8]).
"""
from __future__ import annotations

from typing import Callable, List, Tuple


def shard_slice(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of an n_items batch that shard `rank` keeps (contiguous, ceil-sized shards)."""
    per = -(-n_items // world) if n_items else 0
    return min(rank * per, n_items), min((rank + 1) * per, n_items)


def batch_slices(n_items: int, world: int) -> List[Tuple[int, int]]:
    return [shard_slice(n_items, world, r) for r in range(world)]


def wire_layout(nq: int, k: int) -> Tuple[int, int]:
    """(block bytes, byte offset of the scores) -- the Python statement of crs_wire_bytes / crs_wire_scores_offset
    (tests hold the two to equality)."""
    ids_bytes = nq * k * 8
    return ids_bytes + (nq * k * 4 + 7) // 8 * 8, ids_bytes


def pack_wire(scores, ids):
    """scores fp32 [nq, k], ids int64 [nq, k] (torch, any device) -> uint8 wire block (a copy; the HIP path writes the
    block in place through rag._native.WireBlock instead)."""
    import torch
    nq, k = scores.shape
    nbytes, off = wire_layout(nq, k)
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=scores.device)
    buf[:off].view(torch.int64).copy_(ids.reshape(-1))
    buf[off:off + nq * k * 4].view(torch.float32).copy_(scores.reshape(-1))
    return buf


def unpack_wire(gathered, nlists: int, nq: int, k: int):
    """uint8 [nlists * block] -> (scores fp32 [nlists, nq, k], ids int64 [nlists, nq, k]) (copies)."""
    import torch
    nbytes, off = wire_layout(nq, k)
    blocks = gathered.view(nlists, nbytes)
    ids = torch.stack([blocks[g, :off].clone().view(torch.int64).view(nq, k) for g in range(nlists)])
    scores = torch.stack([blocks[g, off:off + nq * k * 4].clone().view(torch.float32).view(nq, k) for g in range(nlists)])
    return scores, ids


def allgather_merge(dist, block_buf, gathered, nq: int, k_in: int, k_out: int, merge_fn: Callable):
    """THE exchange of a sharded search: one all-gather of this rank's wire block, then the k-way merge.
    block_buf uint8 [block bytes]; gathered uint8 [world * block bytes] (reusable scratch);
    merge_fn(gathered, nlists, nq, k_in, k_out) -> (scores [nq, k_out], ids [nq, k_out])."""
    dist.all_gather_into_tensor(gathered, block_buf)
    return merge_fn(gathered, dist.get_world_size(), nq, k_in, k_out)
