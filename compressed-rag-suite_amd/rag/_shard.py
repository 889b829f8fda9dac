"""Row sharding of the slab across ranks + the one exchange step of a sharded search.

The reference is single-process; this is the new multi-GPU layer of
section 8(e): contiguous row shards per rank, each rank scans its own shard, ONE all-gather of the
per-shard (score, id) lists (KiB-sized, latency-bound) and a k-way merge on every rank.  The
collective and the merge are injected so the same code runs under RCCL with the HIP merge kernel
(product) and under gloo with a CPU merge (tests/test_shard_cpu.py).
"""
from __future__ import annotations

from typing import Callable, List, Tuple

RANK_SHIFT = 40          # ids on the wire = (rank << 40) | local_row  -> order = (rank, local row)
LOCAL_MASK = (1 << RANK_SHIFT) - 1


def shard_slice(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of an n_items batch that `rank` keeps (contiguous, ceil-sized shards)."""
    per = -(-n_items // world) if n_items else 0
    return min(rank * per, n_items), min((rank + 1) * per, n_items)


class ShardMap:
    """Maps (rank, local slab row) back to the row of the host sidecars (ids / documents / metadata),
    across any number of sharded add() batches."""

    def __init__(self, world: int):
        self.world = world
        self.batches: List[Tuple[int, int]] = []   # (sidecar start row, batch rows)

    def add_batch(self, start_row: int, n_items: int) -> None:
        self.batches.append((start_row, n_items))

    def global_row(self, rank: int, local_row: int) -> int:
        for start, n in self.batches:
            lo, hi = shard_slice(n, self.world, rank)
            if local_row < hi - lo:
                return start + lo + local_row
            local_row -= hi - lo
        raise IndexError((rank, local_row))

    def clear(self) -> None:
        self.batches.clear()


def tag(ids, rank: int):
    """local rows (torch int64, -1 = empty) -> wire ids carrying the rank in the high bits."""
    import torch
    return torch.where(ids >= 0, ids + (rank << RANK_SHIFT), ids)


def untag(wire_id: int) -> Tuple[int, int]:
    return wire_id >> RANK_SHIFT, wire_id & LOCAL_MASK


def allgather_merge(dist, scores, wire_ids, k: int, merge_fn: Callable):
    """scores fp32 [nq, k], wire_ids int64 [nq, k] of THIS rank -> merged top-k over all ranks.
    merge_fn(gathered_scores [W, nq, k], gathered_ids [W, nq, k], k) -> (scores, ids)."""
    import torch
    w = dist.get_world_size()
    nq, kk = scores.shape
    # flat [W * nq, k] output (concatenation along dim 0) is the form every backend accepts
    gs = torch.empty((w * nq, kk), dtype=scores.dtype, device=scores.device)
    gi = torch.empty((w * nq, kk), dtype=wire_ids.dtype, device=wire_ids.device)
    dist.all_gather_into_tensor(gs, scores.contiguous())
    dist.all_gather_into_tensor(gi, wire_ids.contiguous())
    return merge_fn(gs.view(w, nq, kk), gi.view(w, nq, kk), k)
