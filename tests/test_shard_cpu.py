"""CPU, world_size 2 over gloo: the sharded-search exchange (contiguous row shards, ids on the wire = global sidecar
rows, ONE all-gather of the wire blocks, k-way merge) against the single-shard oracle.  The local scan and the merge
are the CPU oracle here; on the GPU the same _shard.allgather_merge runs with crs::cosine_topk / crs::merge_topk_wire
(tests/test_sharded_store_gpu.py).  Also: the Python statement of the wire layout equals the C ABI's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import scan_ref


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, nq, k, batches, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "compressed-rag-suite_amd"))
    from rag import _shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    corpus = scan_ref.synth_corpus(n, d, seed=3).astype(np.float16)
    q = scan_ref.synth_queries(corpus.astype(np.float32), nq, seed=4).astype(np.float16)
    corpus[n // 2 + 1] = corpus[5]                     # an exact tie that straddles the two shards
    rows_global, start = [], 0
    for b in batches:                                   # several add() batches, each sharded contiguously
        lo, hi = _shard.shard_slice(b, world, rank)
        rows_global += list(range(start + lo, start + hi))
        start += b
    rows_global = np.asarray(rows_global, dtype=np.int64)
    s, i = scan_ref.cosine_topk_ref(q, corpus[rows_global], k, accumulate=np.float64)
    gi = np.where(i >= 0, rows_global[np.clip(i, 0, None)], -1)       # local row -> global sidecar row

    def merge(gathered, nlists, nq_, k_in, k_out):
        gs, gids = _shard.unpack_wire(gathered, nlists, nq_, k_in)
        ms, mi = scan_ref.merge_topk_ref(gs.numpy(), gids.numpy(), k_out)
        return torch.from_numpy(ms), torch.from_numpy(mi)

    buf = _shard.pack_wire(torch.from_numpy(s), torch.from_numpy(gi))
    gathered = torch.zeros(world * buf.numel(), dtype=torch.uint8)
    ms, mi = _shard.allgather_merge(dist, buf, gathered, nq, k, k, merge)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), scores=ms.numpy(), rows=mi.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batches", [[1000], [300, 501, 199]])
def test_sharded_search_equals_single_shard_oracle(tmp_path, batches):
    n, d, nq, k = sum(batches), 64, 5, 7
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, d, nq, k, batches, str(tmp_path)), nprocs=2, join=True)
    corpus = scan_ref.synth_corpus(n, d, seed=3).astype(np.float16)
    q = scan_ref.synth_queries(corpus.astype(np.float32), nq, seed=4).astype(np.float16)
    corpus[n // 2 + 1] = corpus[5]
    rs, ri = scan_ref.cosine_topk_ref(q, corpus, k, accumulate=np.float64)
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["scores"], rs)
        assert np.array_equal(z["rows"], ri)          # global-row wire ids: the single-shard order, ties included, for ANY batching


def test_shard_slices_cover_everything():
    from rag import _shard
    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            parts = _shard.batch_slices(n, w)
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
    assert _shard.batch_slices(10_000_000, 8)[3] == (3_750_000, 5_000_000)


def test_wire_layout_matches_the_c_abi():
    from rag import _native as nat, _shard
    lib = nat.load()
    for nq, k in ((1, 1), (3, 5), (64, 10), (512, 10), (7, 3), (255, 64)):
        assert _shard.wire_layout(nq, k) == (lib.crs_wire_bytes(nq, k), lib.crs_wire_scores_offset(nq, k))
        assert lib.crs_wire_bytes(nq, k) % 8 == 0
    s = torch.arange(15, dtype=torch.float32).view(3, 5)
    i = torch.arange(15, dtype=torch.int64).view(3, 5) * 1000
    buf = _shard.pack_wire(s, i)
    gs, gi = _shard.unpack_wire(torch.cat([buf, buf]), 2, 3, 5)
    assert torch.equal(gs[1], s) and torch.equal(gi[0], i)
