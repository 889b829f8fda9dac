"""GPU: the two multi-GPU drivers of VectorStore against the single-store oracle (oracle/retrieve_ref.StoreRef).

  * SPMD, ``sharded=True``: 2 ranks (torch.distributed.run, gloo, both on the one card) -- create_index over several
    adds, search, search_batch, where / where_document filters, fp16 / fp16+refine / int8+refine; the exchange is ONE
    all-gather of the wire blocks + crs::merge_topk_wire (tests/_sharded_store_worker.py).
  * ONE process, N devices, ``devices=[...]``: two shards placed on the same card here; same checks plus
    persistence and RAGPipeline on top.  RCCL itself needs >= 2 GPUs: those numbers are the driver's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import retrieve_ref as rr, scan_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spmd_sharded_store_two_ranks(cuda, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_sharded_store_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in range(2):
        v = json.load(open(tmp_path / f"verdict_{rank}.json"))
        assert v["world"] == 2 and len(v["checks"]) > 30
        bad = [name for name, ok in v["checks"] if not ok]
        assert not bad, (rank, bad)


def _chunks(n, seed=0):
    from rag.chunking import Chunk
    rng = np.random.default_rng(seed)
    words = "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi".split()
    return [Chunk(text=" ".join(rng.choice(words, size=int(rng.integers(4, 10)))), chunk_id=f"chunk_{i}", start_char=0,
                  end_char=9, page_number=int(i % 5) + 1, section=None, tokens=5) for i in range(n)]


@pytest.mark.parametrize("dtype,refine", [("fp16", False), ("fp16", True), ("int8", True)])
def test_single_process_two_shards(cuda, dtype, refine):
    from rag.indexing import VectorStore
    n, d = 900, 384
    chunks, emb = _chunks(n, 3), scan_ref.synth_corpus(n, d, seed=4)
    emb[n - 2] = emb[7]                                             # duplicate across the two shards: tie -> lower row first
    store = VectorStore({"devices": ["cuda:0", "cuda:0"], "index_dtype": dtype, "refine_fp32": refine})
    ref = rr.StoreRef()
    for lo, hi in ((0, 301), (301, 302), (302, 900)):
        store.create_index(chunks[lo:hi], emb[lo:hi])
        ref.create_index(chunks[lo:hi], emb[lo:hi])
    st = store.get_stats()
    assert st["count"] == n and len(st["rows_per_device"]) == 2 and sum(st["rows_per_device"]) == n and min(st["rows_per_device"]) > 300
    q = scan_ref.synth_queries(emb, 11, seed=5)
    tol = 2e-6 if refine else 1e-3
    for i in range(11):
        got, exp = store.search(q[i], top_k=6), ref.search(q[i], top_k=6)
        assert got["ids"] == exp["ids"] and got["metadatas"] == exp["metadatas"]
        assert np.abs(np.array(got["distances"][0]) - np.array(exp["distances"][0])).max() < tol
    gb = store.search_batch(q, top_k=9)
    assert [gb["ids"][i] for i in range(11)] == [ref.search(q[i], top_k=9)["ids"][0] for i in range(11)]
    assert store.search(q[0], top_k=5, where={"page_number": 3})["ids"] == ref.search(q[0], top_k=5, where={"page_number": 3})["ids"]
    assert store.search(q[1], top_k=5, where_document={"$not_contains": "beta"})["ids"] == \
        ref.search(q[1], top_k=5, where_document={"$not_contains": "beta"})["ids"]
    assert store.search(q[0], top_k=3, where={"page_number": 99}) == {"ids": [[]], "documents": [[]], "metadatas": [[]], "distances": [[]]}


def test_top_k_above_64_like_the_reference(cuda):
    """The reference accepts any n_results (/root/reference/rag/indexing.py:152-153)."""
    from rag.indexing import VectorStore
    n, d = 3000, 384
    chunks, emb = _chunks(n, 6), scan_ref.synth_corpus(n, d, seed=7)
    q = scan_ref.synth_queries(emb, 3, seed=8)
    for cfg in ({}, {"refine_fp32": True}, {"index_dtype": "int8"}, {"devices": ["cuda:0", "cuda:0"], "refine_fp32": True}):
        store, ref = VectorStore(dict(cfg)), rr.StoreRef()
        store.create_index(chunks, emb); ref.create_index(chunks, emb)
        for k in (65, 200, 5000):
            got, exp = store.search(q[0], top_k=k), ref.search(q[0], top_k=k)
            assert len(got["ids"][0]) == min(k, n) == len(exp["ids"][0])
            gd, ed = np.array(got["distances"][0]), np.array(exp["distances"][0])
            assert np.all(np.diff(gd) >= 0)
            assert np.abs(gd - ed).max() < (5e-3 if cfg.get("index_dtype") == "int8" else 1e-3)
            if cfg.get("refine_fp32"):
                overlap = len(set(got["ids"][0]) & set(exp["ids"][0])) / len(exp["ids"][0])
                assert overlap > 0.999
            assert len(set(got["ids"][0])) == len(got["ids"][0])


def test_two_shard_persistence_round_trip_and_corrupt_file(cuda, tmp_path):
    from rag.indexing import VectorStore
    n = 500
    chunks, emb = _chunks(n, 9), scan_ref.synth_corpus(n, 384, seed=10)
    cfg = {"devices": ["cuda:0", "cuda:0"], "persist_directory": str(tmp_path), "collection_name": "p", "refine_fp32": True}
    a = VectorStore(dict(cfg))
    a.create_index(chunks[:200], emb[:200]); a.create_index(chunks[200:], emb[200:])
    q = scan_ref.synth_queries(emb, 4, seed=11)
    want = [a.search(q[i], top_k=5) for i in range(4)]
    b = VectorStore(dict(cfg))                                     # re-opened from disk, re-sharded over the devices
    assert b.get_stats()["count"] == n and [b.search(q[i], top_k=5) for i in range(4)] == want
    c = VectorStore({"persist_directory": str(tmp_path), "collection_name": "p"})    # ... or onto one device
    assert [c.search(q[i], top_k=5)["ids"] for i in range(4)] == [w["ids"] for w in want]
    with open(tmp_path / "p.slab.bin", "r+b") as fh:              # truncate: must raise, not silently start empty
        fh.truncate(1000)
    with pytest.raises(RuntimeError, match="unreadable"):
        VectorStore(dict(cfg))
    with pytest.raises(NotImplementedError):
        VectorStore({"sharded": True, "persist_directory": str(tmp_path)})
