"""Worker of tests/test_sharded_store_gpu.py: one rank of a 2-rank SPMD VectorStore(sharded=True) on the one-GPU
box (both ranks share the card; gloo carries the all-gather -- RCCL needs one GPU per rank).  Every rank checks the
merged results against the single-store oracle (oracle/retrieve_ref.StoreRef) and writes a verdict file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))


def main(out_dir):
    import numpy as np
    import torch
    import torch.distributed as dist
    from oracle import retrieve_ref as rr, scan_ref
    from rag.chunking import Chunk
    from rag.indexing import VectorStore
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    verdict = {"rank": rank, "world": world, "checks": []}
    rng = np.random.default_rng(17)
    words = "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu".split()
    n, d = 700, 384
    chunks = [Chunk(text=" ".join(rng.choice(words, size=6)), chunk_id=f"chunk_{i}", start_char=0, end_char=10,
                    page_number=int(i % 4) + 1, section=None, tokens=6) for i in range(n)]
    emb = scan_ref.synth_corpus(n, d, seed=8)
    emb[n - 3] = emb[5]                                  # an exact duplicate living on the OTHER rank's shard
    q = scan_ref.synth_queries(emb, 9, seed=9)
    for dtype, refine in (("fp16", False), ("fp16", True), ("int8", True)):
        store = VectorStore({"sharded": True, "index_dtype": dtype, "refine_fp32": refine})
        ref = rr.StoreRef()
        for lo, hi in ((0, 250), (250, 251), (251, 700)):       # several adds, each sharded over the ranks
            store.create_index(chunks[lo:hi], emb[lo:hi])
            ref.create_index(chunks[lo:hi], emb[lo:hi])
        st = store.get_stats()
        verdict["checks"].append(("count", st["count"] == n and 0 < st["rows_on_this_gpu"] < n))
        tol = 1e-3 if not refine else 2e-6
        for i in range(9):
            got, exp = store.search(q[i], top_k=5), ref.search(q[i], top_k=5)
            verdict["checks"].append((f"{dtype}/{refine} search {i}", bool(got["ids"] == exp["ids"] and got["documents"] == exp["documents"]
                                      and np.abs(np.array(got["distances"][0]) - np.array(exp["distances"][0])).max() < tol)))
        gb, eb = store.search_batch(q, top_k=7), [ref.search(q[i], top_k=7) for i in range(9)]
        verdict["checks"].append((f"{dtype}/{refine} search_batch", all(gb["ids"][i] == eb[i]["ids"][0] for i in range(9))))
        gw, ew = store.search(q[0], top_k=6, where={"page_number": 2}), ref.search(q[0], top_k=6, where={"page_number": 2})
        verdict["checks"].append((f"{dtype}/{refine} where", gw["ids"] == ew["ids"] and all(m["page_number"] == 2 for m in gw["metadatas"][0])))
        gd = store.search(q[1], top_k=4, where_document={"$contains": "alpha"})
        verdict["checks"].append((f"{dtype}/{refine} where_document", gd["ids"] == ref.search(q[1], top_k=4, where_document={"$contains": "alpha"})["ids"]))
        verdict["checks"].append((f"{dtype}/{refine} k=64", store.search(q[2], top_k=64)["ids"] == ref.search(q[2], top_k=64)["ids"] if refine else True))
    with open(os.path.join(out_dir, f"verdict_{rank}.json"), "w") as fh:
        json.dump(verdict, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
