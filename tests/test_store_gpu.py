"""GPU: the drop-in VectorStore / EmbeddingModel / ContextRetriever / RAGPipeline classes against
the CPU oracle (oracle/retrieve_ref.StoreRef, encoder_ref, retrieve_ref) on the same inputs."""
import numpy as np
import pytest

from oracle import encoder_ref as er, retrieve_ref as rr, scan_ref

pytestmark = pytest.mark.gpu


def _chunks(n, seed=0):
    from rag.chunking import Chunk
    rng = np.random.default_rng(seed)
    words = "alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi omicron pi rho sigma".split()
    out = []
    for i in range(n):
        text = " ".join(rng.choice(words, size=int(rng.integers(4, 12))))
        out.append(Chunk(text=text, chunk_id=f"chunk_{i}", start_char=0, end_char=len(text),
                         page_number=int(rng.integers(1, 6)), section=None, tokens=len(text.split())))
    return out


def _pair(n=500, d=384, seed=1, **cfg):
    from rag.indexing import VectorStore
    chunks = _chunks(n, seed)
    emb = scan_ref.synth_corpus(n, d, seed=seed)
    store = VectorStore({"collection_name": "t", **cfg})
    ref = rr.StoreRef()
    store.create_index(chunks, emb)
    ref.create_index(chunks, emb)
    return store, ref, chunks, emb


def test_search_contract_matches_reference_semantics(cuda):
    store, ref, chunks, emb = _pair()
    q = scan_ref.synth_queries(emb, 6, seed=9)
    for i in range(6):
        got = store.search(q[i:i + 1], top_k=5)
        exp = ref.search(q[i:i + 1], top_k=5)
        assert set(got) == {"ids", "documents", "metadatas", "distances"}
        assert got["ids"] == exp["ids"] and got["documents"] == exp["documents"] and got["metadatas"] == exp["metadatas"]
        assert np.abs(np.array(got["distances"][0]) - np.array(exp["distances"][0])).max() < 1e-3   # fp16 slab
        assert got["distances"][0] == sorted(got["distances"][0])
        assert all(isinstance(x, float) for x in got["distances"][0])
    # 1-D query, list query and top_k larger than the collection
    got = store.search(q[0], top_k=3)
    assert got["ids"] == ref.search(q[0], top_k=3)["ids"]
    got = store.search(list(map(float, q[0])), top_k=3)
    assert got["ids"] == ref.search(q[0], top_k=3)["ids"]
    small, sref, _, e2 = _pair(n=4, seed=3)
    assert len(small.search(q[0], top_k=10)["ids"][0]) == 4
    assert small.get_stats()["count"] == 4 and small.get_stats()["metadata"] == {"hnsw:space": "cosine"}


def test_errors_and_empty_cases(cuda):
    from rag.indexing import VectorStore
    store = VectorStore({})
    assert store.get_stats() == {"status": "empty", "count": 0}
    with pytest.raises(ValueError, match="No collection available"):
        store.search(np.zeros(384, dtype=np.float32))
    store.create_index([], np.zeros((0, 384), dtype=np.float32))       # warning + no-op
    assert store.collection is None
    with pytest.raises(ValueError, match="doesn't match embedding count"):
        store.create_index(_chunks(3), np.zeros((2, 384), dtype=np.float32))
    store.create_index(_chunks(3), scan_ref.synth_corpus(3, 384))
    with pytest.raises(ValueError):
        store.search(np.zeros(100, dtype=np.float32))                    # wrong dimension
    store.delete_collection()
    assert store.collection is None and store.get_stats()["count"] == 0
    store.reset_collection()


def test_incremental_adds_and_filters(cuda):
    from rag.indexing import VectorStore
    chunks = _chunks(300, 5)
    emb = scan_ref.synth_corpus(300, 384, seed=5)
    store, ref = VectorStore({}), rr.StoreRef()
    for lo in range(0, 300, 70):                                         # growth / re-allocation path
        store.create_index(chunks[lo:lo + 70], emb[lo:lo + 70])
        ref.create_index(chunks[lo:lo + 70], emb[lo:lo + 70])
    q = scan_ref.synth_queries(emb, 3, seed=6)
    assert store.search(q[0], top_k=8)["ids"] == ref.search(q[0], top_k=8)["ids"]
    got = store.search(q[1], top_k=6, where={"page_number": 2})
    assert got["ids"] == ref.search(q[1], top_k=6, where={"page_number": 2})["ids"]
    assert all(m["page_number"] == 2 for m in got["metadatas"][0])
    assert store.search(q[1], top_k=6, where={"page_number": 99})["ids"] == [[]]
    batch = store.search_batch(q, top_k=4)
    assert [batch["ids"][i] for i in range(3)] == [ref.search(q[i], top_k=4)["ids"][0] for i in range(3)]


def test_refine_fp32_gives_exact_fp32_ranking(cuda):
    """fp16 storage can flip near-tied ranks; with refine_fp32 the over-fetched candidates are
    re-scored against the fp32 shadow and the order equals the exact fp32 order."""
    store, ref, chunks, emb = _pair(n=3000, seed=11, refine_fp32=True)
    q = scan_ref.synth_queries(emb, 16, seed=12)
    for i in range(16):
        got, exp = store.search(q[i], top_k=10), ref.search(q[i], top_k=10)
        assert got["ids"] == exp["ids"]
        assert np.abs(np.array(got["distances"][0]) - np.array(exp["distances"][0])).max() < 2e-6


def test_persistence_round_trip(cuda, tmp_path):
    from rag.indexing import VectorStore
    cfg = {"collection_name": "persisted", "persist_directory": str(tmp_path)}
    a = VectorStore(cfg)
    chunks, emb = _chunks(50, 8), scan_ref.synth_corpus(50, 384, seed=8)
    a.create_index(chunks, emb)
    q = scan_ref.synth_queries(emb, 1, seed=1)[0]
    first = a.search(q, top_k=5)
    b = VectorStore(cfg)                                                 # re-opens the collection
    assert b.get_stats()["count"] == 50 and b.search(q, top_k=5) == first
    b.delete_collection()
    assert VectorStore(cfg).collection is None


def test_persistence_appends_and_survives_a_torn_add(cuda, tmp_path):
    """create_index appends to the row files (O(batch), not a rewrite of the index); whatever a crashed add left past the
    header's row count is ignored on load and overwritten by the next add; the round-2 .npz format is still read."""
    import json
    import os
    from rag.indexing import VectorStore
    cfg = {"collection_name": "inc", "persist_directory": str(tmp_path), "index_dtype": "int8"}
    chunks, emb = _chunks(90, 4), scan_ref.synth_corpus(90, 384, seed=4)
    a = VectorStore(cfg)
    a.create_index(chunks[:40], emb[:40])
    size40 = os.path.getsize(tmp_path / "inc.slab.bin")
    ino = os.stat(tmp_path / "inc.slab.bin").st_ino
    a.create_index(chunks[40:70], emb[40:70])
    assert os.path.getsize(tmp_path / "inc.slab.bin") == size40 // 40 * 70          # grew by exactly the batch ...
    assert os.stat(tmp_path / "inc.slab.bin").st_ino == ino                          # ... in place (not re-created)
    assert json.load(open(tmp_path / "inc.meta.json"))["n"] == 70
    # a torn add: rows and sidecar lines past the header (the header is written last)
    for name, junk in (("inc.slab.bin", b"\x7f" * 5000), ("inc.scales.bin", b"\x00" * 40), ("inc.shadow.bin", b"\x01" * 3000),
                       ("inc.docs.jsonl", b'{"id": "ghost", "document": "x", "meta')):
        with open(tmp_path / name, "ab") as fh:
            fh.write(junk)
    b = VectorStore(cfg)
    assert b.get_stats()["count"] == 70 and "ghost" not in b.collection.ids
    q = scan_ref.synth_queries(emb, 2, seed=2)
    want = [a.search(q[i], top_k=5) for i in range(2)]
    assert [b.search(q[i], top_k=5) for i in range(2)] == want
    b.create_index(chunks[70:], emb[70:])                                            # overwrites the junk
    a.create_index(chunks[70:], emb[70:])
    c = VectorStore(cfg)
    assert c.get_stats()["count"] == 90 and [c.search(q[i], top_k=5) for i in range(2)] == [a.search(q[i], top_k=5) for i in range(2)]
    assert os.path.getsize(tmp_path / "inc.slab.bin") == size40 // 40 * 90
    # the round-2 format (one .npz + one .json, rewritten per add) still opens, and the next add converts it
    col = c.collection
    legacy = tmp_path / "old"
    os.makedirs(legacy)
    np.savez(legacy / "inc.slab.npz", slab=col.slab[:90].cpu().numpy(), scales=col.scales[:90].cpu().numpy(),
             shadow=col.shadow[:90].cpu().numpy(), n=np.int64(90), dim=np.int64(384), index_dtype=np.str_("int8"))
    json.dump({"ids": col.ids, "documents": col.documents, "metadatas": col.metadatas}, open(legacy / "inc.docs.json", "w"))
    d = VectorStore(dict(cfg, persist_directory=str(legacy)))
    assert d.get_stats()["count"] == 90 and [d.search(q[i], top_k=5)["ids"] for i in range(2)] == [w["ids"] for w in [a.search(q[i], top_k=5) for i in range(2)]]
    d.create_index(_chunks(95, 4)[90:], scan_ref.synth_corpus(5, 384, seed=77))
    assert os.path.exists(legacy / "inc.meta.json") and not os.path.exists(legacy / "inc.slab.npz")
    assert VectorStore(dict(cfg, persist_directory=str(legacy))).get_stats()["count"] == 95


def test_filters_use_the_inverted_index_and_cached_subslabs(cuda):
    from rag.indexing import VectorStore
    n = 4000
    chunks = _chunks(n, 7)
    emb = scan_ref.synth_corpus(n, 384, seed=7)
    store, ref = VectorStore({}), rr.StoreRef()
    store.create_index(chunks, emb); ref.create_index(chunks, emb)
    q = scan_ref.synth_queries(emb, 6, seed=8)
    for where, wdoc in (({"page_number": 2}, None), ({"page_number": {"$in": [1, 3]}}, None), ({"page_number": {"$ne": 0}}, None),
                        (None, {"$contains": "beta"}), ({"page_number": 1}, {"$not_contains": "beta"}), ({"page_number": {"$gt": 1}}, None)):
        for i in range(3):
            assert store.search(q[i], top_k=7, where=where, where_document=wdoc)["ids"] == \
                ref.search(q[i], top_k=7, where=where, where_document=wdoc)["ids"], (where, wdoc)
    assert len(store._filters) == 6 and all(("shards" in e) for e in store._filters.values())
    ent = store._filter_entry({"page_number": 2}, None)
    sub = ent["shards"][0]
    assert sub["n"] == len(ent["rows"]) and sub["slab"].shape[0] == sub["n"]         # compacted once, kept
    assert store._filter_entry({"page_number": 2}, None) is ent                       # ... and reused
    batch = store.search_batch(q, top_k=5, where={"page_number": 2})
    assert [batch["ids"][i] for i in range(6)] == [ref.search(q[i], top_k=5, where={"page_number": 2})["ids"][0] for i in range(6)]
    store.create_index(_chunks(n + 10, 7)[n:], scan_ref.synth_corpus(10, 384, seed=9))   # new rows: cached entries are stale
    assert store._filter_entry({"page_number": 2}, None) is not ent


def test_pipeline_end_to_end_vs_oracle(cuda):
    """index_documents -> retrieve through the product classes (synthetic MiniLM weights, hash
    tokeniser) vs oracle encoder + StoreRef + retrieve_ref fed the same token ids."""
    from rag import RAGPipeline
    from rag.embedding import synthetic_weights
    from rag.tokenizer import pad_batch
    cfg = {"embedding": {"model_name": "synthetic:minilm", "batch_size": 8, "normalize": True, "synthetic_seed": 5},
           "chunking": {"strategy": "sentence", "chunk_size": 120, "min_chunk_size": 20},
           "retrieval": {"top_k": 3, "similarity_threshold": 0.0, "rerank": True, "diversity_penalty": 0.1},
           "vector_store": {"collection_name": "e2e"}}
    pipe = RAGPipeline(cfg)
    pipe.setup(model_interface=None)
    docs = ["Quantization compresses model weights to four bits. Perplexity stays close to the baseline. "
            "Dense retrieval compares embeddings with cosine similarity. The nearest chunks go to the prompt.",
            "Attention heads mix token information across the sequence. Retrieval augmented generation grounds "
            "answers in fetched passages. A vector index stores one embedding per chunk of the document."]
    secs = pipe.index_documents(docs, show_progress=False)
    assert secs > 0 and pipe.get_stats()["vector_store"]["count"] >= 4
    em = pipe.embedding_model
    shape = em.shape
    w = synthetic_weights(shape, 5)
    ocfg = er.EncoderConfig(shape.vocab_size, shape.hidden, shape.layers, shape.heads, shape.ffn, shape.max_pos, 2,
                            shape.ln_eps, shape.max_seq, shape.pooling)

    def oracle_embed(texts):
        if isinstance(texts, str):
            texts = [texts]
        ids, lens = pad_batch(em.tokenize(texts))
        mask = (np.arange(ids.shape[1])[None] < lens[:, None]).astype(np.int32)
        return er.encode_ref(ids, mask, w, ocfg)

    col = pipe.vector_store.collection
    from rag.chunking import Chunk
    ref = rr.StoreRef()
    chunks = [Chunk(t, i, 0, len(t)) for i, t in zip(col.ids, col.documents)]
    ref.create_index(chunks, oracle_embed(col.documents))
    ref.metas = col.metadatas
    for query in ("how does quantization affect perplexity", "what stores the embeddings", "attention heads"):
        got = pipe.retrieve(query)
        exp = rr.retrieve(query, search=ref.search, embed=oracle_embed, top_k=3, similarity_threshold=0.0,
                          do_rerank=True, diversity_penalty=0.1)
        assert [c["chunk_id"] for c in got] == [c["chunk_id"] for c in exp]
        assert np.abs(np.array([c["score"] for c in got]) - np.array([c["score"] for c in exp])).max() < 1e-3
        assert pipe.retrieve_batch([query])[0] == got
    stats = pipe.get_stats()
    assert stats["embedding_dim"] == 384 and stats["retrieval"]["distance_metric"] == "cosine"
    # batched IR evaluation (fills the fields the reference's results leave null), relevance = oracle top-3
    from rag import ir_eval
    qs = ["how does quantization affect perplexity", "what stores the embeddings"]
    rel = [{c["chunk_id"] for c in rr.retrieve(q, search=ref.search, embed=oracle_embed, top_k=3, similarity_threshold=0.0,
                                               do_rerank=True, diversity_penalty=0.1)} for q in qs]
    res = ir_eval.evaluate_pipeline(pipe, qs, rel, ks=(1, 3))
    assert res["recall@3"] == 1.0 and res["precision@3"] == 1.0 and res["mrr"] == 1.0 and res["num_questions"] == 2


@pytest.mark.parametrize("dtype,d", [("fp16", 384), ("fp16", 768), ("int8", 768)])
def test_large_search_batches_agree_with_single_queries(cuda, dtype, d):
    """search_batch with more than 64 queries takes the large-batch scan kernels (scan_wide / scan_w1, several
    query blocks for int8): every row of the batch must equal the one-query search of the same vector, and the
    reference-semantics store on the host."""
    store, ref, chunks, emb = _pair(n=3000, d=d, seed=9, index_dtype=dtype)
    q = scan_ref.synth_queries(emb, 150, seed=10)
    batch = store.search_batch(q, top_k=5)
    assert len(batch["ids"]) == 150
    for i in (0, 1, 63, 64, 65, 127, 128, 149):
        single = store.search(q[i], top_k=5)
        assert batch["ids"][i] == single["ids"][0]
        assert np.allclose(batch["distances"][i], single["distances"][0], atol=1e-6)
    if dtype == "fp16":
        same = np.mean([batch["ids"][i] == ref.search(q[i], top_k=5)["ids"][0] for i in range(150)])
        assert same > 0.97   # fp16 storage vs the fp32 host store: only near-ties may differ
