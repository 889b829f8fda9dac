"""CPU: host-side logic of the drop-in package (no GPU, no compute calls into the library)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from test_oracle_golden import _cases, assert_case, replay_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- the product ContextRetriever against the reference's own outputs --------------------------------
class _Col:
    def __init__(self, space):
        self.metadata = {"hnsw:space": space}


class _Store:
    def __init__(self, search, space, have_collection):
        self.search = search
        self.collection = _Col(space) if have_collection else None


class _Emb:
    def __init__(self, embed):
        self.embed = embed


@pytest.mark.parametrize("idx", range(90))
def test_product_retriever_matches_reference_golden(idx):
    from rag.retrieval import ContextRetriever
    case = _cases()[idx]

    def fn(query, search, embed, cfg, k_arg, metric):
        r = ContextRetriever(vector_store=_Store(search, case["space"], case["have_collection"]),
                             embedding_model=_Emb(embed), config=cfg)
        assert r.distance_metric == metric
        return r.retrieve(query, top_k=k_arg)
    got, calls = replay_case(case, fn)
    assert_case(case, got, calls)


def test_product_context_string_and_distance_table():
    from rag.retrieval import ContextRetriever
    with open(os.path.join(ROOT, "tests", "golden", "distance_table.json")) as fh:
        for row in json.load(fh):
            r = ContextRetriever(_Store(None, row["metric"], True), None, {})
            assert r._distance_to_similarity(row["distance"]) == row["similarity"]
    for case in _cases()[:30]:
        def fn(query, search, embed, cfg, k_arg, metric):
            r = ContextRetriever(_Store(search, case["space"], case["have_collection"]), _Emb(embed), cfg)
            return r.get_context_string(query, top_k=k_arg)
        got, _ = replay_case(case, fn)
        assert got == case["expected_context_string"]


def test_retrieve_batch_equals_retrieve_per_query():
    from rag.retrieval import ContextRetriever
    cases = [c for c in _cases() if c["store"]["ids"] and c["query"]][:6]
    for case in cases:
        st = case["store"]
        table = {t: np.asarray(v, dtype=np.float32) for t, v in case["embeddings"].items()}

        class S:
            collection = _Col("cosine")

            def search(self, query_embedding, top_k=5, where=None, where_document=None):
                n = min(top_k, len(st["ids"]))
                return {"ids": [st["ids"][:n]], "documents": [st["documents"][:n]],
                        "metadatas": [st["metadatas"][:n]], "distances": [st["distances"][:n]]}

            def search_batch(self, q, top_k=5, where=None, where_document=None):
                one = self.search(None, top_k)
                return {k: [v[0] for _ in range(len(q))] for k, v in one.items()}

        def embed(texts, show_progress=False):
            texts = [texts] if isinstance(texts, str) else texts
            return np.stack([table[t] for t in texts])
        r = ContextRetriever(S(), _Emb(embed), case["config"])
        single = r.retrieve(case["query"])
        batch = r.retrieve_batch([case["query"], case["query"]])
        strip = lambda cs: [(c["chunk_id"], c["score"], c.get("rerank_score")) for c in cs]
        assert strip(batch[0]) == strip(single) and strip(batch[1]) == strip(single)


# ---- C ABI: the library loads and exports everything the headers declare ----------------------------
def _declared_symbols():
    names = set()
    for h in ("crs_hip.h", "crs_encoder.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(crs_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    from rag import _native as nat
    import rag._encoder  # noqa: F401  (registers the encoder entry points)
    lib = nat.load()
    declared = _declared_symbols()
    assert len(declared) >= 13
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert declared == set(nat.exported_symbols()), "binding table and headers disagree"
    assert lib.crs_abi_version() == 3
    assert [lib.crs_padded_dim(d) for d in (1, 100, 128, 384, 768, 1000)] == [128, 128, 128, 384, 768, 1024]


def test_torch_custom_ops_are_registered():
    """libcrs_torch.so loads on a CPU-only host and registers every op of the north_star's
    'PyTorch-ROCm custom ops' boundary under torch.ops.crs (HIP backend only: no CPU kernel exists)."""
    import torch
    from rag import _native as nat
    ops = nat.ops()
    for name in ("slab_append", "queries_to_f16", "cosine_topk", "cosine_topk_out", "refine_f32", "refine_f32_out",
                 "refine_f32_cert_out", "escalate_exact", "merge_topk", "merge_topk_out", "merge_topk_wire_out", "encoder_forward"):
        assert hasattr(ops, name), name
    schema = str(torch.ops.crs.cosine_topk.default._schema)
    assert "Tensor q16, Tensor slab, Tensor? scales, int n_rows, int dim, int k, int id_base" in schema
    with pytest.raises((RuntimeError, NotImplementedError)):       # no CPU implementation, by design
        ops.merge_topk(torch.zeros(2, 3, 4), torch.zeros(2, 3, 4, dtype=torch.long), 2)


def test_argument_validation_without_gpu():
    """Pure argument checks return error codes before any HIP call."""
    from rag import _native as nat
    lib = nat.load()
    out = ctypes.c_size_t(0)
    assert lib.crs_scan_workspace_bytes(0, 384, 10, 1000, ctypes.byref(out)) == -1
    assert lib.crs_scan_workspace_bytes(4, 384, 65, 1000, ctypes.byref(out)) == -1
    assert b"k must be" in lib.crs_last_error()
    assert lib.crs_merge_topk(None, None, 1, 1, 1, 1, None, None, None) == -1
    assert lib.crs_slab_append_f32(None, 5, 384, 7, None, None, None, 0, None, None) == -1
    # exactness workspace: [thr | count | lists], and the analytic row-error bounds (fp16: 2^-11 relative; int8: 1/254 per element)
    assert lib.crs_exact_workspace_bytes(64, 1024) == 2 * 256 + 256 + 64 * 1024 * 8     # thresholds | counters | blocks-through counter | lists
    assert lib.crs_exact_workspace_bytes(0, 1024) == 0
    assert 4.88e-4 < lib.crs_exact_row_error_bound(384, 0) < 4.95e-4
    assert abs(lib.crs_exact_row_error_bound(768, 1) - 768 ** 0.5 / 254) < 1e-4
    assert lib.crs_refine_f32_cert(None, None, 4, 384, 0, None, 100, 0, None, None, 16, 10, 0.0, None, None, None, None, 0, 1024, None) == -1
    assert lib.crs_escalate_exact(None, None, 4, 384, 0, None, None, None, 100, 0, 10, None, None, None, None, 0, 70000, None) == -1
    assert b"cap must be" in lib.crs_last_error()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rag import _native as nat
    from rag.indexing import VectorStore
    from rag.embedding import EmbeddingModel
    from rag.chunking import Chunk
    with pytest.raises(nat.NativeError):
        VectorStore({}).create_index([Chunk("a b", "chunk_0", 0, 3)], np.zeros((1, 384), dtype=np.float32))
    with pytest.raises(nat.NativeError):
        EmbeddingModel({"model_name": "synthetic:tiny"})


# ---- tokenizer / chunker / document processor -----------------------------------------------------------
def test_wordpiece_tokenizer():
    from rag.tokenizer import HashTokenizer, WordPieceTokenizer, basic_tokenize, pad_batch
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "un", "##aff", "##able", "hello", ",", "world", "!", "cafe", "##s", "a"]
    tok = WordPieceTokenizer({t: i for i, t in enumerate(vocab)})
    assert basic_tokenize("Hello, WORLD!  Café's") == ["hello", ",", "world", "!", "cafe", "'", "s"]
    assert tok.encode("unaffable", 16) == [2, 4, 5, 6, 3]
    assert tok.encode("Hello, world!", 16) == [2, 7, 8, 9, 10, 3]
    assert tok.encode("cafés xyz", 16) == [2, 11, 12, 1, 3]            # accent stripped; unknown word -> [UNK]
    assert tok.encode("a " * 50, 8) == [2, 13, 13, 13, 13, 13, 13, 3]   # truncation keeps [CLS] / [SEP]
    assert tok.encode("x" * 101, 8) == [2, 1, 3]                       # > 100 chars -> [UNK]
    ids, lens = pad_batch([[2, 7, 3], [2, 3]])
    assert ids.tolist() == [[2, 7, 3], [2, 3, 0]] and lens.tolist() == [3, 2] and ids.dtype == np.int32
    h = HashTokenizer(30522)
    a, b = h.encode("the same words", 32), h.encode("the same words", 32)
    assert a == b and a[0] == 101 and a[-1] == 102 and all(1000 <= t < 30522 for t in a[1:-1])


def test_chunker_and_document_processor():
    from rag.chunking import Chunk, TextChunker
    from rag.document_processing import DocumentProcessor
    c = TextChunker({"strategy": "fixed", "chunk_size": 5, "chunk_overlap": 2})
    chunks = c.chunk("one two three four five six seven eight", page_num=3)
    assert [x.text for x in chunks] == ["one two three four five", "four five six seven eight", "seven eight"]
    assert [x.chunk_id for x in chunks] == ["chunk_0", "chunk_1", "chunk_2"] and chunks[0].page_number == 3
    assert chunks[0].tokens == 5 and isinstance(chunks[0], Chunk)
    c.reset_chunk_ids()
    assert c.chunk("x y", 1)[0].chunk_id == "chunk_0"
    s = TextChunker({"strategy": "sentence", "chunk_size": 30})
    out = s.chunk("First sentence here. Second one follows. Third.")
    assert len(out) >= 2 and all(len(x.text) <= 40 for x in out)
    sem = TextChunker({"strategy": "semantic", "chunk_size": 80, "min_chunk_size": 10, "chunk_overlap": 2})
    out = sem.chunk("A first paragraph that is long enough to count.\n\nA second paragraph, also long enough.\n\nshort")
    assert len(out) == 2 and out[1].text.startswith("to count.")
    assert TextChunker({}).chunk("   ") == []
    with pytest.raises(ValueError):
        TextChunker({"strategy": "nope"}).chunk("text")
    dp = DocumentProcessor({})
    assert dp.process_string("See   [12] the\nresult (Smith et al., 2020) at https://x.y/z ok") == "See  the result  at  ok"   # like the reference: collapse first, then strip
    with pytest.raises(FileNotFoundError):
        dp.process_file("/nonexistent.txt")


def test_store_host_logic_shapes():
    from rag.indexing import VectorStore
    from rag.chunking import Chunk
    st = VectorStore({"collection_name": "x"})
    assert st.collection is None and st.get_stats() == {"status": "empty", "count": 0}
    with pytest.raises(ValueError, match="No collection available"):
        st.search(np.zeros(4, dtype=np.float32))
    with pytest.raises(ValueError, match="doesn't match embedding count"):
        st.create_index([Chunk("a", "chunk_0", 0, 1)], np.zeros((2, 4), dtype=np.float32))
    assert st.create_index([], np.zeros((0, 4))) is None
    meta = VectorStore._chunk_metadata(Chunk("a", "c", 0, 1, page_number=2, section=None, tokens=1), ["page_number", "section", "tokens"])
    assert meta == {"page_number": 2, "tokens": 1}
    with pytest.raises(ValueError):
        VectorStore({"index_dtype": "fp8"})


def test_synthetic_weights_match_oracle_generator():
    """The product's seeded checkpoint generator and the oracle's are the same streams (bench.py feeds
    the product one to the GPU and the oracle restatement to the CPU baseline)."""
    from oracle import encoder_ref as er
    from rag._encoder import ModelShape
    from rag.embedding import synthetic_weights
    cfg = er.TINY
    shape = ModelShape(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.max_pos, cfg.ln_eps, cfg.pooling, cfg.max_seq)
    a, b = synthetic_weights(shape, 5), er.make_weights(cfg, seed=5)
    assert a.keys() == b.keys()
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_ir_metrics_match_reference_golden():
    from rag import ir_eval
    with open(os.path.join(ROOT, "tests", "golden", "ir_metrics.json")) as fh:
        cases = json.load(fh)
    assert len(cases) == 60
    for c in cases:
        ret, rel = c["retrieved"], set(c["relevant"])
        assert ir_eval.mean_reciprocal_rank(ret, rel) == c["mrr"]
        assert ir_eval.average_precision(ret, rel) == c["ap"]
        for k in (0, 1, 3, 5, 10):
            assert ir_eval.precision_at_k(ret, rel, k) == c[f"p@{k}"]
            assert ir_eval.recall_at_k(ret, rel, k) == c[f"r@{k}"]
            assert ir_eval.f1_at_k(ret, rel, k) == c[f"f@{k}"]
    agg = ir_eval.evaluate_rankings([c["retrieved"] for c in cases], [set(c["relevant"]) for c in cases], ks=(1, 10))
    assert abs(agg["mrr"] - sum(c["mrr"] for c in cases) / 60) < 1e-12 and "f1@10" in agg


def test_short_batches_are_padded_to_a_fused_length():
    from rag.tokenizer import pad_batch
    ids, lens = pad_batch([[101, 7, 102], [101, 102]], 0, short_steps=(16, 32, 64))
    assert ids.shape == (2, 16) and lens.tolist() == [3, 2] and (ids[0, 3:] == 0).all()
    ids, lens = pad_batch([[1] * 40], 0, short_steps=(16, 32, 64))
    assert ids.shape == (1, 64)
    ids, lens = pad_batch([[1] * 70], 0, short_steps=(16, 32, 64))
    assert ids.shape == (1, 70)
    ids, lens = pad_batch([[1] * 5], 0)
    assert ids.shape == (1, 5)


def test_fast_tokenizer_backend_matches_the_restatement(tmp_path):
    """rag.tokenizer.FastWordPieceTokenizer (the `tokenizers` library, the reference's own backend) against the pure
    Python restatement: identical ids on accents, casing, punctuation runs, CJK, control characters, over-long
    words, unknown pieces and truncation."""
    pytest.importorskip("tokenizers")
    from rag.tokenizer import FastWordPieceTokenizer, WordPieceTokenizer, make_wordpiece_tokenizer
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "the", "quick", "brown", "fox", "jump", "##s", "##ed", "##ing", "over", "lazy",
             "dog", ",", ".", "!", "?", "(", ")", "-", "'", "re", "##tri", "##eval", "aug", "##ment", "cafe", "naive", "un",
             "##believ", "##able", "a", "b", "c", "##a", "##b", "##c", "1", "2", "##3", "中", "文", "$", "%", "e", "##x"]
    vocab = {w: i for i, w in enumerate(words)}
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(words) + "\n", encoding="utf-8")
    slow = WordPieceTokenizer(vocab)
    fast = FastWordPieceTokenizer.from_vocab(vocab)
    assert isinstance(make_wordpiece_tokenizer(str(vp)), FastWordPieceTokenizer)
    texts = ["The quick brown fox jumps over the lazy dog.", "Retrieval-augmented... (unbelievable)!?", "Café naïve CAFE",
             "abc cab 123 12 3", "中文 and 文中", "tab\tnew\nline\x00null​zw", "x" * 120 + " fox", "", "   ", "quick$%fox",
             "jumping jumped jumps jumpx", "it's 'quoted'", " ".join(["fox"] * 50)]
    for max_len in (8, 16, 64):
        got = fast.encode_batch(texts, max_len)
        for t, g in zip(texts, got):
            assert g == slow.encode(t, max_len), (t, max_len)
            assert fast.encode(t, max_len) == g


# ---- local sentence-transformers directory loader (f1) -----------------------------------------------------
@pytest.mark.parametrize("tokenizer_json", [False, True])
@pytest.mark.parametrize("pooling", ["mean", "cls"])
def test_load_local_dir_reads_the_sentence_transformers_layout(tmp_path, pooling, tokenizer_json):
    """config.json / model.safetensors ('bert.' prefix) / vocab.txt or tokenizer.json / sentence_bert_config.json /
    modules.json / 1_Pooling/config.json -- what SentenceTransformer(path) reads (/root/reference/rag/embedding.py:33)."""
    from _modeldir import write_model_dir
    from rag.embedding import _load_local_dir
    d = str(tmp_path / "m")
    weights, cfg = write_model_dir(d, pooling=pooling, bert_prefix=True, sbert_lower=False, tok_lower=True,
                                   tokenizer_json=tokenizer_json)
    shape, w, tok, pre_lower, has_norm = _load_local_dir(d)
    assert (shape.hidden, shape.layers, shape.heads, shape.ffn, shape.vocab_size) == (64, 2, 4, 128, cfg["vocab_size"])
    assert shape.pooling == pooling and shape.max_seq == 48 and shape.max_pos == 64
    assert pre_lower is False and has_norm is True
    assert set(weights) <= set(w) and not any(k.startswith("bert.") for k in w)
    for k in weights:
        assert np.array_equal(w[k], weights[k])
    # casing comes from the TOKENIZER's files, not from sentence_bert_config.json (do_lower_case=false there):
    a, b = tok.encode("The Quick Brown FOX", 48), tok.encode("the quick brown fox", 48)
    assert a == b
    vocab = {t: i for i, t in enumerate(open(os.path.join(d, "vocab.txt"), encoding="utf-8").read().split("\n"))}
    assert a == [vocab["[CLS]"], vocab["the"], vocab["quick"], vocab["brown"], vocab["fox"], vocab["[SEP]"]]
    assert tok.encode("Caf\u00e9 r\u00e9sum\u00e9", 48) == tok.encode("cafe resume", 48)          # accents stripped with lower-casing
    assert len(tok.encode("the " * 100, 48)) == 48                                                # truncation to max_seq_length


def test_local_dir_tokenizers_agree(tmp_path, monkeypatch):
    """vocab.txt through the pure-Python restatement == through the `tokenizers` library == tokenizer.json."""
    from _modeldir import write_model_dir
    from rag.tokenizer import tokenizer_from_model_dir
    d = str(tmp_path / "m")
    write_model_dir(d, tokenizer_json=True)
    fast_json = tokenizer_from_model_dir(d)
    os.remove(os.path.join(d, "tokenizer.json"))
    fast_vocab = tokenizer_from_model_dir(d)
    monkeypatch.setenv("CRS_TOKENIZER", "python")
    slow = tokenizer_from_model_dir(d)
    assert type(slow).__name__ == "WordPieceTokenizer" and type(fast_vocab).__name__ == "FastWordPieceTokenizer"
    for text in ["The quick brown fox jumps over the lazy dog.", "Retrieval-augmented generation, embeds CHUNKS!",
                 "unknownword jumped quickly", "caf\u00e9 na\u00efve", "", "a" * 120, "vector   store\tcosine\nsimilarity?"]:
        ids = slow.encode(text, 32)
        assert ids == fast_vocab.encode(text, 32) == fast_json.encode(text, 32), text


def test_cased_tokenizer_config_is_respected(tmp_path):
    from _modeldir import write_model_dir
    from rag.tokenizer import tokenizer_from_model_dir
    d = str(tmp_path / "m")
    write_model_dir(d, tok_lower=False)
    tok = tokenizer_from_model_dir(d)
    assert tok.encode("The", 16) != tok.encode("the", 16)


def test_unsupported_pooling_mode_raises(tmp_path):
    import json as _json
    from _modeldir import write_model_dir
    from rag.embedding import _load_local_dir
    d = str(tmp_path / "m")
    write_model_dir(d)
    _json.dump({"pooling_mode_max_tokens": True}, open(os.path.join(d, "1_Pooling", "config.json"), "w"))
    with pytest.raises(NotImplementedError):
        _load_local_dir(d)


# ---- document processor pinned to the reference's own outputs (f3) ------------------------------------------
def test_document_processor_matches_reference_goldens(tmp_path):
    """tests/golden/clean_text.json was produced by running /root/reference/rag/document_processing.py:129-217
    (oracle/make_golden.py gen_clean_text); the product class must agree bit for bit."""
    from rag.document_processing import DocumentProcessor
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clean_text.json")) as fh:
        g = json.load(fh)
    assert len(g["clean"]) >= 150
    for case in g["clean"]:
        dp = DocumentProcessor(dict(case["config"]))
        assert dp._clean_text(case["input"]) == case["clean"], case
        assert dp.process_string(case["input"]) == case["process_string"]
    dp = DocumentProcessor({})
    for f in g["files"]:
        pth = tmp_path / f["name"]
        pth.write_text(f["body"], encoding="utf-8")
        assert [list(p) for p in dp.process_file(str(pth))] == f["pages"], f["name"]
    for case in g["sections"]:
        assert DocumentProcessor.extract_sections(dp, case["input"]) == case["sections"], case["input"]
    assert dp.extract_sections is False                      # the flag shadows the method on instances, as in the reference
    with pytest.raises(FileNotFoundError):
        dp.process_file(str(tmp_path / "missing.txt"))
    (tmp_path / "x.docx").write_text("x")
    with pytest.raises(ValueError):
        dp.process_file(str(tmp_path / "x.docx"))


def test_ascii_fast_path_of_basic_tokenize_equals_the_general_path():
    import random
    from rag import tokenizer as tk
    rnd = random.Random(3)
    alphabet = [chr(c) for c in range(0, 128)]
    for _ in range(300):
        text = "".join(rnd.choice(alphabet) if rnd.random() < 0.3 else rnd.choice("abcXYZ 019.,-'\t\n") for _ in range(rnd.randint(0, 80)))
        for lower in (True, False):
            fast = tk.basic_tokenize(text, lower)
            slow = tk.basic_tokenize(text + "\u00e9", lower)       # a non-ASCII tail forces the general path
            assert slow[:len(fast)] == fast or slow[:-1] == fast[:-1], (repr(text), fast, slow)
            # exact check: general path on the same text with the tail token removed
            tail = tk.basic_tokenize("x \u00e9", lower)[-1]
            gen = tk.basic_tokenize(text + " " + "\u00e9", lower)
            assert gen[-1] == tail and gen[:-1] == fast, (repr(text), fast, gen)


def test_metadata_filters_inverted_index_equals_row_by_row_evaluation():
    """SlabCollection.rows_matching (inverted index; product) against oracle/retrieve_ref._where_ok / _doc_ok (one row at
    a time) on random metadata, every operator the reference may forward to ChromaDB (rag/indexing.py:129-130,174)."""
    from oracle import retrieve_ref as rr
    from rag.indexing import SlabCollection
    rng = np.random.default_rng(3)
    col = SlabCollection("f", "fp16", False, [])
    words = ["alpha", "beta", "gamma", "delta"]
    for r in range(700):
        meta = {"page_number": int(rng.integers(1, 6)), "tokens": int(rng.integers(3, 40))}
        if rng.random() < 0.5:
            meta["section"] = str(rng.choice(["intro", "method", "results"]))
        if rng.random() < 0.2:
            meta["score"] = float(rng.integers(0, 4)) / 2
        col.ids.append(f"chunk_{r}"); col.metadatas.append(meta)
        col.documents.append(" ".join(rng.choice(words, size=5)))
        if r == 300:          # the index is extended as rows arrive
            assert list(col.rows_matching({"page_number": 2}, None)) == [i for i, m in enumerate(col.metadatas) if m["page_number"] == 2]
    cases = [({"page_number": 2}, None), ({"page_number": {"$eq": 2}}, None), ({"page_number": {"$ne": 2}}, None),
             ({"section": "intro"}, None), ({"section": {"$ne": "intro"}}, None), ({"section": None}, None),
             ({"page_number": {"$in": [1, 3, 77]}}, None), ({"page_number": {"$nin": [1, 3]}}, None), ({"page_number": {"$in": []}}, None),
             ({"tokens": {"$gt": 20}}, None), ({"tokens": {"$lte": 5}}, None), ({"score": {"$gte": 1}}, None), ({"score": {"$lt": 0.75}}, None),
             ({"page_number": 2, "section": "method"}, None), ({"$and": [{"page_number": {"$gte": 2}}, {"tokens": {"$lt": 30}}]}, None),
             ({"$or": [{"section": "results"}, {"page_number": 5}]}, None), ({"page_number": {"$regex": "x"}}, None),
             ({"nokey": 1}, None), ({"page_number": 2.0}, None), ({"page_number": "2"}, None),
             (None, {"$contains": "beta"}), (None, {"$not_contains": "beta"}), ({"page_number": 3}, {"$contains": "gamma alpha"}),
             (None, {"$or": [{"$contains": "alpha alpha"}, {"$contains": "delta delta"}]}),
             (None, {"$and": [{"$contains": "alpha"}, {"$not_contains": "beta"}]})]
    for where, wdoc in cases:
        want = [i for i in range(700) if (not where or rr._where_ok(col.metadatas[i], where)) and (not wdoc or rr._doc_ok(col.documents[i], wdoc))]
        got = col.rows_matching(where, wdoc)
        assert got.tolist() == want, (where, wdoc)
    assert col.rows_matching(None, None) is None and col.rows_matching({}, {}) is None


def test_plan_layout_rules():
    from rag._engine import RetrievalEngine as E
    gb = 1 << 30
    p = E.plan_layout(384, 8 * gb)                                   # C4 on one GPU
    assert (p["pipelined"], p["encode_group"], p["n_ctx"], p["n_enc"], p["n_srch"]) == (True, 16, 32, 1, 1)
    p = E.plan_layout(384, 1 * gb, multi=True)                       # one rank of an 8-GPU step
    assert (p["encode_group"], p["n_ctx"], p["n_enc"], p["n_srch"]) == (32, 64, 1, 2)
    p = E.plan_layout(768, 2 * gb)                                   # C3 / C5
    assert (p["encode_group"], p["n_ctx"], p["n_enc"], p["n_srch"]) == (8, 24, 2, 1)
    p = E.plan_layout(768, 8 * gb, batch_tokens=1024)                # C5: 64-query batches of a bge-class encoder
    assert (p["encode_group"], p["n_ctx"], p["n_enc"], p["n_srch"]) == (16, 32, 2, 1)
    p = E.plan_layout(768, 2 * gb, multi=True)                       # bge-class with N > 1: one stream per batch, as before
    assert p["pipelined"] is False and p["encode_group"] == 1
    p = E.plan_layout(384, 100 << 20)                                # C2: the chain was the batch -- grouped forwards, two search lanes
    assert (p["pipelined"], p["encode_group"], p["n_ctx"], p["n_enc"], p["n_srch"]) == (True, 32, 64, 1, 2)
    p = E.plan_layout(768, 100 << 20)                                # bge-class over a small store: one stream per batch, as before
    assert p["pipelined"] is False and p["encode_group"] == 1 and p["n_ctx"] == 8
    p = E.plan_layout(384, 8 * gb, group_cap=4)                      # a caller whose calls bring four batches
    assert p["encode_group"] == 4 and p["n_ctx"] == 8
    p = E.plan_layout(384, 8 * gb, n_ctx=3)                          # explicit buffer sets win; the group divides them
    assert p["n_ctx"] == 3 and p["encode_group"] == 3
