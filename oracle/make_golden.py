#!/usr/bin/env python3
"""Generates tests/golden/* in the BUILD container (never on the GPU box).

What pins what:
  retrieve_cases.json, distance_table.json
      outputs of the reference's own ContextRetriever (/root/reference/rag/retrieval.py), loaded
      standalone with stub sibling modules and driven with a duck-typed fake store / embedder.
      -> pins oracle/retrieve_ref.py AND the product's rag/retrieval.py.
  encoder_tiny.npz, encoder_minilm.npz, encoder_bge.npz
      outputs of the container-local transformers.BertModel (the library the reference's
      sentence-transformers wraps; random-init, weights from oracle/encoder_ref.make_weights)
      + the published Pooling / Normalize steps.  -> pins oracle/encoder_ref.py.
  clean_text.json
      outputs of the reference's own DocumentProcessor (/root/reference/rag/document_processing.py:129-217:
      _clean_text / process_string / process_text / extract_sections), loaded standalone (PyPDF2 is optional
      there, :8-11).  -> pins the product's rag/document_processing.py bit-for-bit.
      The reference's rag/chunking.py raises ImportError at import without nltk (:10-20); loading it would
      need a stand-in module, so the chunker stays "parity unpinned" (DESIGN section 5).
  scan_g2.npz
      exact top-k from an independent torch fp64 matmul + stable sort (ChromaDB itself is not
      installable here: the search boundary stays "parity unpinned").

Only data (inputs + expected outputs) is written; no reference source text is stored.
Run:  python oracle/make_golden.py
"""
from __future__ import annotations

import importlib.util
import itertools
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def load_reference_retriever():
    saved = {k: sys.modules.get(k) for k in ("rag", "rag.indexing", "rag.embedding")}
    for name, attrs in (("rag", {}), ("rag.indexing", {"VectorStore": object}),
                        ("rag.embedding", {"EmbeddingModel": object})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("_ref_retrieval", os.path.join(REF, "rag", "retrieval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    return mod.ContextRetriever


WORDS = ("retrieval augmented generation improves factual grounding of language models by fetching "
         "passages from a vector index quantization compresses weights to four bits while keeping "
         "perplexity close to the full precision baseline attention heads mix token information across "
         "the sequence dense embeddings are compared with cosine similarity and the nearest chunks are "
         "returned to the prompt what is how does why the a of in").split()


def make_text(rng, n_words):
    return " ".join(rng.choice(WORDS, size=n_words))


class FakeCollection:
    def __init__(self, space):
        self.metadata = {"hnsw:space": space} if space else {}


class FakeStore:
    """Canned nearest-neighbour list: returns the first top_k entries, ChromaDB-shaped."""
    def __init__(self, ids, docs, metas, dists, space="cosine", have_collection=True):
        self.ids, self.docs, self.metas, self.dists = ids, docs, metas, dists
        self.collection = FakeCollection(space) if have_collection else None
        self.calls = []

    def search(self, query_embedding, top_k=5, where=None, where_document=None):
        self.calls.append({"shape": list(np.asarray(query_embedding).shape), "top_k": top_k, "where": where})
        n = min(top_k, len(self.ids))
        return {"ids": [self.ids[:n]], "documents": [self.docs[:n]], "metadatas": [self.metas[:n]],
                "distances": [self.dists[:n]]}


class FakeEmbedder:
    def __init__(self, table, dim):
        self.table, self.dim = table, dim

    def embed(self, texts, show_progress=False):
        if isinstance(texts, str):
            texts = [texts]
        return np.stack([self.table[t] for t in texts]).astype(np.float32)


def gen_retrieve_cases(ContextRetriever):
    rng = np.random.default_rng(20240611)
    cases = []
    dim = 16
    grid = itertools.product([False, True], [0.0, 0.1, 0.5], [0.0, 0.3, 0.9], [1, 3, 5, 10, 20])
    for ci, (rr, div, thr, k) in enumerate(grid):
        n_avail = int(rng.integers(0, 45))
        if ci % 11 == 0:
            n_avail = 0
        dists = np.sort(rng.uniform(0.0, 1.2, size=n_avail)).tolist()
        if n_avail > 3 and ci % 5 == 0:
            dists[2] = dists[1]  # an exact tie
        if n_avail > 1 and ci % 7 == 0:
            dists[0] = -0.0001  # negative distance: clamp path
        ids = [f"chunk_{i}" for i in range(n_avail)]
        docs = [make_text(rng, int(rng.integers(5, 40))) for _ in range(n_avail)]
        if n_avail > 4 and ci % 3 == 0:
            docs[3] = docs[0]  # duplicate text -> identical embeddings -> MMR similarity 1
        metas = [{"page_number": int(rng.integers(1, 15))} for _ in range(n_avail)]
        query = make_text(rng, int(rng.integers(1, 9)))
        if ci % 13 == 0:
            query = ""
        table = {}
        for t in docs + [query]:
            if t not in table:
                v = rng.standard_normal(dim).astype(np.float32)
                if ci % 4 == 0:
                    v = np.abs(v)  # all-positive -> large positive similarities
                table[t] = v
        space = ["cosine", "cosine", "l2", "ip", "weird"][ci % 5] if ci % 9 == 0 else "cosine"
        store = FakeStore(ids, docs, metas, dists, space=space, have_collection=(ci % 6 != 1))
        emb = FakeEmbedder(table, dim)
        cfg = {"top_k": 3, "similarity_threshold": thr, "rerank": rr, "diversity_penalty": div}
        r = ContextRetriever(vector_store=store, embedding_model=emb, config=cfg)
        use_override = (ci % 2 == 0)
        got = r.retrieve(query, top_k=k if use_override else None)
        calls = [dict(c) for c in store.calls]   # one search per retrieve()
        cases.append({
            "config": cfg, "top_k_arg": k if use_override else None, "query": query,
            "space": space, "have_collection": store.collection is not None,
            "store": {"ids": ids, "documents": docs, "metadatas": metas, "distances": dists},
            "embeddings": {t: table[t].tolist() for t in table},
            "search_calls": calls,
            "metric_used": r.distance_metric,
            "expected": [{"chunk_id": c["chunk_id"], "score": c["score"], "distance": c["distance"],
                          "rerank_score": c.get("rerank_score"), "text": c["text"], "metadata": c["metadata"]}
                         for c in got],
            "expected_context_string": r.get_context_string(query, top_k=k if use_override else None),
        })
    return cases


def gen_distance_table(ContextRetriever):
    rows = []
    for space in ("cosine", "l2", "ip", "other"):
        store = FakeStore([], [], [], [], space=space)
        r = ContextRetriever(vector_store=store, embedding_model=None, config={})
        for d in (-0.1, 0.0, 1e-7, 0.1, 0.12, 0.3, 0.5, 1.0, 1.4142135623730951, 1.9999, 2.0, 2.5, 10.0):
            rows.append({"metric": space, "distance": d, "similarity": r._distance_to_similarity(d)})
    return rows


def gen_encoder(cfg_name, cfg, batch, seq, seed):
    import torch
    from transformers import BertConfig, BertModel
    from oracle import encoder_ref as er
    w = er.make_weights(cfg, seed=seed)
    hf = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                    num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                    max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.type_vocab,
                    layer_norm_eps=cfg.ln_eps, hidden_act="gelu", hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0)
    model = BertModel(hf, add_pooling_layer=False)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m and "token_type_ids" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    ids, mask = er.synth_tokens(cfg, batch, seq, seed=seed + 1)
    with torch.no_grad():
        hidden = model(input_ids=torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long(),
                       token_type_ids=torch.zeros(ids.shape, dtype=torch.long)).last_hidden_state
        m = torch.from_numpy(mask).float()
        mean = (hidden * m[..., None]).sum(1) / m.sum(1, keepdim=True).clamp(min=1e-9)
        cls = hidden[:, 0]
        out = {
            "mean_norm": torch.nn.functional.normalize(mean, p=2, dim=1).numpy(),
            "cls_norm": torch.nn.functional.normalize(cls, p=2, dim=1).numpy(),
            "mean_raw": mean.numpy(),
        }
    np.savez_compressed(os.path.join(OUT, f"encoder_{cfg_name}.npz"), ids=ids, mask=mask, seed=np.int64(seed),
                        hidden_first_row=hidden[0].numpy().astype(np.float32), **out)


def gen_scan():
    import torch
    from oracle import scan_ref
    c = scan_ref.synth_corpus(4096, 384, seed=1234)
    q = scan_ref.synth_queries(c, 8, seed=4321)
    c16 = c.astype(np.float16)
    q16 = q.astype(np.float16)
    # planted exact duplicates (ties) of the best match of query 0 and query 3
    f = q16.astype(np.float64) @ c16.astype(np.float64).T
    for qi, spots in ((0, (7, 2000, 4095)), (3, (1, 2, 3))):
        b = int(f[qi].argmax())
        for s in spots:
            c16[s] = c16[b]
    full = torch.from_numpy(q16.astype(np.float64)) @ torch.from_numpy(c16.astype(np.float64)).T
    k = 10
    # independent of oracle/scan_ref: stable descending sort => equal scores keep ascending id order
    order = torch.sort(full, dim=1, descending=True, stable=True).indices[:, :k]
    np.savez_compressed(os.path.join(OUT, "scan_g2.npz"), q16=q16, c16=c16, k=np.int64(k),
                        ids=order.numpy().astype(np.int64),
                        scores=torch.gather(full, 1, order).numpy().astype(np.float32))


def gen_clean_text():
    """tests/golden/clean_text.json from the reference's DocumentProcessor."""
    import tempfile
    spec = importlib.util.spec_from_file_location("_ref_docproc", os.path.join(REF, "rag", "document_processing.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(77)
    raw = [
        "", "   ", "a  b\n\n c\t d", "12", " 12 ", "x \n 12 \n y", "Page 3 of the report", "see page 12 and PAGE 7.",
        "Prior work [12] and [3], also (Smith et al., 2020) and (see Fig. 2) but (in 19999 cases) too.",
        "Visit http://example.com/a?b=1 or https://x.org. Then continue", "no url at all",
        "ef\u00ef\u00acne \u00ef\u00ac\u201aow \ufb01x \ufb02y \u201cquoted\u201d \u2018single\u2019",
        "\"dq\" 'sq' it's", "weird , \"'\").replace( artefact", "(2020)", "[1][2][3]", "(a (b 2021) c)",
        "Multi\n\nparagraph text\n\nwith   blank    lines\n\n\nand a trailing page number\n\n14\n",
        "Intro\f second page 2024 \f third",
    ]
    for _ in range(25):      # seeded mixes of the above ingredients
        parts = []
        for _ in range(int(rng.integers(2, 8))):
            parts.append(str(rng.choice(["alpha beta", "Page 9", "[7]", "(Doe 1999)", "https://a.b/c", "\n\n", "  ", "\u00ef\u00ac", "gamma.",
                                         "12", "\t", "(no year)", "delta [x]", "Section 2 Methods", "\u00ef\u00ac\u201a"])))
        raw.append(" ".join(parts))
    cases = []
    for cfg in ({}, {"remove_headers": False}, {"remove_citations": False}, {"remove_headers": False, "remove_citations": False}):
        dp = m.DocumentProcessor(dict(cfg))
        for t in raw:
            cases.append({"config": cfg, "input": t, "clean": dp._clean_text(t), "process_string": dp.process_string(t)})
    dp = m.DocumentProcessor({})
    files = []
    with tempfile.TemporaryDirectory() as td:
        for name, body in (("a.txt", raw[17]), ("b.md", "# Title\n\nBody [1] text (Roe 2001)."), ("c.markdown", raw[18]), ("empty.txt", "  \n ")):
            pth = os.path.join(td, name)
            with open(pth, "w", encoding="utf-8") as fh:
                fh.write(body)
            files.append({"name": name, "body": body, "pages": [list(p) for p in dp.process_file(pth)]})
    sect_inputs = [
        "# Intro\nhello\nworld\n## Methods\nwe did\n\n### Deep\nthings",
        "preamble line\nAbstract\nthis is it\nRelated Work:\nstuff here\n2. Experiments\nnumbers\n3 Results\nmore",
        "Introduction\n\nfirst\nsecond\nlowercase header\nNot A header because 1 digit\nConclusion\n",
        "", "only text without headers", "Header Only", "# A\n# B\nbody of b",
    ]
    sections = [{"input": t, "sections": m.DocumentProcessor.extract_sections(dp, t)} for t in sect_inputs]   # via the class: the
    # instance attribute of the same name (the config flag, :31) shadows the method on instances
    with open(os.path.join(OUT, "clean_text.json"), "w") as fh:
        json.dump({"clean": cases, "files": files, "sections": sections}, fh, ensure_ascii=True)


def gen_c1_known_answers():
    """tests/golden/c1_known_answers.json: the reference's only end-to-end known answers for the hot path -- per question
    the `context_scores` of the retrieved chunks (+ chunk count and context length) as its committed Kaggle run logged them
    (/root/reference/results/mistral_fp16/detailed_responses.json; identical in the AWQ run, i.e. they depend on the
    embed -> index -> retrieve path alone).  Data only: questions and numbers."""
    src = "/root/reference/results/mistral_fp16/detailed_responses.json"
    with open(src) as fh:
        rows = json.load(fh)
    out = {"_source": "results/mistral_fp16/detailed_responses.json of the reference (config.json: all-MiniLM-L6-v2, top_k 3, rerank, "
                      "diversity_penalty 0.1, similarity_threshold 0.3; data/2308.07633v4-clean.pdf)",
           "cases": [{"question": r["question"], "context_scores": r["context_scores"], "num_chunks_retrieved": r["num_chunks_retrieved"],
                      "context_length_chars": r["context_length_chars"]} for r in rows]}
    with open(os.path.join(OUT, "c1_known_answers.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("c1_known_answers.json:", len(out["cases"]), "questions")


def main():
    os.makedirs(OUT, exist_ok=True)
    gen_clean_text()
    CR = load_reference_retriever()
    with open(os.path.join(OUT, "retrieve_cases.json"), "w") as fh:
        json.dump(gen_retrieve_cases(CR), fh)
    with open(os.path.join(OUT, "distance_table.json"), "w") as fh:
        json.dump(gen_distance_table(CR), fh, indent=1)
    from oracle import encoder_ref as er
    gen_encoder("tiny", er.TINY, batch=4, seq=24, seed=11)
    gen_encoder("minilm", er.MINILM_L6, batch=3, seq=32, seed=12)
    gen_encoder("bge", er.BGE_BASE, batch=2, seq=16, seed=13)
    gen_scan()
    gen_ir_metrics()
    gen_c1_known_answers()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))



def gen_ir_metrics():
    """tests/golden/ir_metrics.json: outputs of the reference's evaluation/retrieval/retrieval_metrics.py
    (RetrievalMetrics static methods; loads standalone, numpy only) on seeded random rankings."""
    import random
    spec = importlib.util.spec_from_file_location("_ref_metrics", os.path.join(REF, "evaluation", "retrieval", "retrieval_metrics.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    R = m.RetrievalMetrics
    rng = random.Random(7)
    cases = []
    for _ in range(60):
        n_ret = rng.choice([0, 1, 3, 5, 10, 12])
        retrieved = [f"chunk_{rng.randrange(30)}" for _ in range(n_ret)]
        relevant = sorted({f"chunk_{rng.randrange(30)}" for _ in range(rng.choice([0, 1, 2, 5, 10]))})
        row = {"retrieved": retrieved, "relevant": relevant, "mrr": R.mean_reciprocal_rank(retrieved, set(relevant)),
               "ap": R.average_precision(retrieved, set(relevant))}
        for k in (0, 1, 3, 5, 10):
            row[f"p@{k}"] = R.precision_at_k(retrieved, set(relevant), k)
            row[f"r@{k}"] = R.recall_at_k(retrieved, set(relevant), k)
            row[f"f@{k}"] = R.f1_at_k(retrieved, set(relevant), k)
        cases.append(row)
    with open(os.path.join(OUT, "ir_metrics.json"), "w") as fh:
        json.dump(cases, fh)

if __name__ == "__main__":
    main()
