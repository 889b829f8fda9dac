"""GPU: the reference's CALLERS replayed against the product with a stub model_interface -- nothing but the call
sequence and the keys they read:
  * /root/reference/main.py:84-109          RAGPipeline(rag_config) -> setup(model_interface) -> index_documents(path)
                                            -> get_stats() -> query(q, return_context=True, return_chunks=True)
                                            -> chunk['score'] (formatted :.3f), chunk['text'][:200], result['answer']
  * /root/reference/evaluation/retrieval/benchmark.py:218-283   index_documents(documents, show_progress=True) ->
        get_stats()['vector_store'].get('count', 0) -> retrieve(question) -> ctx.get('chunk_id', ctx.get('id', ...)),
        ctx 'text' / 'score' -> generate_answer(question, contexts) -> generator.generate_without_context(question)
  * same file :858-910 (ablation)           get / set retriever.top_k around repeated runs, restored afterwards
The rag section is the reference config.json's (:2-38) with the embedding model swapped for seeded synthetic weights
(no checkpoint can be fetched here) and the persist directory pointed at tmp_path."""
import os

import pytest

pytestmark = pytest.mark.gpu


class StubModelInterface:
    """What RAGGenerator needs of a ModelInterface (/root/reference/models/model_interface.py:11-128): generate()."""
    def __init__(self):
        self.prompts = []

    def generate(self, prompt, **kwargs):
        self.prompts.append((prompt, kwargs))
        return "stub answer. " + prompt[-40:].replace("\n", " ")

    def get_model(self):
        return None

    def get_tokenizer(self):
        return None


def _rag_config(tmp_path):
    return {
        "document_processing": {"remove_headers": True, "remove_citations": True, "extract_sections": False},
        "chunking": {"strategy": "semantic", "chunk_size": 512, "chunk_overlap": 128, "min_chunk_size": 150},
        "embedding": {"model_name": "synthetic:minilm", "device": "cuda", "batch_size": 32, "normalize": True},
        "retrieval": {"top_k": 3, "similarity_threshold": 0.3, "rerank": True, "diversity_penalty": 0.1},
        "generation": {"max_new_tokens": 128, "temperature": 0.3, "top_p": 0.9, "do_sample": True, "repetition_penalty": 1.15,
                       "use_chat_template": True},
        "vector_store": {"collection_name": "rag_documents", "persist_directory": str(tmp_path / "vector_db")},
    }


PARAS = [
    "Retrieval augmented generation grounds a language model in passages fetched from a vector index, which reduces "
    "hallucination on knowledge intensive questions and lets the knowledge base change without retraining the model at all.",
    "Post training quantization compresses the weights of a large language model to four bits per parameter while "
    "keeping perplexity close to the full precision baseline, trading a little accuracy for a much smaller memory footprint.",
    "The sentence encoder maps every chunk of the document to a dense embedding; cosine similarity between the query "
    "embedding and the stored chunk embeddings ranks the chunks, and the best few are placed in the prompt as context.",
    "Attention heads mix token information across the whole sequence, while the feed forward blocks transform each "
    "position independently; layer normalisation and residual connections keep the activations well conditioned.",
    "Maximal marginal relevance re-orders the retrieved chunks so that near duplicate passages do not crowd out "
    "complementary evidence, and a lexical overlap re-ranker blends term matches into the dense similarity score.",
]


def test_main_py_call_sequence(cuda, tmp_path):
    from rag import RAGPipeline
    doc = tmp_path / "paper.txt"
    doc.write_text("\n\n".join(PARAS * 3), encoding="utf-8")
    mi = StubModelInterface()
    pipeline = RAGPipeline(_rag_config(tmp_path))                 # main.py:84
    pipeline.setup(mi)                                            # :85
    processing_time = pipeline.index_documents(str(doc))          # :90
    assert isinstance(processing_time, float) and f"{processing_time:.2f}"
    stats = pipeline.get_stats()                                  # :92
    assert stats["vector_store"]["count"] >= 1 and stats["embedding_dim"] == 384
    assert stats["retrieval"] == {"top_k": 3, "similarity_threshold": 0.3, "rerank": True, "diversity_penalty": 0.1,
                                  "distance_metric": "cosine"}
    result = pipeline.query("how does quantization affect perplexity", return_context=True, return_chunks=True)   # :99
    assert set(result) == {"answer", "context", "chunks"}
    assert result["chunks"], "the reference prints the retrieved chunks"
    for chunk in result["chunks"]:                                # :105-109
        assert f"{chunk['score']:.3f}" and isinstance(chunk["text"][:200], str)
        assert {"text", "score", "distance", "metadata", "chunk_id"} <= set(chunk)
    assert isinstance(result["answer"], str) and result["answer"].startswith("stub answer")
    assert isinstance(pipeline.query("what is attention"), str)   # no flags -> the bare answer string
    assert os.path.exists(tmp_path / "vector_db" / "rag_documents.meta.json")       # persist_directory honoured
    assert os.path.exists(tmp_path / "vector_db" / "rag_documents.slab.bin")
    # a second pipeline on the same persist directory re-opens the collection (PersistentClient behaviour)
    again = RAGPipeline(_rag_config(tmp_path)); again.setup(mi)
    assert again.get_stats()["vector_store"]["count"] == stats["vector_store"]["count"]
    assert [c["chunk_id"] for c in again.retrieve("what is attention")] == [c["chunk_id"] for c in pipeline.retrieve("what is attention")]


def test_retrieval_benchmark_call_sequence(cuda, tmp_path):
    from rag import RAGPipeline
    cfg = _rag_config(tmp_path)
    cfg["vector_store"] = {"collection_name": "bench"}            # in-memory, like the Kaggle runs
    cfg["chunking"]["min_chunk_size"] = 50
    mi = StubModelInterface()
    rag_pipeline = RAGPipeline(cfg)
    rag_pipeline.setup(mi)
    documents = [p + "\n\n" + q for p, q in zip(PARAS, PARAS[1:] + PARAS[:1])]
    rag_pipeline.index_documents(documents, show_progress=True)                     # benchmark.py:218
    stats = rag_pipeline.get_stats()
    assert stats["vector_store"].get("count", 0) > 0                                 # :221-223
    questions = ["what does retrieval augmented generation do", "how are chunks ranked", "what is maximal marginal relevance"]
    for question in questions:                                                       # :241-278
        contexts = rag_pipeline.retrieve(question)
        assert 0 < len(contexts) <= 3
        chunk_ids = [ctx.get("chunk_id", ctx.get("id", f"chunk_{i}")) for i, ctx in enumerate(contexts)]
        assert all(cid.startswith("chunk_") for cid in chunk_ids) and len(set(chunk_ids)) == len(chunk_ids)
        assert all(isinstance(ctx.get("text", ctx.get("content", "")), str) and 0.3 <= ctx["score"] <= 1.0 for ctx in contexts)
        answer = rag_pipeline.generate_answer(question, contexts)
        assert isinstance(answer, str) and answer
        assert isinstance(rag_pipeline.generator.generate_without_context(question), str)
    # ablation over k (:858-910): retriever.top_k is read, overridden per run, restored
    original_k = rag_pipeline.retriever.top_k
    assert original_k == 3
    try:
        for k in (1, 3, 5, 10, 20):
            rag_pipeline.retriever.top_k = k
            got = rag_pipeline.retrieve(questions[0])
            assert 0 < len(got) <= k
    finally:
        rag_pipeline.retriever.top_k = original_k
    assert len(rag_pipeline.retrieve(questions[0])) <= 3
    # batched additive entry point agrees with the per-question loop
    assert rag_pipeline.retrieve_batch(questions) == [rag_pipeline.retrieve(q) for q in questions]


def test_retrieve_batch_through_the_engine_on_a_tiny_store(cuda, tmp_path):
    """>= batch_queries questions take the throughput engine (role lanes, encode groups capped by the call size, fused search
    graphs) even over a handful of chunks: same dicts as the per-question loop, in order, for a call that does not fill its last
    batch and for one that brings several batches."""
    from rag import RAGPipeline
    cfg = _rag_config(tmp_path)
    cfg["vector_store"] = {"collection_name": "tiny"}
    cfg["chunking"]["min_chunk_size"] = 50
    rag_pipeline = RAGPipeline(cfg)
    rag_pipeline.setup(StubModelInterface())
    rag_pipeline.index_documents([p + "\n\n" + q for p, q in zip(PARAS, PARAS[1:] + PARAS[:1])], show_progress=False)
    words = "retrieval generation quantization encoder attention relevance chunks model memory evidence passages index".split()
    for n_q in (70, 200):
        questions = [f"what about {words[i % len(words)]} and {words[(3 * i + 1) % len(words)]} number {i}" for i in range(n_q)]
        batch = rag_pipeline.retrieve_batch(questions)
        assert len(batch) == n_q
        singles = [rag_pipeline.retrieve(q) for q in questions]
        same = sum(1 for a, b in zip(batch, singles) if [c["chunk_id"] for c in a] == [c["chunk_id"] for c in b])
        assert same >= n_q - 2          # (batched encoder forwards round differently in the last bit: a near-tie may swap)
        for a, b in zip(batch, singles):
            for ca, cb in zip(a, b):
                if ca["chunk_id"] == cb["chunk_id"]:
                    assert abs(ca["score"] - cb["score"]) < 1e-3
