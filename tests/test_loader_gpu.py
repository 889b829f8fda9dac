"""GPU: EmbeddingModel on a LOCAL sentence-transformers directory (f1) -- the loader the reference's
SentenceTransformer(model_name) call maps to offline (/root/reference/rag/embedding.py:33,65-71) -- against the
encoder oracle fed the same weights and the same token ids, for mean and CLS pooling configs, with the 'bert.'
tensor prefix, modules.json, and casing taken from the tokenizer's own files."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TEXTS = ["The quick brown fox jumps over the lazy dog.", "Retrieval augmented generation embeds chunks of text",
         "vector store cosine similarity", "Query", "a b c d e " * 20, "Café résumé naïve paper"]


@pytest.mark.parametrize("pooling", ["mean", "cls"])
def test_embedding_model_from_local_dir_matches_oracle(cuda, tmp_path, pooling):
    from _modeldir import write_model_dir
    from oracle import encoder_ref as er
    from rag.embedding import EmbeddingModel
    d = str(tmp_path / "model")
    weights, cfg = write_model_dir(d, pooling=pooling, bert_prefix=True, sbert_lower=False, tok_lower=True, tokenizer_json=True, seed=5)
    em = EmbeddingModel({"model_name": d, "batch_size": 4, "normalize": True})
    assert em.get_dimension() == 64 and em.shape.pooling == pooling and em.shape.max_seq == 48
    got = em.embed(TEXTS)
    assert got.shape == (len(TEXTS), 64) and got.dtype == np.float32
    # oracle on the same ids
    ocfg = er.EncoderConfig(cfg["vocab_size"], 64, 2, 4, 128, 64, 2, 1e-12, 48, pooling)
    toks = em.tokenize(TEXTS)
    assert max(len(t) for t in toks) == 48                      # the long text was truncated to max_seq_length
    assert em.tokenize(["THE QUICK"]) == em.tokenize(["the quick"])
    for r, t in enumerate(toks):
        ids = np.asarray(t, dtype=np.int64)[None, :]
        ref = er.encode_ref(ids, np.ones_like(ids), weights, ocfg)
        assert float((got[r] * ref[0]).sum()) > 1 - 2e-4, (r, float((got[r] * ref[0]).sum()))
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    one = em.embed(TEXTS[0])
    assert one.shape == (1, 64) and float((one[0] * got[0]).sum()) > 1 - 1e-5


def test_many_batches_on_two_lanes_equal_one_batch(cuda, tmp_path):
    """embed() of more texts than batch_size runs consecutive batches on two side streams (own workspace each) and
    scatters them back into input order: same embeddings as ONE batch of all texts (up to padding-length effects: none
    for a masked encoder), whatever the text lengths."""
    from _modeldir import write_model_dir
    from rag.embedding import EmbeddingModel
    d = str(tmp_path / "model")
    write_model_dir(d, pooling="mean", bert_prefix=False, sbert_lower=False, tok_lower=True, tokenizer_json=True, seed=9)
    rng = np.random.default_rng(3)
    words = ["vector", "store", "cosine", "retrieval", "chunk", "query", "index", "fox", "dog", "paper"]
    texts = [" ".join(rng.choice(words, size=int(rng.integers(1, 40)))) for _ in range(37)]
    many = EmbeddingModel({"model_name": d, "batch_size": 4, "normalize": True}).embed(texts)
    one = EmbeddingModel({"model_name": d, "batch_size": 64, "normalize": True}).embed(texts)
    assert many.shape == one.shape == (37, 64)
    cos = (many * one).sum(1)
    assert float(cos.min()) > 1 - 1e-5, float(cos.min())
    again = EmbeddingModel({"model_name": d, "batch_size": 4, "normalize": True}).embed(texts)
    assert np.array_equal(many, again)          # no race between the lanes: bit-identical on a re-run
