#!/usr/bin/env python3
"""C1-shaped single-query latency: what evaluation/retrieval/benchmark.py:241-247 of the reference times
(perf_counter around rag_pipeline.retrieve(question)) on a ~14-chunk index with the reference's default retrieval
config (top_k 3, rerank, diversity_penalty 0.1 => 2k over-fetch + MMR re-embedding of the retrieved chunk texts).
The reference's committed runs give 23.7-27.6 ms per query on a Tesla T4 (results/mistral_*/retrieval_results.json);
this is the same call sequence on the MI355X path with seeded synthetic MiniLM weights and synthetic page-long
chunks (no checkpoint / PDF library offline), so it compares latency, not content."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "compressed-rag-suite_amd"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import logging; logging.disable(logging.CRITICAL)
import numpy as np
from rag import RAGPipeline

class Stub:
    def generate(self, prompt, **kw): return "n/a"
rng = np.random.default_rng(0)
words = ("retrieval augmented generation language model quantization weights perplexity attention embedding cosine "
         "similarity vector index chunk context answer question compression memory latency throughput accuracy").split()
docs = [" ".join(rng.choice(words, size=520)) for _ in range(14)]          # ~3.3 k characters per chunk, one chunk per page
cfg = {"chunking": {"strategy": "semantic", "chunk_size": 4000, "chunk_overlap": 0, "min_chunk_size": 100},
       "embedding": {"model_name": "synthetic:minilm", "device": "cuda", "batch_size": 32, "normalize": True},
       "retrieval": {"top_k": 3, "similarity_threshold": 0.0, "rerank": True, "diversity_penalty": 0.1},
       "vector_store": {"collection_name": "c1"}}
p = RAGPipeline(cfg); p.setup(Stub())
t_index = p.index_documents(docs, show_progress=False)
qs = [" ".join(rng.choice(words, size=int(rng.integers(5, 12)))) for _ in range(40)]
for q in qs[:5]: p.retrieve(q)
times = []
for q in qs:
    t0 = time.perf_counter(); ctx = p.retrieve(q); times.append((time.perf_counter() - t0) * 1e3)
print(f"index: {p.get_stats()['vector_store']['count']} chunks in {t_index*1e3:.1f} ms; retrieve(): mean {statistics.mean(times):.3f} ms, "
      f"std {statistics.pstdev(times):.3f}, min {min(times):.3f}, max {max(times):.3f} over {len(times)} queries "
      f"({len(ctx)} chunks returned; reference on T4: 23.7-27.6 ms)")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for q in qs: p.retrieve(q)
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
