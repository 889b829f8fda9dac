"""Batched IR evaluation over the MI355X retrieve path.

The reference's harness loops one question at a time (evaluation/retrieval/benchmark.py:241-283) and every
IR field of its committed results is null (results/mistral_fp16/retrieval_results.json:2-15).  This module
fills those fields with ONE encoder pass + ONE scan launch for the whole question set
(``ContextRetriever.retrieve_batch``), using the metric definitions of
/root/reference/evaluation/retrieval/retrieval_metrics.py (precision_at_k :32-45, recall_at_k :47-58,
f1_at_k :60-68, mean_reciprocal_rank :70-78, average_precision :80-98) -- pinned by
tests/golden/ir_metrics.json, which holds that class's own outputs.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Set


def precision_at_k(retrieved: Sequence[str], relevant: Set[str], k: int) -> float:
    top = retrieved[:k]
    if k == 0 or not top:
        return 0.0
    return sum(1 for c in top if c in relevant) / len(top)       # divides by what was returned, not by k


def recall_at_k(retrieved: Sequence[str], relevant: Set[str], k: int) -> float:
    if not relevant:
        return 0.0
    return sum(1 for c in retrieved[:k] if c in relevant) / len(relevant)


def f1_at_k(retrieved: Sequence[str], relevant: Set[str], k: int) -> float:
    p, r = precision_at_k(retrieved, relevant, k), recall_at_k(retrieved, relevant, k)
    return 0.0 if p + r == 0 else 2 * (p * r) / (p + r)


def mean_reciprocal_rank(retrieved: Sequence[str], relevant: Set[str]) -> float:
    for rank, c in enumerate(retrieved, 1):
        if c in relevant:
            return 1.0 / rank
    return 0.0


def average_precision(retrieved: Sequence[str], relevant: Set[str]) -> float:
    if not relevant:
        return 0.0
    hits, total = 0, 0.0
    for rank, c in enumerate(retrieved, 1):
        if c in relevant:
            hits += 1
            total += hits / rank
    return total / len(relevant)


def evaluate_rankings(rankings: List[Sequence[str]], relevant: List[Set[str]], ks: Iterable[int] = (1, 3, 5, 10)) -> Dict:
    """Mean metrics over questions, keyed like the reference's result files (precision@k, recall@k, f1@k, mrr, map)."""
    n = max(len(rankings), 1)
    out: Dict[str, float] = {}
    for k in ks:
        out[f"precision@{k}"] = sum(precision_at_k(r, s, k) for r, s in zip(rankings, relevant)) / n
        out[f"recall@{k}"] = sum(recall_at_k(r, s, k) for r, s in zip(rankings, relevant)) / n
        out[f"f1@{k}"] = sum(f1_at_k(r, s, k) for r, s in zip(rankings, relevant)) / n
    out["mrr"] = sum(mean_reciprocal_rank(r, s) for r, s in zip(rankings, relevant)) / n
    out["map"] = sum(average_precision(r, s) for r, s in zip(rankings, relevant)) / n
    return out


def evaluate_pipeline(pipeline, questions: List[str], relevant: List[Set[str]], ks: Iterable[int] = (1, 3, 5, 10)) -> Dict:
    """Retrieve max(ks) chunks for every question in one batch and score the rankings."""
    ks = tuple(ks)
    batches = pipeline.retrieve_batch(questions, top_k=max(ks))
    rankings = [[c["chunk_id"] for c in chunks] for chunks in batches]
    out = evaluate_rankings(rankings, relevant, ks)
    out["num_questions"] = len(questions)
    out["avg_chunks_retrieved"] = sum(len(r) for r in rankings) / max(len(rankings), 1)
    return out
