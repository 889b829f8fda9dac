"""Shared acceptance check for top-k results (used by the GPU parity tests).

Integer part of the contract (ids) is exact: the returned id set must equal the oracle's wherever
the oracle's fp64 scores separate candidates by more than fp32 accumulation noise; inside such a
near-tie band either candidate is accepted but the reported score must still match.  Exact fp32
ties (duplicate rows) must come out lower-id first.  Score tolerance 1e-3 is north_star's; the
checks below use a much tighter 2e-5 because the kernel accumulates the same fp16 products in fp32.
"""
import numpy as np

SCORE_TOL = 1e-3      # BASELINE.json north_star: cosine scores within 1e-3
TIGHT_TOL = 2e-5      # what fp32 accumulation of exact fp16 products actually delivers
BAND = 4e-6           # fp64-score gap below which a rank swap is not an error


def check_topk(gpu_s, gpu_i, full64, k, id_base=0, band=BAND, tight=TIGHT_TOL):
    gpu_s = np.asarray(gpu_s); gpu_i = np.asarray(gpu_i)
    nq, n = full64.shape
    kk = min(k, n)
    for r in range(nq):
        ids = gpu_i[r]; sc = gpu_s[r]
        assert (ids[kk:] == -1).all(), f"q{r}: tail ids must be -1"
        assert np.isneginf(sc[kk:]).all(), f"q{r}: tail scores must be -inf"
        ids = ids[:kk] - id_base; sc = sc[:kk]
        assert ((ids >= 0) & (ids < n)).all(), f"q{r}: id out of range {ids}"
        assert len(set(ids.tolist())) == kk, f"q{r}: duplicate ids {ids}"
        true = full64[r, ids]
        assert np.abs(true - sc).max() <= tight, f"q{r}: score err {np.abs(true - sc).max()}"
        assert np.abs(true - sc).max() <= SCORE_TOL
        # order: descending score, equal fp32 score -> ascending id
        for a in range(kk - 1):
            assert sc[a] > sc[a + 1] or (sc[a] == sc[a + 1] and ids[a] < ids[a + 1]), \
                f"q{r}: order violated at {a}: {sc[a]},{ids[a]} vs {sc[a+1]},{ids[a+1]}"
        # set: everything clearly above the true k-th best must be present; nothing clearly below
        order = np.sort(full64[r])[::-1]
        kth = order[kk - 1]
        must = set(np.nonzero(full64[r] > kth + band)[0].tolist())
        got = set(ids.tolist())
        assert must <= got, f"q{r}: missing ids {sorted(must - got)[:5]}"
        assert (true >= kth - band).all(), f"q{r}: id below the k-th best returned"


def check_exact_ids(gpu_i, ref_i):
    assert np.array_equal(np.asarray(gpu_i), np.asarray(ref_i))
