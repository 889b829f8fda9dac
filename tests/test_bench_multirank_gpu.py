"""bench.py's N > 1 step on the one-GPU box: two ranks sharing the card, gloo for the collectives (RCCL needs one
GPU per rank).  Everything else is the real path -- graph-captured device segments, every rank scanning its row
shard (over-fetch k' = 16), the fp32 shadow re-rank written straight into the rank's wire block, ONE all-gather
of the wire blocks and the final merge -- and bench.py itself checks the exchanged result against the exact
fp64 ranking of the unquantised rows of both shards (config.recall_at_10_vs_fp32, config.check_ok; it exits
non-zero when that check fails).  Strong scaling (the default: fixed corpus split over the ranks, fixed 64-query batch
encoded in shards and all-gathered, or replicated with one collective per batch) and weak scaling (queries all-gathered first: the wide scan kernel)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, CRS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "4", "--warmup", "1", "--streams", "3", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_strong_step_is_exact_vs_fp32(cuda):
    d = _run(["--workload", "c4", "--rows", "600000", "--encode", "sharded", "--lanes", "split"], 29533)   # (auto: from 4 GPUs on / 512 MB scans)
    c = d["config"]
    # short scans under role lanes: ONE encoder lane whose forward serves a group of batches (3 buffer sets: a group of 3),
    # two search lanes; the group's embeddings travel in one all-gather
    assert c["lanes"].startswith("1 encoder + 2 search") and c["batches_per_encoder_forward"] == 3
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert c["corpus_rows"] == 600_000 and c["rows_per_gpu"] == 300_000 and c["queries_per_batch"] == 64
    assert c["queries_per_step"] == 64 * 3 and c["collectives_per_batch"] == pytest.approx(1 + 1 / 3, abs=1e-3)
    assert c["query_encode"].startswith("sharded: 32 of 64")            # each rank encodes half of the global batch
    assert c["check_ok"] and c["recall_at_10_vs_fp32"]["timed_path"] == 1.0
    assert c["recall_at_10_vs_fp32"]["max_abs_score_err_vs_fp64"] < 1e-5
    assert c["hip_graph"] is True and c["refine_fp32"] is True
    assert "scan_tb_kernel" in d["roofline"]["kernel"]


def test_two_rank_strong_step_replicated_encode_is_one_collective(cuda):
    d = _run(["--workload", "c4", "--rows", "600000"], 29536)                           # N = 2 default: replicated
    c = d["config"]
    assert c["lanes"].startswith("1 encoder + 2 search") and c["batches_per_encoder_forward"] == 3   # MiniLM-class: role lanes at every size
    assert c["collectives_per_batch"] == 1 and c["query_encode"] == "replicated"
    assert c["check_ok"] and c["recall_at_10_vs_fp32"]["timed_path"] == 1.0


def test_two_rank_weak_step_is_exact_vs_fp32(cuda):
    d = _run(["--workload", "c2", "--scaling", "weak"], 29534)
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert c["corpus_rows"] == 200_000 and c["queries_per_batch"] == 128
    assert c["collectives_per_batch"] == pytest.approx(1 + 1 / 3, abs=1e-3)             # the group's queries in one all-gather, a wire block per batch
    assert c["check_ok"] and c["recall_at_10_vs_fp32"]["timed_path"] == 1.0
    assert "scan_wide_kernel" in d["roofline"]["kernel"]


def test_int8_two_rank_strong_recall_is_one_after_refine(cuda):
    d = _run(["--workload", "c5", "--rows", "400000"], 29535)
    c = d["config"]
    assert c["check_ok"] and c["recall_at_10_vs_fp32"]["timed_path"] == 1.0
    assert c["recall_at_10_vs_fp32"]["scan_only_no_refine"] < 1.0      # int8 alone does flip ranks: the refine is what fixes it


def _run_single_rccl(extra):
    """The N > 1 step on ONE rank over the real RCCL backend (process group of world size 1): the collectives' dtypes
    (uint8 wire blocks, fp32 embeddings), their interplay with graph replay and side streams, and the wire merge."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    env.pop("CRS_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--dist-single", "--steps", "4", "--warmup", "1", "--streams", "3",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_rank_rccl_strong_step(cuda):
    d = _run_single_rccl(["--workload", "c4", "--rows", "1400000"])                      # 1.07 GB scan: auto picks the split lanes
    c = d["config"]
    assert c["lanes"].startswith("1 encoder + 2 search") and c["batches_per_encoder_forward"] == 3   # a short scan: grouped forwards, two search lanes
    assert c["dist_single_rank"] is True and c["collectives_per_batch"] == 1 and c["hip_graph"] is True
    assert c["check_ok"] and c["recall_at_10_vs_fp32"]["timed_path"] == 1.0


def test_single_rank_rccl_grouped_query_gather(cuda):
    """Encode groups with gathered queries over RCCL: the group's embeddings in ONE all-gather (issued on the encoder lane), each
    batch's rows picked out of it, the wire-block all-gathers issued from the two search lanes."""
    d = _run_single_rccl(["--workload", "c4", "--rows", "1400000", "--scaling", "weak"])
    c = d["config"]
    assert c["lanes"].startswith("1 encoder + 2 search") and c["batches_per_encoder_forward"] == 3
    assert c["collectives_per_batch"] == pytest.approx(1 + 1 / 3, abs=1e-3)
    assert c["check_ok"] and c["recall_at_10_vs_fp32"]["timed_path"] == 1.0


def test_single_rank_rccl_weak_step_gathers_queries(cuda):
    d = _run_single_rccl(["--workload", "c2", "--scaling", "weak"])
    c = d["config"]
    assert c["dist_single_rank"] is True and c["collectives_per_batch"] == pytest.approx(1 + 1 / 3, abs=1e-3)
    assert c["check_ok"] and c["recall_at_10_vs_fp32"]["timed_path"] == 1.0
