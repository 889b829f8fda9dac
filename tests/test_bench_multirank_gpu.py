"""bench.py's N > 1 step on the one-GPU box: two ranks sharing the card, gloo for the collectives (RCCL needs one
GPU per rank).  Everything else is the real path -- graph-captured device segments, the all-gather of the fp16
queries, every rank scanning its shard for the queries of both ranks (the wide scan kernel), the all-gather of
the per-shard lists and the final merge -- and bench.py itself checks the exchanged result against a torch brute
force over both shards (config.exchange_check)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_step_exchange_is_exact(cuda):
    env = dict(os.environ, CRS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--streams", "3", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["queries_per_step"] == 128 and d["config"]["corpus_rows"] == 200_000
    chk = d["config"]["exchange_check"]
    assert chk["ok"] and chk["id_match"] > 0.999 and chk["max_score_err"] < 2e-5
    assert d["config"]["hip_graph"] is True
    assert "scan_wide_kernel" in d["roofline"]["kernel"]
