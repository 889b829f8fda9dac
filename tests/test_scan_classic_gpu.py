"""The threshold/compaction scan kernels (scan.hip variants 0-3 and the multi-query-block grid) stay the
path for long streams, k > 16 and int8 slabs; short streams and large batches now default to the
group-best kernels.  This module re-runs the scan parity modules in a child process with
CRS_SCAN_TB=0 CRS_SCAN_WIDE=0 (both switches are read once per process) so that the classic path keeps
its small-case coverage: ties, duplicates, adversarial orders, ragged tiles, several query blocks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_classic_kernels_in_child_process(cuda):
    env = dict(os.environ, CRS_SCAN_TB="0", CRS_SCAN_WIDE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_scan_gpu.py"),
                        os.path.join(ROOT, "tests", "test_scan_wide_gpu.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
