"""Build-time guard (no GPU): the kernels a plan can select use no scratch memory, and the kernels that pace their LDS
reads / LDS-DMA with hand-counted s_waitcnt values contain no instruction those counts do not expect
(tools/check_resources.py; ADVICE round 2: a compiler bump would otherwise turn a counted wait into a silent LDS race).
Compiles the five translation units that hold the headline scans, the int8 scan and the asm-paced GEMMs (~1 minute on 8 cores);
`python tools/check_resources.py` with no arguments checks all of csrc/."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_selectable_kernels_do_not_spill_and_counted_loops_are_clean():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_resources.py"), "scan_tb.hip", "scan_w1.hip", "scan_i8.hip",
                        "enc_gemm8.hip", "enc_gemm_big.hip"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "0 violation(s)" in r.stdout
