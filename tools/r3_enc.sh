#!/bin/bash
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "gemm_vs_torch or bge-256x16 or golden" 2>&1 | tail -3
for g in 1 0; do
  echo "== CRS_GEMM8=$g c3"
  CRS_GEMM8=$g timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline --recall-queries 2048 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['ms_per_batch'], d['roofline']['kernel_ms'], d['roofline']['scan_merge_refine_ms'], d['config']['check_ok'])"
  CRS_GEMM8=$g python3 tools/enc_chain_profile.py bge 256
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_chain_bge_256 -- python3 tools/enc_chain_profile.py bge 256 > /dev/null 2>&1
f=$(ls -t gpurun_out/r3_chain_bge_256/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
rows=list(csv.reader(open("$f")))
for r in rows[1:14]:
    print("   ", r[0][:90].ljust(92), r[1].rjust(6), ("%.1f"%(float(r[3])/1000)).rjust(8), "us avg; total %", r[-1] if r else "")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_stats_c3 -- python3 bench.py --workload c3 --no-cpu-baseline --recall-queries 512 > /dev/null 2>&1
f=$(ls -t gpurun_out/r3_stats_c3/*/*kernel_stats.csv | head -1)
python3 - <<PY
import csv
rows=list(csv.reader(open("$f")))
for r in rows[1:16]:
    print("   ", r[0][:90].ljust(92), r[1].rjust(6), ("%.1f"%(float(r[3])/1000)).rjust(8), "us avg")
PY
