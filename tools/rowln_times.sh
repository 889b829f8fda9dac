#!/bin/bash
# per-call durations of gemm_rowln2_kernel in a MiniLM index-build forward (out-proj and FFN-down alternate)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rlt -- python3 bench.py --workload enc-minilm --steps 10 --warmup 2 --enc-inflight 1 > gpurun_out/rlt.log 2>&1
f=$(ls -t gpurun_out/rlt/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
seq = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows if "rowln2" in r["Kernel_Name"]]
a, b = seq[0::2], seq[1::2]
print("out-proj %.1f us  FFN-down %.1f us" % (sum(a) / len(a) / 1e3, sum(b) / len(b) / 1e3))
PY
