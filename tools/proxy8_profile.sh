#!/bin/bash
# Kernel-trace stats of one rank's share of an 8-GPU C4 step (1.25 M rows, queries encoded in shards of 8: single-GPU proxy).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ks in 32 16; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proxy8_k$ks -- python3 bench.py --rows 1250000 --no-cpu-baseline --proxy-encode-shard 8 --steps 40 --recall-queries 512 --k-scan $ks > gpurun_out/proxy8_k$ks.log 2>&1 || tail -5 gpurun_out/proxy8_k$ks.log
f=$(ls -t gpurun_out/proxy8_k$ks/*/*kernel_stats.csv | head -1)
echo "== k_scan $ks"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:18]:
    print(r[0][:100].ljust(102), r[1].rjust(7), ("%.1f" % (float(r[3]) / 1000)).rjust(8), "us avg", ("%.1f" % (float(r[2]) / 1e6)).rjust(8), "ms total")
PY
grep '^{' gpurun_out/proxy8_k$ks.log | cut -c100-200
done
