#!/bin/bash
# Kernel-trace stats of one rank's share of an 8-GPU C4 step (1.25 M rows, queries encoded in shards of 8: single-GPU proxy).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proxy8 -- python3 bench.py --rows 1250000 --no-cpu-baseline --proxy-encode-shard 8 --steps 40 > gpurun_out/proxy8.log 2>&1 || tail -5 gpurun_out/proxy8.log
f=$(ls -t gpurun_out/proxy8/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
tot = 0
for r in rows[1:16]:
    print(r[0][:90].ljust(92), r[1].rjust(7), ("%.1f" % (float(r[3]) / 1000)).rjust(8), "us avg", ("%.1f" % (float(r[2]) / 1e6)).rjust(8), "ms total", r[4])
PY
grep '^{' gpurun_out/proxy8.log | cut -c100-260
