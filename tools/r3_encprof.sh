#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in enc-bge enc-minilm; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_stats_$w -- python3 bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/r3_stats_$w.log 2>&1
f=$(ls -t gpurun_out/r3_stats_$w/*/*kernel_stats.csv | head -1)
echo "== $w  $(grep '^{' gpurun_out/r3_stats_$w.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
python3 - "$f" <<'PY'
import csv, sys, subprocess
rows = list(csv.reader(open(sys.argv[1])))
names = subprocess.run(['c++filt'], input="\n".join(r[0] for r in rows[1:14]), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows[1:14], names):
    print(n[:110].ljust(112), r[1].rjust(6), ("%.1f" % (float(r[3]) / 1000)).rjust(8), "us avg", ("%.1f" % (float(r[2]) / 1e6)).rjust(8), "ms total", r[-1][:6])
PY
done
