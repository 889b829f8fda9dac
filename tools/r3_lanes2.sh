#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-58s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f  lanes=%s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], c['lanes'], c['check_ok']))" "$1"; }
B="python3 bench.py --no-cpu-baseline --recall-queries 512"
for a in "--k-scan 16" "--k-scan 16 --enc-lanes 3" "--k-scan 16 --enc-lanes 3 --streams 12" "--k-scan 16 --enc-lanes 1" "--k-scan 16 --lanes batch" "--k-scan 16 --lanes batch --enc-small-lds on"; do
  $B --rows 1250000 --proxy-encode-shard 8 $a 2>/dev/null | show "proxy8 1.25M rows $a"
done
for a in "--enc-lanes 3" "--enc-lanes 3 --streams 12"; do
  $B --workload c4 $a 2>/dev/null | show "c4 $a"
done
