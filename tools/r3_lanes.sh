#!/bin/bash
# round-3 lane / over-fetch A/B in one call (recall check kept small: these are timing runs)
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-58s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f  lanes=%s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], c['lanes'], c['check_ok']))" "$1"; }
B="python3 bench.py --no-cpu-baseline --recall-queries 512"
for a in "" "--search-lanes 2" "--k-scan 16" "--k-scan 16 --search-lanes 2" "--enc-lanes 3 --search-lanes 2"; do
  $B --rows 1250000 --proxy-encode-shard 8 $a 2>/dev/null | show "proxy8 1.25M rows $a"
done
for a in "" "--enc-small-lds on" "--lanes split --enc-lanes 3" "--lanes split --enc-lanes 4" "--lanes split --enc-lanes 4 --search-lanes 2"; do
  $B --workload c5 $a 2>/dev/null | show "c5 $a"
done
for a in "" "--enc-small-lds on" "--lanes split --enc-lanes 4" "--streams 12"; do
  $B --workload c3 $a 2>/dev/null | show "c3 $a"
done
for a in "" "--search-lanes 2"; do
  $B --workload c4 $a 2>/dev/null | show "c4 $a"
done
