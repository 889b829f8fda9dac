#!/bin/bash
# C4 (long scan): does grouping the MiniLM forwards pay with ONE search lane?  two passes per setting, one box
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-72s %9.1f q/s  batch %.4f ms  seg_in_run %s %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], c['lanes'][:21], c['check_ok']))" "$1"; }
for rep in 1 2; do
for a in "" "--encode-group 8 --streams 16 --enc-lanes 2 --search-lanes 1" "--encode-group 8 --streams 16 --enc-lanes 1 --search-lanes 1" "--encode-group 16 --streams 32 --enc-lanes 2 --search-lanes 1" "--encode-group 16 --streams 32 --enc-lanes 1 --search-lanes 1" "--encode-group 16 --streams 32 --enc-lanes 1 --search-lanes 2"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_c4l_err.log | show "c4 $a" || tail -3 gpurun_out/r3_c4l_err.log
done; done
