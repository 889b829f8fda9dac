#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-72s %9.1f q/s  batch %.4f ms  seg_in_run %s  ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], c['check_ok']))" "$1"; }
P="--rows 1250000 --proxy-encode-shard 8"
for a in "$P" "$P --encode-group 8 --streams 16 --lanes split --enc-lanes 1 --post-lane on" "$P --encode-group 4 --lanes split --enc-lanes 1 --post-lane on" "$P --encode-group 8 --streams 16 --lanes split --enc-lanes 1 --search-lanes 2 --post-lane off" "$P --encode-group 8 --streams 16 --lanes split --enc-lanes 1 --search-lanes 2 --post-lane on" "$P --lanes split --post-lane on" \
  "--encode-group 4 --lanes split --enc-lanes 1 --post-lane on" "--encode-group 8 --streams 16 --lanes split --enc-lanes 1 --post-lane on" "--lanes split --post-lane on" "--workload c3 --post-lane on" "--workload c3 --post-lane off" "--workload c5 --post-lane on" "--workload c5 --post-lane off"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_p8_err.log | show "$a" || tail -3 gpurun_out/r3_p8_err.log
done
