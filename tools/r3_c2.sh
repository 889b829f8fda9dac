#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-74s %9.1f q/s  batch %.4f ms  seg_in_run %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], c['check_ok']))" "$1"; }
for a in "--lanes split --encode-group 32 --streams 64 --enc-lanes 1 --search-lanes 2" "--lanes split --encode-group 64 --streams 128 --enc-lanes 1 --search-lanes 2" "--lanes split --encode-group 32 --streams 96 --enc-lanes 1 --search-lanes 2" "--lanes split --encode-group 32 --streams 64 --enc-lanes 2 --search-lanes 2" "--lanes split --encode-group 32 --streams 64 --enc-lanes 1 --search-lanes 2 --search-fuse 8"; do
  timeout -k 10 300 python3 bench.py --workload c2 --no-cpu-baseline --recall-queries 4096 $a 2>gpurun_out/r3_c2_err.log | show "c2 $a" || tail -3 gpurun_out/r3_c2_err.log
done
P="--rows 1250000 --proxy-encode-shard 8"
for a in "" "--encode-group 16 --streams 32" "--encode-group 32 --streams 64" "--encode-group 16 --streams 48"; do
  timeout -k 10 300 python3 bench.py $P --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_c2_err.log | show "proxy8 $a" || tail -3 gpurun_out/r3_c2_err.log
done
for a in "" "--encode-group 16 --streams 32 --enc-lanes 1 --search-lanes 2" "--encode-group 32 --streams 64 --enc-lanes 1 --search-lanes 2"; do
  timeout -k 10 300 python3 bench.py --rows 1250000 --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_c2_err.log | show "1.25M full encode $a" || tail -3 gpurun_out/r3_c2_err.log
done
