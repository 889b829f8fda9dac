#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('   %-52s %9.1f q/s  batch %.4f ms grp %s sets %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], c['batches_per_encoder_forward'], c['batches_per_step'], c['check_ok']))" "$1"; }
for a in "" "--encode-group 16 --streams 32" "--encode-group 16 --streams 48" "--encode-group 32 --streams 64" "--encode-group 24 --streams 48"; do
  timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_bg_err.log | show "c5 $a" || tail -3 gpurun_out/r3_bg_err.log
done
for a in "" "--encode-group 12 --streams 24" "--encode-group 16 --streams 32"; do
  timeout -k 10 300 python3 bench.py --workload c3 --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_bg_err.log | show "c3 $a" || tail -3 gpurun_out/r3_bg_err.log
done
