#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-58s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f grp %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], c['batches_per_encoder_forward'], c['check_ok']))" "$1"; }
for a in "--encode-group 8 --streams 16 --lanes split" "--encode-group 4 --streams 16 --lanes split" "--encode-group 16 --streams 32 --lanes split" "--encode-group 8 --streams 24 --lanes split"; do
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_group_err.log | show "c5 $a" || tail -5 gpurun_out/r3_group_err.log
done
for a in "--encode-group 16 --streams 32 --lanes split" "--encode-group 8 --streams 24 --lanes split" "--encode-group 8 --streams 16 --lanes split --exact off"; do
timeout -k 10 400 python3 bench.py --workload c3 --no-cpu-baseline --recall-queries 2048 $a 2>gpurun_out/r3_group_err.log | show "c3 $a" || tail -5 gpurun_out/r3_group_err.log
done
