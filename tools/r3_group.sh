#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-44s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f grp %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], c['batches_per_encoder_forward'], c['check_ok']))" "$1"; }
for a in "--encode-group 1" "--encode-group 2" "--encode-group 4" "--encode-group 4 --streams 16" "--encode-group 8 --streams 16" "--encode-group 4 --enc-lanes 1 --lanes split"; do
timeout -k 10 400 python3 bench.py --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_group_err.log | show "c4 $a" || tail -5 gpurun_out/r3_group_err.log
done
for a in "--encode-group 1" "--encode-group 4"  "--encode-group 4 --streams 16" "--encode-group 4 --lanes split" "--encode-group 4 --lanes split --enc-small-lds off"; do
timeout -k 10 400 python3 bench.py --workload c3 --no-cpu-baseline --recall-queries 2048 $a 2>gpurun_out/r3_group_err.log | show "c3 $a" || tail -5 gpurun_out/r3_group_err.log
done
for a in "--encode-group 1" "--encode-group 2" "--encode-group 4"; do
timeout -k 10 300 python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_group_err.log | show "proxy8 $a" || tail -3 gpurun_out/r3_group_err.log
done
