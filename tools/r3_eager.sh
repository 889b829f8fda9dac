#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-44s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], c['check_ok']))" "$1"; }
for e in 0 1; do
CRS_SEARCH_EAGER=$e timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_eager_err.log | show "c4 search eager=$e" || tail -3 gpurun_out/r3_eager_err.log
CRS_SEARCH_EAGER=$e timeout -k 10 300 python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_eager_err.log | show "proxy8 search eager=$e" || tail -3 gpurun_out/r3_eager_err.log
done
# bge-base forward at 4096 / 8192 / 16384 tokens of 16-token queries: would encoding two or four C3 batches per forward pay?
for b in 256 512 1024; do python3 tools/bench_encoder.py bge $b 16 2>&1 | tail -1; done
