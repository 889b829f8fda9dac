#!/bin/bash
# diagnostic (NOT the metric's configuration): C4 / the 8-GPU rank share at larger query batches -- what coalescing batches would buy
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-46s %9.1f q/s  batch %.4f ms  kern %.4f  %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['kernel_ms'], r['kernel'][:44], c['check_ok']))" "$1"; }
for q in 64 128 256; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 1024 --queries $q 2>gpurun_out/r3_qb_err.log | show "c4 queries/batch=$q" || tail -3 gpurun_out/r3_qb_err.log
  timeout -k 10 300 python3 bench.py --rows 1250000 --no-cpu-baseline --recall-queries 1024 --queries $q 2>gpurun_out/r3_qb_err.log | show "1.25 M rows queries/batch=$q" || tail -3 gpurun_out/r3_qb_err.log
done
