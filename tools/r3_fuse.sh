#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-34s %9.1f q/s  batch %.4f ms  seg_in_run %s fuse %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], c['searches_per_graph'], c['check_ok']))" "$1"; }
python3 -m pytest tests/test_engine_gpu.py tests/test_caller_contract_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for f in 1 4 8 16; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 --search-fuse $f 2>gpurun_out/r3_fuse_err.log | show "c4 fuse=$f" || tail -3 gpurun_out/r3_fuse_err.log
done; done
for f in 1 4 8; do
  timeout -k 10 300 python3 bench.py --workload c3 --no-cpu-baseline --recall-queries 512 --search-fuse $f 2>gpurun_out/r3_fuse_err.log | show "c3 fuse=$f" || tail -3 gpurun_out/r3_fuse_err.log
  timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline --recall-queries 512 --search-fuse $f 2>gpurun_out/r3_fuse_err.log | show "c5 fuse=$f" || tail -3 gpurun_out/r3_fuse_err.log
done
