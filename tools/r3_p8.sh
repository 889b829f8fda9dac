#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-44s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], c['check_ok']))" "$1"; }
python3 -m pytest tests/test_scan_dynamic_gpu.py -m gpu -x -q 2>&1 | tail -5
for a in "" "--lanes split --search-lanes 2" "--lanes split --search-lanes 2 --enc-lanes 1" "--lanes split --enc-lanes 3 --search-lanes 2"; do
  timeout -k 10 300 python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_p8_err.log | show "proxy8 $a" || tail -3 gpurun_out/r3_p8_err.log
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 --lanes split --search-lanes 2 2>gpurun_out/r3_p8_err.log | show "c4 2 search lanes" || tail -3 gpurun_out/r3_p8_err.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_p8_err.log | show "c4 default" || tail -3 gpurun_out/r3_p8_err.log
