#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-44s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], c['check_ok']))" "$1"; }
python3 -m pytest tests/test_scan_dynamic_gpu.py tests/test_scan_i8_gpu.py -m gpu -x -q 2>&1 | tail -5
for d in 0 85; do
CRS_TB_DYN=$d timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_i8dyn_err.log | show "c5 dyn=$d" || tail -3 gpurun_out/r3_i8dyn_err.log
done
bash tools/r3_eager.sh
