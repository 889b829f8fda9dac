#!/bin/bash
# dynamic tile tickets in the chain-mode tile-best scan: C4 default line and the 8-GPU rank proxy per CRS_TB_DYN / CRS_TB_DYN_G
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-34s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f  certified %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], q['certified_frac'], c['check_ok']))" "$1"; }
for v in "0 1" "85 8" "95 8" "100 8" "100 16" "85 16"; do
  set -- $v
  CRS_TB_DYN=$1 CRS_TB_DYN_G=$2 timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_dyn_err.log | show "c4 dyn=$1 x$2" || { tail -5 gpurun_out/r3_dyn_err.log; exit 1; }
done
for v in "0 1" "100 8" "100 4"; do
  set -- $v
  CRS_TB_DYN=$1 CRS_TB_DYN_G=$2 timeout -k 10 300 python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_dyn_err.log | show "proxy8 dyn=$1 x$2" || { tail -5 gpurun_out/r3_dyn_err.log; exit 1; }
done
