#!/bin/bash
# what in the encoder slows the co-running sweep: its work or its 38 kernel boundaries?  (C4, dynamic tile schedule on)
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-44s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], c['check_ok']))" "$1"; }
export CRS_TB_DYN=85 CRS_TB_DYN_G=8
for a in "" "--scan-only" "--proxy-encode-shard 8" "--proxy-encode-shard 64" "--enc-lanes 1 --lanes split" "--enc-small-lds off"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_intf_err.log | show "c4 $a" || tail -3 gpurun_out/r3_intf_err.log
done
