#!/bin/bash
# the chip partitioned by CU masks: search lanes on CUs [0, n), encoder lanes on the rest (CRS_SEARCH_CUS=n; 0 = shared)
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-34s %9.1f q/s  batch %.4f ms  seg_in_run %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], c['check_ok']))" "$1"; }
for w in c5 c3 c4; do for n in 0 224 192 160; do
  CRS_SEARCH_CUS=$n timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_part_err.log | show "$w search_cus=$n" || tail -3 gpurun_out/r3_part_err.log
done; done
