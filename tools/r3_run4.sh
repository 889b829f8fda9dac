#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
python -m pytest tests -q -m gpu > $O/r3_full4.log 2>&1; echo "pytest rc=$?" >> $O/r3_full4.log; tail -6 $O/r3_full4.log
for a in "--k-scan 16" "--k-scan 32"; do
  timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['config']['recall_at_10_vs_fp32']; print('c5 $a', d['value'], d['config']['ms_per_batch'], r['timed_path'], r['queries_exact_up_to_fp32_resolution'], r['certified_frac'], r['scan_only_no_refine'], d['roofline']['kernel_ms'])"
done
