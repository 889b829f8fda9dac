#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
python -m pytest tests -q -m gpu > $O/r3_full2.log 2>&1; echo "pytest rc=$?" >> $O/r3_full2.log; tail -6 $O/r3_full2.log
for w in c4 c3; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/r3_b2_$w.json 2> $O/r3_b2_$w.err; echo "$w rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_b2_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['config']['recall_at_10_vs_fp32']; ro=d['roofline']
        print(f, d['value'], d['config']['ms_per_batch'], 'k_scan',d['config']['k_scan'],'kern',ro['kernel_ms'],'tot',ro['scan_merge_refine_ms'],'inrun',ro['search_segment_ms_in_run'],'cert_ms',ro['fp32_rerank_cert_ms'],'frac',ro['frac'],
              'rec',r['timed_path'],'exact',r['queries_exact_up_to_fp32_resolution'],'cert',r['certified_frac'],'esc',r['escalated'],'ok',d['config']['check_ok'])
    except Exception as e: print(f,'ERR',e)
PY
