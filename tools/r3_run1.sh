#!/bin/bash
# round-3 checkpoint run: full GPU suite, then the bench lines
cd "$(dirname "$0")/.."
O=gpurun_out
python -m pytest tests -x -q -m gpu > $O/r3_full.log 2>&1; echo "pytest rc=$?" >> $O/r3_full.log; tail -4 $O/r3_full.log
for w in c4 c5 c3 c2; do
  timeout -k 10 300 python bench.py --workload $w > $O/r3_b_$w.json 2> $O/r3_b_$w.err; echo "$w rc=$?"
done
timeout -k 10 300 python bench.py --workload c4 --k-scan 16 --no-cpu-baseline > $O/r3_b_c4_k16.json 2> $O/r3_b_c4_k16.err; echo "c4k16 rc=$?"
timeout -k 10 300 python bench.py --workload c2 --through-pipeline --steps 5 > $O/r3_pipe_c2.json 2> $O/r3_pipe_c2.err; echo "pipe c2 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['config']['recall_at_10_vs_fp32']; ro=d['roofline']
        print(f, d['value'], d['config']['ms_per_batch'], 'k_scan',d['config']['k_scan'],'kern',ro['kernel_ms'],'tot',ro['scan_merge_refine_ms'],'inrun',ro['search_segment_ms_in_run'],'frac',ro['frac'],
              'rec',r['timed_path'],'exact',r['queries_exact_up_to_fp32_resolution'],'cert',r['certified_frac'],'esc',r['escalated'],'unp',r['unproven'],'ok',d['config']['check_ok'], 'err', r['max_abs_score_err_vs_fp64'])
    except Exception as e: print(f,'ERR',e)
PY
