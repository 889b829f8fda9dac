for rep in 1 2; do
for m in 0 2 1; do
  for w in "c3 256" "c4 256 --rows 1250000" "c4 512 --rows 1250000"; do
    set -- $w
    CRS_SCAN_W1=$m python bench.py --workload $1 --queries $2 $3 $4 --scan-only --no-cpu-baseline --steps 5 --warmup 1 --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('mode $m', '$1', $2, 'kernel_ms', r['kernel_ms'], 'mfma', r['mfma_frac'], 'hbm', r['hbm_frac'], r['kernel'][:40], 'recall', d['config']['recall_at_10_vs_fp32']['timed_path'])
"
  done
done
done
