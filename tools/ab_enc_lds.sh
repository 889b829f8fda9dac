# A/B on one box: default encoder kernels vs the small-LDS forms that can sit beside running scan workgroups
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d['value'], 'q/s  ms/batch', d['config']['ms_per_batch'], 'scan', d['roofline']['kernel_ms'], 'recall', d['config']['recall_at_10_vs_fp32']['timed_path'])"; }
for rep in 1 2; do
echo "default";                              run
echo "CRS_PANEL_KC=128";                     CRS_PANEL_KC=128 run
echo "CRS_PANEL_KC=128 CRS_ENC_QKVATTN=0";   CRS_PANEL_KC=128 CRS_ENC_QKVATTN=0 run
echo "scan-only";                            run --scan-only
done
