run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d['value'], 'q/s  ms/batch', d['config']['ms_per_batch'], 'scan', d['roofline']['kernel_ms'])"; }
for rep in 1 2; do
echo "c3 default";                                  run --workload c3
echo "c3 CRS_SPLITK_MAX_TOKENS=2048";               CRS_SPLITK_MAX_TOKENS=2048 run --workload c3
echo "c3 round-1 dispatch";                         CRS_PANEL_KC=384 CRS_ENC_PANEL_MULTI=0 CRS_SPLITK_MAX_TOKENS=2048 CRS_PANEL_MAX_SPLIT=8 run --workload c3
echo "c3 1 stream default";                         run --workload c3 --streams 1
echo "c3 1 stream SPLITK 2048";                     CRS_SPLITK_MAX_TOKENS=2048 run --workload c3 --streams 1
done
