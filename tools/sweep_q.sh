#!/bin/bash
# Scan-only sweep over query-batch sizes (diagnostic): tools/sweep_q.sh TAG "c3 c2 c4" "256 512"
TAG=${1:-sw}; WL=${2:-"c3 c2 c4"}; QS=${3:-"256 512"}
mkdir -p gpurun_out
for w in $WL; do for q in $QS; do
  python bench.py --workload $w --queries $q --scan-only --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/${TAG}_${w}_q$q.json 2>gpurun_out/${TAG}_err.log || { tail -5 gpurun_out/${TAG}_err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_${w}_q$q.json"))
print("$w", $q, "q/s", d["value"], "ms/step", d["ms_per_step"], "scan_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], d["config"]["exchange_check"])
PY
done; done
