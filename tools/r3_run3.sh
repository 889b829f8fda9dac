#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
python -m pytest tests/test_store_gpu.py tests/test_sharded_store_gpu.py tests/test_torch_ops_gpu.py tests/test_scan_wide_gpu.py tests/test_scan_classic_gpu.py tests/test_exact_gpu.py tests/test_caller_contract_gpu.py tests/test_c1_known_answers_gpu.py tests/test_scan_gpu.py -q -m gpu > $O/r3_t3.log 2>&1; echo "pytest rc=$?" >> $O/r3_t3.log; tail -5 $O/r3_t3.log
timeout -k 10 300 python bench.py --workload c2 --through-pipeline --steps 5 > $O/r3_pipe_c2.json 2> $O/r3_pipe_c2.err; echo "pipe c2 rc=$?"
timeout -k 10 900 python bench.py --workload c4 --through-pipeline --steps 5 > $O/r3_pipe_c4.json 2> $O/r3_pipe_c4.err; echo "pipe c4 rc=$?"; tail -3 $O/r3_pipe_c4.err
python - <<'PY'
import json
for w in ('c2','c4'):
    try:
        d=json.loads(open(f'gpurun_out/r3_pipe_{w}.json').read().strip().splitlines()[-1])
        for r in d['config']['results']: print(w, json.dumps(r))
    except Exception as e: print(w,'ERR',e)
PY
