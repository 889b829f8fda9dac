#!/bin/bash
cd "$(dirname "$0")/.."
python3 -m pytest tests/test_caller_contract_gpu.py tests/test_store_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -2
for w in c2 c4; do timeout -k 10 400 python3 bench.py --workload $w --through-pipeline --steps 7 2>gpurun_out/r3_pipe_err.log > gpurun_out/r03_bench_pipeline_$w.json || tail -5 gpurun_out/r3_pipe_err.log; python3 -c "
import json; d=json.loads(open('gpurun_out/r03_bench_pipeline_$w.json').read().strip().splitlines()[-1]); print('   pipeline $w', [(r['queries_per_call'], r['queries_per_s'], r['split_ms']) for r in d['config']['results']])"; done
