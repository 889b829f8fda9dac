#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('   %-28s %9.1f q/s  batch %.4f ms ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], c['check_ok']))" "$1"; }
python3 -m pytest tests/test_encoder_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -3
for s in 0 1; do for w in c3 c5 c4 c2; do
  CRS_ATTN_SHORT=$s timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_short_err.log | show "short=$s $w" || tail -3 gpurun_out/r3_short_err.log
done; done
for s in 0 1; do CRS_ATTN_SHORT=$s python3 tools/bench_encoder.py bge 2048 16 2>&1 | tail -1; CRS_ATTN_SHORT=$s python3 tools/bench_encoder.py minilm 1024 16 2>&1 | tail -1; done
