#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-44s %9.1f q/s  batch %.4f ms  grp %s sets %s lanes %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], c['batches_per_encoder_forward'], c['batches_per_step'], c['lanes'][:21], c['check_ok']))" "$1"; }
python3 -m pytest tests/test_bench_multirank_gpu.py -m gpu -x -q 2>&1 | tail -3
for a in "--rows 1250000 --proxy-encode-shard 8" "--rows 2500000 --proxy-encode-shard 4" "--rows 1250000 --proxy-encode-shard 8" ""; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_prep_err.log | show "$a" || tail -3 gpurun_out/r3_prep_err.log
done
bash tools/r3_rehearse.sh
