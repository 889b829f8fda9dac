#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; q=c['recall_at_10_vs_fp32']; print('   %-12s %9.1f q/s  batch %.4f ms certified %s escalated %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], q['certified_frac'], q['escalated'], c['check_ok']))" "$1"; }
python3 -m pytest tests/test_exact_gpu.py tests/test_refine_gpu.py tests/test_store_gpu.py tests/test_torch_ops_gpu.py -m gpu -x -q 2>&1 | tail -3
for w in c2 c4 c5; do timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --recall-queries 2048 2>gpurun_out/r3_esc_err.log | show "$w" || tail -3 gpurun_out/r3_esc_err.log; done
timeout -k 10 300 python3 bench.py --workload c5 --exact on --no-cpu-baseline --recall-queries 1024 2>gpurun_out/r3_esc_err.log | show "c5 exact on" || tail -3 gpurun_out/r3_esc_err.log
timeout -k 10 300 python3 bench.py --k-scan 10 --no-cpu-baseline --recall-queries 4096 2>gpurun_out/r3_esc_err.log | show "c4 k'=10" || tail -3 gpurun_out/r3_esc_err.log
timeout -k 10 300 python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_esc_err.log | show "proxy8" || tail -3 gpurun_out/r3_esc_err.log
