#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-30s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f grp %s sets %s lanes %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], c['batches_per_encoder_forward'], c['batches_per_step'], c['lanes'][:22], c['check_ok']))" "$1"; }
for w in c2 c4 c3 c5; do
timeout -k 10 400 python3 bench.py --workload $w --no-cpu-baseline --recall-queries 4096 2>gpurun_out/r3_def_err.log | show "$w defaults" || tail -5 gpurun_out/r3_def_err.log
done
timeout -k 10 300 python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_def_err.log | show "proxy8" || tail -5 gpurun_out/r3_def_err.log
timeout -k 10 300 python3 bench.py --rows 2500000 --proxy-encode-shard 4 --no-cpu-baseline --recall-queries 512 2>gpurun_out/r3_def_err.log | show "proxy4" || tail -5 gpurun_out/r3_def_err.log
for w in c2 c4; do timeout -k 10 400 python3 bench.py --workload $w --through-pipeline --steps 5 2>gpurun_out/r3_def_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   pipeline $w', [(r['queries_per_call'], r['queries_per_s']) for r in d['config']['results']])" || tail -5 gpurun_out/r3_def_err.log; done
python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests.log 2>&1; tail -3 gpurun_out/r3_gpu_tests.log
bash tools/r3_rehearse.sh
