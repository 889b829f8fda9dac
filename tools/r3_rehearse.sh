#!/bin/bash
# the driver's own N = 2 / N = 4 command lines (full 10 M-row corpus, default flags), ranks sharing the one card over gloo:
# functional rehearsal of the default layouts (groups, lanes, group all-gather) -- the gloo rates mean nothing
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('   %-10s %9.1f q/s  batch %.4f ms  grp %s sets %s lanes %s encode %s collectives/batch %s recall %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], c['batches_per_encoder_forward'], c['batches_per_step'], c['lanes'][:21], c['query_encode'][:24], c['collectives_per_batch'], c['recall_at_10_vs_fp32']['timed_path'], c['check_ok']))" "$1"; }
for n in 2 4; do
CRS_DIST_BACKEND=gloo timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29560 + n)) bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --recall-queries 1024 2>gpurun_out/r3_rehearse_err.log | show "N=$n" || tail -12 gpurun_out/r3_rehearse_err.log
done
