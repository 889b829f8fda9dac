#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print('   %-58s %9.1f q/s  batch %.4f ms  grp %s sets %s lanes %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], c['batches_per_encoder_forward'], c['batches_per_step'], c['lanes'][:22], c['check_ok']))" "$1"; }
# two ranks on the one card over gloo: the grouped + sharded-encode layout (the N >= 4 default) end to end
CRS_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --workload c4 --rows 3000000 --encode sharded --recall-queries 1024 2>gpurun_out/r3_multi_err.log | show "2 ranks, 1.5 M rows each, sharded encode" || tail -8 gpurun_out/r3_multi_err.log
CRS_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --workload c4 --rows 3000000 --recall-queries 1024 2>gpurun_out/r3_multi_err.log | show "2 ranks, 1.5 M rows each, replicated encode" || tail -8 gpurun_out/r3_multi_err.log
for a in "--rows 1250000 --proxy-encode-shard 8" "--rows 2500000 --proxy-encode-shard 4" "--rows 2500000 --proxy-encode-shard 4 --encode-group 1 --streams 8 --enc-lanes 2 --search-lanes 1" "--rows 1250000" "--rows 1250000 --encode-group 1 --streams 8 --enc-lanes 2 --search-lanes 1" ""; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --recall-queries 512 $a 2>gpurun_out/r3_multi_err.log | show "$a" || tail -3 gpurun_out/r3_multi_err.log
done
