#!/bin/bash
# the bench lines of every workload with the final defaults (no traces): profiles/r03_bench_*.json
TAG=r03
cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 bench.py > $O/${TAG}_bench_c4.json 2> $O/${TAG}_bench_c4.err || { tail -5 $O/${TAG}_bench_c4.err; exit 1; }
for w in c5 c3 c2; do
  python3 bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || { tail -5 $O/${TAG}_bench_$w.err; exit 1; }
done
for w in enc-minilm enc-bge; do
  python3 bench.py --workload $w --steps 20 --warmup 3 > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || { tail -5 $O/${TAG}_bench_$w.err; exit 1; }
done
python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline > $O/${TAG}_bench_proxy8.json 2> $O/${TAG}_bench_proxy8.err || tail -3 $O/${TAG}_bench_proxy8.err
python3 bench.py --workload c2 --through-pipeline --steps 7 > $O/${TAG}_bench_pipeline_c2.json 2> $O/${TAG}_bench_pipeline_c2.err || tail -3 $O/${TAG}_bench_pipeline_c2.err
python3 bench.py --workload c4 --through-pipeline --steps 7 > $O/${TAG}_bench_pipeline_c4.json 2> $O/${TAG}_bench_pipeline_c4.err || tail -3 $O/${TAG}_bench_pipeline_c4.err
for f in $O/${TAG}_bench_*.json; do python3 -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f'.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], r.get('frac'), (d.get('config') or {}).get('check_ok'))"; done
