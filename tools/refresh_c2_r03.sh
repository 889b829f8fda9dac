#!/bin/bash
set -o pipefail
TAG=r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O/${TAG}_profiles
python3 bench.py --workload c2 --no-cpu-baseline > $O/${TAG}_bench_c2.json 2> $O/${TAG}_bench_c2.err || { tail -5 $O/${TAG}_bench_c2.err; exit 1; }
python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline > $O/${TAG}_bench_proxy8.json 2> $O/${TAG}_bench_proxy8.err || tail -3 $O/${TAG}_bench_proxy8.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_c2 -- python3 bench.py --workload c2 --no-cpu-baseline --recall-queries 512 > $O/${TAG}_stats_c2.log 2>&1 || { tail -5 $O/${TAG}_stats_c2.log; exit 1; }
python3 tools/timeline.py $O/${TAG}_stats_c2 2 > $O/${TAG}_profiles/${TAG}_timeline_c2.txt 2>&1 || true
bash tools/r3_multi.sh > $O/${TAG}_profiles/${TAG}_layouts.txt 2>&1 || true
python3 tools/collect_profiles_r03.py $O/${TAG}_profiles > $O/${TAG}_collect.log 2>&1 || tail -5 $O/${TAG}_collect.log
rm -rf $O/${TAG}_stats_*
ls $O/${TAG}_profiles
