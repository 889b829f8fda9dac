#!/bin/bash
# the C4 part of tools/make_profiles_r03.sh alone (after a change of the default layout)
set -o pipefail
TAG=r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O/${TAG}_profiles
python3 bench.py > $O/${TAG}_bench_c4.json 2> $O/${TAG}_bench_c4.err || { tail -5 $O/${TAG}_bench_c4.err; exit 1; }
python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline > $O/${TAG}_bench_proxy8.json 2> $O/${TAG}_bench_proxy8.err || tail -3 $O/${TAG}_bench_proxy8.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_c4 -- python3 bench.py --workload c4 --no-cpu-baseline --recall-queries 512 > $O/${TAG}_stats_c4.log 2>&1 || { tail -5 $O/${TAG}_stats_c4.log; exit 1; }
python3 tools/timeline.py $O/${TAG}_stats_c4 2 > $O/${TAG}_profiles/${TAG}_timeline_c4.txt 2>&1 || true
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_${c}_c4 -- python3 bench.py --workload c4 --steps 6 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --recall-queries 64 > $O/${TAG}_pmc_${c}_c4.log 2>&1 || { tail -5 $O/${TAG}_pmc_${c}_c4.log; exit 1; }
done
bash tools/sq_pmc.sh $TAG c4 > $O/${TAG}_sq_c4.log 2>&1 || tail -3 $O/${TAG}_sq_c4.log
python3 tools/collect_profiles_r03.py $O/${TAG}_profiles > $O/${TAG}_collect.log 2>&1 || tail -5 $O/${TAG}_collect.log
rm -rf $O/${TAG}_stats_* $O/${TAG}_pmc_* $O/${TAG}_sq[12]_*
ls $O/${TAG}_profiles
