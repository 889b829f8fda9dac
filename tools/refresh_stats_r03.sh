#!/bin/bash
# kernel-trace stats + per-lane timelines of c4 / c3 / c5 / c2 with the final code (bench lines: tools/refresh_lines_r03.sh)
TAG=r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O/${TAG}_profiles
for w in c4 c3 c5 c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_$w -- python3 bench.py --workload $w --no-cpu-baseline --recall-queries 512 > $O/${TAG}_stats_$w.log 2>&1 || { tail -5 $O/${TAG}_stats_$w.log; exit 1; }
  python3 tools/timeline.py $O/${TAG}_stats_$w 2 > $O/${TAG}_profiles/${TAG}_timeline_$w.txt 2>&1 || true
done
python3 tools/collect_profiles_r03.py $O/${TAG}_profiles > $O/${TAG}_collect.log 2>&1 || tail -5 $O/${TAG}_collect.log
rm -rf $O/${TAG}_stats_*
ls $O/${TAG}_profiles
