#!/bin/bash
# kernel timeline of the default C4 line (and of its variants): which lane waits for what
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for v in "" "--no-refine" "--scan-only" "--rows 1250000 --proxy-encode-shard 8"; do
  t=$(echo "tl$v" | tr -d ' -' | cut -c1-24)
  rocprofv3 --kernel-trace --output-format csv -d $O/$t -- python3 bench.py --no-cpu-baseline --recall-queries 64 --steps 40 $v > $O/$t.log 2>&1 || { tail -5 $O/$t.log; exit 1; }
  echo "=== c4 $v"; grep '^{' $O/$t.log | cut -c1-120
  python3 tools/timeline.py $O/$t 2
  rm -rf $O/$t
done
