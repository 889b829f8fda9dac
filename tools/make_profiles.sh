#!/bin/bash
# Collect the round's evidence on the GPU box (one gpurun call): bench lines for every workload, the
# rocprofv3 kernel-trace stats of the default bench command, and separate --pmc passes (FETCH_SIZE,
# WRITE_SIZE) for c2 and c4.  Outputs land in gpurun_out/<TAG>_*; tools/pmc_summary.py turns them into
# profiles/.   usage: tools/make_profiles.sh TAG
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in c2 c3 c4 c5; do
  extra=""; [ $w != c2 ] && extra="--no-cpu-baseline"
  python3 bench.py --workload $w $extra > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err || { tail -5 gpurun_out/${TAG}_bench_$w.err; exit 1; }
  echo "bench $w done: $(cut -c1-140 gpurun_out/${TAG}_bench_$w.json)"
done
# kernel-trace stats of the same (default) command; program directly after --
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_c2 -- python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_stats_c2.log 2>&1 || { tail -5 gpurun_out/${TAG}_stats_c2.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_c4 -- python3 bench.py --workload c4 --no-cpu-baseline > gpurun_out/${TAG}_stats_c4.log 2>&1 || exit 1
echo "stats done"
# PMC passes: counters only with --kernel-trace (one --pmc set per run)
for w in c2 c4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_${c}_$w -- python3 bench.py --workload $w --steps 20 --warmup 2 --streams 1 --no-graph --no-cpu-baseline > gpurun_out/${TAG}_pmc_${c}_$w.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc_${c}_$w.log; exit 1; }
  done
  echo "pmc $w done"
done
