#!/bin/bash
# Collect the round's evidence on the GPU box (one gpurun call): bench lines for every workload, the
# rocprofv3 kernel-trace stats of the same commands, and separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
# for the default workload.  Outputs land in gpurun_out/<TAG>_*; tools/pmc_summary.py turns them into
# profiles/.   usage: tools/make_profiles.sh TAG [workloads...]
set -o pipefail
TAG=${1:-r02}; shift
WL=${@:-c4 c5 c3 c2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in $WL; do
  extra=""; [ $w != c4 ] && extra="--no-cpu-baseline"
  python3 bench.py --workload $w $extra > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err || { tail -5 gpurun_out/${TAG}_bench_$w.err; exit 1; }
  echo "bench $w done: $(cut -c1-200 gpurun_out/${TAG}_bench_$w.json)"
  # kernel-trace stats of the same command; program directly after --
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_$w -- python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/${TAG}_stats_$w.log 2>&1 || { tail -5 gpurun_out/${TAG}_stats_$w.log; exit 1; }
  echo "stats $w done"
done
# encoder index-build workloads (bench line + kernel stats of the same command)
for w in enc-minilm enc-bge; do
  python3 bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err || { tail -5 gpurun_out/${TAG}_bench_$w.err; exit 1; }
  echo "bench $w done: $(cut -c100-240 gpurun_out/${TAG}_bench_$w.json)"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_$w -- python3 bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/${TAG}_stats_$w.log 2>&1 || { tail -5 gpurun_out/${TAG}_stats_$w.log; exit 1; }
  echo "stats $w done"
done
# PMC passes: counters only with --kernel-trace (one --pmc set per run)
for w in c4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc_${c}_$w -- python3 bench.py --workload $w --steps 6 --warmup 1 --streams 1 --no-graph --no-cpu-baseline > gpurun_out/${TAG}_pmc_${c}_$w.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc_${c}_$w.log; exit 1; }
  done
  echo "pmc $w done"
done
