cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in enc-minilm enc-bge; do
  python3 bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/r02b_bench_$w.json 2>gpurun_out/r02b_bench_$w.err || tail -5 gpurun_out/r02b_bench_$w.err
  cut -c1-400 gpurun_out/r02b_bench_$w.json
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02b_stats_$w -- python3 bench.py --workload $w --steps 20 --warmup 3 > gpurun_out/r02b_stats_$w.log 2>&1 || tail -5 gpurun_out/r02b_stats_$w.log
done
