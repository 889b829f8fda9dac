#!/bin/bash
# where does C4's batch time go beyond the search segment?  (one call, same box)
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; q=c['recall_at_10_vs_fp32']; print('   %-44s %9.1f q/s  batch %.4f ms  seg_in_run %s  kern %.4f tot %.4f cert %s  k_scan %d certified %s esc %s ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], r['search_segment_ms_in_run'], r['kernel_ms'], r['scan_merge_refine_ms'], r['fp32_rerank_cert_ms'], c['k_scan'], q['certified_frac'], q['escalated'], c['check_ok']))" "$1"; }
B="python3 bench.py --no-cpu-baseline"
for a in "" "--k-scan 24" "--k-scan 16" "--scan-only" "--lanes batch" "--streams 4" "--exact off" "--no-refine"; do
  $B $a 2>/dev/null | show "c4 $a"
done
python3 bench.py --workload c3 --no-cpu-baseline --recall-queries 2048 2>/dev/null | show "c3"
# FETCH_SIZE pass for c4 (pmc summary)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r03_pmc_${c}_c4 -- python3 bench.py --workload c4 --steps 6 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --recall-queries 64 > gpurun_out/r03_pmc_${c}_c4.log 2>&1 || tail -5 gpurun_out/r03_pmc_${c}_c4.log
done
python3 bench.py --no-cpu-baseline --recall-queries 64 > gpurun_out/r03_bench_c4.json 2>/dev/null
python3 tools/collect_profiles_r03.py gpurun_out/r03_profiles2 > gpurun_out/r03_collect2.log 2>&1; cat gpurun_out/r03_profiles2/r03_pmc_summary.json | tail -15
rm -rf gpurun_out/r03_pmc_*
