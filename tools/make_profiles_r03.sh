#!/bin/bash
# Round-3 evidence in ONE gpurun call (outputs under gpurun_out/r03_*; tools/collect_profiles_r03.py copies the summaries to profiles/):
#   bench lines (c4 default incl. cpu_baseline, c5, c3, c2, enc-bge, enc-minilm, --through-pipeline c2 / c4, the 8-GPU rank proxy),
#   rocprofv3 --kernel-trace --stats of the same commands, FETCH_SIZE / WRITE_SIZE passes for c4, SQ counters for c4 / c5 / c3 and for the GEMMs.
set -o pipefail
TAG=r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
python3 bench.py > $O/${TAG}_bench_c4.json 2> $O/${TAG}_bench_c4.err || { tail -5 $O/${TAG}_bench_c4.err; exit 1; }
echo "bench c4: $(cut -c1-160 $O/${TAG}_bench_c4.json)"
for w in c5 c3 c2; do
  python3 bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || { tail -5 $O/${TAG}_bench_$w.err; exit 1; }
  echo "bench $w: $(cut -c100-200 $O/${TAG}_bench_$w.json)"
done
for w in enc-minilm enc-bge; do
  python3 bench.py --workload $w --steps 20 --warmup 3 > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err || { tail -5 $O/${TAG}_bench_$w.err; exit 1; }
  echo "bench $w: $(cut -c100-200 $O/${TAG}_bench_$w.json)"
done
python3 bench.py --rows 1250000 --proxy-encode-shard 8 --no-cpu-baseline > $O/${TAG}_bench_proxy8.json 2> $O/${TAG}_bench_proxy8.err || tail -3 $O/${TAG}_bench_proxy8.err
python3 bench.py --workload c2 --through-pipeline --steps 7 > $O/${TAG}_bench_pipeline_c2.json 2> $O/${TAG}_bench_pipeline_c2.err || tail -3 $O/${TAG}_bench_pipeline_c2.err
python3 bench.py --workload c4 --through-pipeline --steps 7 > $O/${TAG}_bench_pipeline_c4.json 2> $O/${TAG}_bench_pipeline_c4.err || tail -3 $O/${TAG}_bench_pipeline_c4.err
echo "pipeline lines done"
# kernel-trace stats of the same commands (program directly after --)
for w in c4 c5 c3 c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_$w -- python3 bench.py --workload $w --no-cpu-baseline --recall-queries 512 > $O/${TAG}_stats_$w.log 2>&1 || { tail -5 $O/${TAG}_stats_$w.log; exit 1; }
done
for w in enc-minilm enc-bge; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_$w -- python3 bench.py --workload $w --steps 20 --warmup 3 > $O/${TAG}_stats_$w.log 2>&1 || { tail -5 $O/${TAG}_stats_$w.log; exit 1; }
done
# per-queue timelines of the same traces (which lane waits for what): C4 (2 encoder + 1 search lanes), C3 / C5 (encode groups), and
# the dynamic tile schedule's own check (identical lists, the counter's value, time per setting)
mkdir -p $O/${TAG}_profiles
for w in c4 c3 c5; do python3 tools/timeline.py $O/${TAG}_stats_$w 2 > $O/${TAG}_profiles/${TAG}_timeline_$w.txt 2>&1 || true; done
python3 tools/tb_dyn_check.py 10000000 > $O/${TAG}_profiles/${TAG}_tb_dyn_check.txt 2>&1 || true
bash tools/r3_multi.sh > $O/${TAG}_profiles/${TAG}_layouts.txt 2>&1 || true
echo "stats done"
# PMC passes: counters only with --kernel-trace (one --pmc set per run)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_pmc_${c}_c4 -- python3 bench.py --workload c4 --steps 6 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --recall-queries 64 > $O/${TAG}_pmc_${c}_c4.log 2>&1 || { tail -5 $O/${TAG}_pmc_${c}_c4.log; exit 1; }
done
echo "pmc c4 done"
for w in c5 c4 c3; do bash tools/sq_pmc.sh $TAG $w > $O/${TAG}_sq_$w.log 2>&1 || tail -3 $O/${TAG}_sq_$w.log; done
echo "sq done"
# SQ counters (matrix pipe busy, in-kernel clock) of the GEMM kernels on bge-base's and MiniLM's index-build shapes + 4096^3
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${TAG}_gemm_sq$i -- python3 tools/bench_gemm.py 4096 4096 4096 0 32768 2304 768 0 32768 3072 768 1 32768 768 3072 2 32768 768 768 2 65536 1536 384 1 65536 1152 384 0 65536 384 1536 2 > $O/${TAG}_gemm_sq$i.log 2>&1 || tail -3 $O/${TAG}_gemm_sq$i.log
done
python3 tools/bench_gemm.py 4096 4096 4096 0 32768 2304 768 0 32768 3072 768 1 32768 768 3072 2 32768 768 768 2 65536 1536 384 1 65536 1152 384 0 65536 384 1536 2 4096 2304 768 0 4096 3072 768 1 > $O/${TAG}_gemm_tflops.txt 2>&1
echo "gemm done"
# summarise ON the box (gpurun copies back at most 64 MiB: the raw traces stay here), keep logs + summaries only
python3 tools/collect_profiles_r03.py $O/${TAG}_profiles > $O/${TAG}_collect.log 2>&1 || tail -5 $O/${TAG}_collect.log
rm -rf $O/${TAG}_stats_* $O/${TAG}_pmc_* $O/${TAG}_sq[12]_* $O/${TAG}_gemm_sq[12] 
ls $O/${TAG}_profiles | head -50
