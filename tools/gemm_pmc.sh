#!/bin/bash
# PMC passes over the index-build GEMM shapes of one model (tools/bench_gemm.py M): counters only with --kernel-trace.
# usage: tools/gemm_pmc.sh TAG M      -> gpurun_out/TAG_pmc<i>/..._counter_collection.csv, summarised per kernel on stdout
set -o pipefail
TAG=${1:-gp}; M=${2:-65536}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc$i -- python3 tools/bench_gemm.py $M > gpurun_out/${TAG}_pmc$i.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc$i.log; exit 1; }
  f=$(ls -t gpurun_out/${TAG}_pmc$i/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "crs" not in k: continue
    acc[k[22:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
