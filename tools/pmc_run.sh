#!/bin/bash
# tools/pmc_run.sh TAG "COUNTERS..." -- bench args     (one --pmc pass, kernel trace only)
TAG=$1; CNT=$2; shift 3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d gpurun_out/$TAG -- python3 bench.py "$@" --steps 10 --warmup 2 --streams 1 --no-graph --no-cpu-baseline > gpurun_out/$TAG.log 2>&1
python3 tools/pmc_parse.py gpurun_out/$TAG
