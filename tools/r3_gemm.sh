#!/bin/bash
cd "$(dirname "$0")/.."
for v in 0 1 2 3; do
  echo "== CRS_GEMM8_VAR=$v"
  CRS_GEMM8_VAR=$v timeout -k 10 200 python tools/bench_gemm.py 4096 4096 4096 0 32768 2304 768 0 32768 3072 768 1 32768 768 3072 2 4096 2304 768 0 65536 1536 384 1 2>&1 | grep -v "amdgpu.ids\|CRS_GEMM_BIG"
done
