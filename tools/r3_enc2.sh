#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in 0 1; do
  for w in enc-bge enc-minilm; do
    CRS_GEMM8_VAR=$v timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VAR=$v $w', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
done
CRS_GEMM8=0 timeout -k 10 200 python bench.py --workload enc-bge --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GEMM8=0 enc-bge', d['value'], d['ms_per_step'], d['roofline']['frac'])"
