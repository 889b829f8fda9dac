#!/bin/bash
cd "$(dirname "$0")/.."
for v in 0 1; do
  echo "== CRS_GEMM8_VAR=$v  (0: persistent k-tile stream, 1: one workgroup per tile)"
  CRS_GEMM8_VAR=$v timeout -k 10 200 python tools/bench_gemm.py 4096 4096 4096 0 8192 8192 8192 0 32768 2304 768 0 32768 3072 768 1 32768 768 3072 2 32768 768 768 2 65536 1536 384 1 65536 1024 384 0 4096 2304 768 0 2>&1 | grep -v "amdgpu.ids\|CRS_GEMM_BIG"
done
python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "gemm_vs_torch or bge or index_build" 2>&1 | tail -3
for w in enc-bge enc-minilm; do
  timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
