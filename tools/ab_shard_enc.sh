# what ONE rank of an N-GPU strong-scaling C4 step executes, on one GPU (no collectives): replicated vs sharded query encoding
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d['value'], 'q/s  ms/batch', d['config']['ms_per_batch'], 'scan', d['roofline']['kernel_ms'], d['config']['query_encode'], 'recall', d['config']['recall_at_10_vs_fp32']['timed_path'])"; }
for n in 8 4 2; do
  rows=$((10000000 / n))
  echo "N=$n shard ($rows rows): replicated"; run --rows $rows
  echo "N=$n shard: sharded proxy";           run --rows $rows --proxy-encode-shard $n
done
