#!/bin/bash
cd "$(dirname "$0")/.."
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1', d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'])"; }
for f in 1 2 3; do for w in enc-minilm enc-bge; do
python3 bench.py --workload $w --steps 20 --warmup 3 --enc-inflight $f 2>/dev/null | show "inflight=$f $w"
done; done
