#!/bin/bash
# A/B of bench.py's stream layout in ONE gpurun call: --lanes batch (every in-flight batch wholly on its own stream)
# against --lanes split (encoder lanes + search lanes; sets CRS_PANEL_KC=128 CRS_ENC_QKVATTN=0 so that the encoder's
# kernels fit beside the scan's workgroups).  Shard sizes 5 M / 2.5 M / 1.25 M rows = one rank of a 2 / 4 / 8-GPU C4 step.
show() { grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('   %-46s %9.1f q/s  batch %.4f ms  lanes=%s  ok=%s' % (sys.argv[1], d['value'], c['ms_per_batch'], c['lanes'], c['check_ok']))" "$1"; }
for rep in 1 2 3; do
  for l in batch split "split --enc-lanes 1 --search-lanes 2"; do python3 bench.py --no-cpu-baseline --lanes $l 2>/dev/null | show "c4 $l (rep $rep)"; done
done
for w in c5 c3 c2; do
  for l in batch split; do python3 bench.py --workload $w --no-cpu-baseline --lanes $l 2>/dev/null | show "$w $l"; done
done
for rows in 5000000 2500000; do
  for l in batch split "split --enc-lanes 1 --search-lanes 2"; do python3 bench.py --rows $rows --no-cpu-baseline --lanes $l 2>/dev/null | show "$rows rows $l"; done
done
for l in batch split "split --enc-lanes 1 --search-lanes 2"; do
  python3 bench.py --rows 1250000 --no-cpu-baseline --proxy-encode-shard 8 --lanes $l 2>/dev/null | show "1250000 rows, 8 queries encoded, $l"
done
