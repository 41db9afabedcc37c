#!/bin/bash
# in-HBM dense kernel, 1 M hillslopes x 1 day at 1-minute stride (57.6 GB of rows), every device layout
cd "$(dirname "$0")/.."
for lay in ${1:-0 1 2 3}; do
  python bench.py --days 1 --dense-stride 1 --dense-tiled $lay --steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('layout $lay: kernel ms %.2f  %.0f GB/s  frac %.3f  attempts/hillslope %.1f' % (r['kernel_ms'], r['achieved'], r['frac'], d['attempts_per_hillslope']))
"
done
