#!/bin/bash
# the config-5 share by window length: is the per-launch cost a fixed fill + drain?
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for w in 1440 2880 4320 7200; do
  timeout -k 10 300 python bench.py --no-variants --no-cpu-baseline --steps 1 --warmup 0 --dense-ns 65536 --dense-hbm-ns 0 --c5-window $w 2>/dev/null | python -c "
import sys,json
c=json.loads(sys.stdin.read().strip().splitlines()[-1])['dense_output']['c5_share']
k=c['kernel_ms_each']
print('window', $w, 'launches', c['windows'], 'kernel ms sum %.1f' % c['kernel_ms_sum'], 'frac %.4f' % c['frac_of_hbm_spec'], 'first/mid/last ms', round(k[0],2), round(k[len(k)//2],2), round(k[-1],2))
" || exit 1
done
