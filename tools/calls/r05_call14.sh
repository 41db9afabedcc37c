#!/bin/bash
# round 5, call 14: hybrid waves at two waves per SIMD (no spills): time on the four workloads
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline"
for h in 1 -1; do
  for w in "" "--hu saturating --attempt-cap 300000" "--hu saturating --attempt-cap 300000 --group-tiles 8" "--wet-surface"; do
    timeout -k 10 200 python bench.py $B --hybrid $h $w 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hybrid', '$h', '[$w]', 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'flagged %.4f' % d['flagged_fraction'], 'attempts', int(d['attempts_per_pass']))"
  done
done | tee $O/hybrid_ab2.txt
