#!/bin/bash
# round 5, call 13: hybrid waves -- bit equality with the per-form blocks, then time on the saturating and the headline workload
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -x -k "hybrid" > $O/pytest_hybrid.log 2>&1; echo "pytest hybrid rc=$?"; grep -v amdgpu.ids $O/pytest_hybrid.log | tail -6 | cut -c1-500
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline"
for h in -1 1; do
  for w in "" "--hu saturating --attempt-cap 300000" "--wet-surface" "--tmin -5 --group-tiles 8"; do
    timeout -k 10 200 python bench.py $B --hybrid $h $w 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hybrid', '$h', '[$w]', 'kernel_ms %.2f' % d['roofline']['kernel_ms'], 'flagged %.4f' % d['flagged_fraction'], 'attempts', int(d['attempts_per_pass']))"
  done
done | tee $O/hybrid_ab.txt
