#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c27
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "dense or streamed or selected or c5 or toy or radau or manual_window" > gpurun_out/c27/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/c27/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-variants --no-cpu-baseline --steps 1 --warmup 0 --dense-ns 65536 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['dense_output']
print('in_hbm', d['in_hbm_kernel']['kernel_ms'], d['in_hbm_kernel']['roofline']['frac'], 'c5', d['c5_share']['kernel_ms_sum'], d['c5_share']['frac_of_hbm_spec'])"
