#!/bin/bash
# round 4, call 31: STRICT powers from csrc/hlm_strict_pow.h -- the math self-tests, the bench's STRICT variants, the
# whole GPU suite (no -x: which labels' exempt sets moved), then a recording run of the suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c31
timeout -k 10 300 python -m pytest tests -m gpu -q -k "elementary_functions or single_evaluations" > gpurun_out/c31/math.log 2>&1; echo "math rc=$?"; tail -3 gpurun_out/c31/math.log
timeout -k 10 600 python bench.py --no-dense --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/c31/bench_line.json 2> gpurun_out/c31/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c31/bench_line.json').read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], 'wet', d['variant_wet_surface']['kernel_ms'], d['variant_wet_surface'].get('kernel_ms_queue_run_pieces_1'),
      'strict', d['variant_strict_math']['kernel_ms'], 'strict wet', d['variant_strict_math_wet_surface']['kernel_ms'])
PY
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c31/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/c31/pytest.log
HLM_PARITY_RECORD=$PWD/gpurun_out/c31/exempt_sets.json timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c31/pytest_record.log 2>&1; echo "record rc=$?"; tail -3 gpurun_out/c31/pytest_record.log
