#!/bin/bash
# round 5, call 22: Radau pass -- piece counts in LDS, the breaks of the state machine batched: tests, then A/B of the batch size and the pieces
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_output_selection_gpu.py -m gpu -q -x -k "radau" > $O/pytest_radau.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_radau.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline --tmin -5 --group-tiles 8"
run() {  # tag, env..., args
  local tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py $B --radau $EXTRA > $O/radau_$tag.json 2> $O/radau_$tag.err || return 1
  python -c "
import json; d=json.loads(open('$O/radau_$tag.json').read().strip().splitlines()[-1]); print('$tag: ms_per_step', round(d['ms_per_step'],2))"
}
V=$PWD/tiger_hlm_gpu_amd/lib/variants
EXTRA=""
run restart_default X=1 || exit 1
run restart_p8 HLM_RADAU_PIECES=8 || exit 1
run restart_p1 HLM_RADAU_PIECES=1 || exit 1
run restart_b1 HLM_HIP_LIB=$V/libhlm_b1.so || exit 1
run restart_b2 HLM_HIP_LIB=$V/libhlm_b2.so || exit 1
run restart_b8 HLM_HIP_LIB=$V/libhlm_b8.so || exit 1
EXTRA="--radau-continue"
run cont_default X=1 || exit 1
run cont_b1 HLM_HIP_LIB=$V/libhlm_b1.so || exit 1
run cont_b2 HLM_HIP_LIB=$V/libhlm_b2.so || exit 1
run cont_b8 HLM_HIP_LIB=$V/libhlm_b8.so || exit 1
