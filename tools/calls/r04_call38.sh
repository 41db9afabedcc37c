#!/bin/bash
# round 4, call 38: the math self-test, then a recording run (counts + index sets) of the parity tests whose STRICT
# exemptions move with the correctly rounded powers
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c38
timeout -k 10 300 python -m pytest tests -m gpu -q -k "elementary_functions" > gpurun_out/c38/math.log 2>&1; echo "math rc=$?"; tail -3 gpurun_out/c38/math.log
HLM_PARITY_RECORD=$PWD/gpurun_out/c38/exempt_sets.json timeout -k 10 900 python -m pytest tests -m gpu -q -k "fuzz_parity or wide_parameter or toy_model_three or no_dense_launches_vs_oracle or cold" > gpurun_out/c38/pytest_record.log 2>&1; echo "record rc=$?"; tail -8 gpurun_out/c38/pytest_record.log
