#!/bin/bash
# round 4, call 34: the capped automatic run schedule on three workloads; the STRICT math self-test against decimal
# references; a recording run of the parity tests whose STRICT exempt sets moved with the new powers
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c34
bash tools/calls/r04_call33.sh > gpurun_out/c34/capped.log 2>&1; echo "capped rc=$?"; cat gpurun_out/c33/capped_runs.txt
timeout -k 10 300 python -m pytest tests -m gpu -q -k "elementary_functions" > gpurun_out/c34/math.log 2>&1; echo "math rc=$?"; tail -3 gpurun_out/c34/math.log
HLM_PARITY_RECORD=$PWD/gpurun_out/c34/exempt_sets.json timeout -k 10 900 python -m pytest tests -m gpu -q -k "fuzz_parity or wide_parameter or toy_model_three or no_dense_launches_vs_oracle or cold" > gpurun_out/c34/pytest_record.log 2>&1; echo "record rc=$?"; tail -5 gpurun_out/c34/pytest_record.log
