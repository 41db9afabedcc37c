#!/bin/bash
# round 4, call 39: HEAD with the correctly rounded STRICT powers and the capped run schedule: the whole GPU suite
# (no -x), the stiff-flags test in recording mode, smoke(), the default bench line, the round's rocprofv3 record
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c39
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c39/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/c39/pytest.log
HLM_PARITY_RECORD=$PWD/gpurun_out/c39/exempt_sets.json timeout -k 10 300 python -m pytest tests -m gpu -q -k "stiff_flags" > gpurun_out/c39/pytest_record.log 2>&1; echo "record rc=$?"; tail -2 gpurun_out/c39/pytest_record.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c39/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/c39/smoke.log
timeout -k 10 900 python bench.py > gpurun_out/c39/bench_line.json 2> gpurun_out/c39/bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/c39/bench_line.json; echo
bash tools/profile_round.sh r04 > gpurun_out/c39/profile.log 2>&1; echo "profile rc=$?"
