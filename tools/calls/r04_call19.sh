#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c19
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "group_tiles or time_pieces or radau" > gpurun_out/c19/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/c19/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/radau_machine_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
HLM_PROBE_PIECES=64 timeout -k 10 200 python tools/radau_machine_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
