#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c29
timeout -k 5 200 python tools/queue_hang_probe.py gpurun_out/c29/probe.txt; rc=$?; echo "probe rc=$rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "time_pieces or radau_continue" > gpurun_out/c29/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/c29/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/pieces_probe.py > gpurun_out/c29/pieces.txt 2>&1; echo "pieces rc=$?"; tail -25 gpurun_out/c29/pieces.txt
timeout -k 10 500 tools/profile_round.sh r03 && python tools/summarize_profile.py gpurun_out/prof_r03 r03 > gpurun_out/c29/summary.json; echo "profile rc=$?"; grep "avg_duration\|hbm_\|effective" gpurun_out/c29/summary.json
