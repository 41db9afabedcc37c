#!/bin/bash
# round 5, call 4: the saturating tests with the second tier sampling both sides, recording
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export HLM_PARITY_RECORD=$PWD/$O/exempt_sets_new.json
rm -f $HLM_PARITY_RECORD
timeout -k 10 1100 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -s -k "saturating or counting or group_passes" > $O/pytest_new3.log 2>&1; echo "pytest new rc=$?"; grep -v amdgpu.ids $O/pytest_new3.log | grep "third tier\|no-dense\|shares\|passed\|failed\|Error" | cut -c1-600
