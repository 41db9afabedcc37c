#!/bin/bash
# round 5, call 7: the saturated-regime tests once more (attempt-count and slope-jump asserts), recording; then the whole GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export HLM_PARITY_RECORD=$PWD/$O/exempt_sets_new.json
rm -f $HLM_PARITY_RECORD
timeout -k 10 1100 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -s -k "saturating or counting or group_passes" > $O/pytest_new5.log 2>&1; echo "pytest new rc=$?"; grep -v amdgpu.ids $O/pytest_new5.log | grep "no-dense\|shares\|passed\|failed\|Error\|one attempt" | cut -c1-900
unset HLM_PARITY_RECORD
python - <<'PY'
import json
new = json.load(open("gpurun_out/r05/exempt_sets_new.json"))
cur = json.load(open("tests/golden/parity_exempt_sets.json"))
for k, v in new.items():
    if isinstance(v, list):
        cur[k] = v
json.dump(cur, open("tests/golden/parity_exempt_sets.json", "w"), indent=0, sort_keys=True)
PY
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -5 $O/pytest_all.log | cut -c1-300
