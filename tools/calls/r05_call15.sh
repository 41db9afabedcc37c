#!/bin/bash
# round 5, call 15: Radau continue against the oracle's pass from the GPU's flag states; wide ranges with the distribution guard;
# the automatic sub-windows test; negative capacities at the tighter bound
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -s -k "radau_continue or wide_parameter or automatic_dense_sub or negative_capacities" > $O/pytest_misc.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/pytest_misc.log | grep "radau continue\|wide ranges\|negative cap\|passed\|failed\|Error" | cut -c1-700
