#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export HLM_PROBE_PIECES=auto,4,8,16
timeout -k 10 400 python tools/streamed_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "chunk   5760\|chunk  11520\|chunk  -2880" | cut -c1-300
