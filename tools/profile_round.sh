#!/bin/bash
# rocprofv3 recipe for the round's profiles (run on the GPU box from the repo root):
#   1. kernel trace + stats of the exact bench.py command
#   2. separate PMC passes (FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE) -- never combined
#      with tracing (guide: MI355X_MICROARCH.md "rocprofv3 PMC slots")
set -e
ROOTDIR="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$ROOTDIR/gpurun_out/prof_${1:-r02}"
ARGS="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline ${2:-}"   # (--no-dense: the 57.6 GB dense section is profiled by itself, tools/profile_dense.sh)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_grbm" -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/bench_grbm.json" 2> "$OUT/grbm.err"
find "$OUT" -name "*.csv" | head -50
