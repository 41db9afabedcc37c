#!/bin/bash
# PC sampling of one bench.py configuration (development aid; run on the GPU box from the repo root):
#   tools/pcsample.sh <tag> <method: stochastic|host_trap> <interval> "<bench args>"
# writes gpurun_out/pcs_<tag>/ (rocprofv3 csv) and a per-instruction histogram pcs_<tag>/hist.txt
ROOTDIR="${GRAFT_REPO_ROOT:-$PWD}"
TAG=$1; METHOD=$2; INTERVAL=$3; ARGS=$4
OUT="$ROOTDIR/gpurun_out/pcs_$TAG"; mkdir -p "$OUT"
UNIT=cycles; [ "$METHOD" = host_trap ] && UNIT=time
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 10 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT \
    --pc-sampling-interval $INTERVAL --output-format csv -d "$OUT/raw" -- python3 "$ROOTDIR/bench.py" $ARGS \
    > "$OUT/bench.json" 2> "$OUT/err.txt"
rc=$?
echo "rocprofv3 pc sampling rc=$rc" | tee -a "$OUT/err.txt"
tail -5 "$OUT/err.txt"
find "$OUT/raw" -type f | head -20
python3 "$ROOTDIR/tools/pcsample_hist.py" "$OUT" || true
exit $rc
