#!/bin/bash
# Instruction-level PMC of the headline kernel (round-3 judge item 2): tools/pmc_issue.sh <tag>
# tools/pmc_issue.sh <tag> ["workload ..."]: headline / wet surface / the saturating capacities / SURVEY's active-surface draw / SURVEY's cold cells (+ Radau), two SQ passes each (8 SQ slots + GRBM per pass,
# /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots"); never combined with tracing; the program itself
# after "--".  tools/summarize_pmc_issue.py condenses the output into profiles/<tag>_pmc_issue.json.
ROOTDIR="${GRAFT_REPO_ROOT:-$PWD}"
TAG=${1:-r05}
WORKLOADS=${2:-"headline wet saturating active_surface cold cold_radau cold_radau_continue"}
OUT="$ROOTDIR/gpurun_out/pmc_issue_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BASE="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline"
PASS_A="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
PASS_B="SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
for W in $WORKLOADS; do
  case $W in
    headline) EXTRA="";;
    wet) EXTRA="--wet-surface";;
    saturating) EXTRA="--hu saturating --attempt-cap 300000";;
    active_surface) EXTRA="--hu active-surface --attempt-cap 300000";;
    cold) EXTRA="--tmin -5 --group-tiles 8";;
    cold_radau) EXTRA="--tmin -5 --radau --group-tiles 8";;
    cold_radau_continue) EXTRA="--tmin -5 --radau --radau-continue --group-tiles 8";;
  esac
  for P in A B; do
    if [ $P = A ]; then C="$PASS_A"; else C="$PASS_B"; fi
    rocprofv3 --pmc $C --output-format csv -d "$OUT/${W}_$P" -- python3 "$ROOTDIR/bench.py" $BASE $EXTRA > "$OUT/bench_${W}_$P.json" 2> "$OUT/err_${W}_$P.txt" || echo "pass $W $P failed"
    echo "done $W $P"
  done
done
python3 "$ROOTDIR/tools/summarize_pmc_issue.py" "$OUT" "$TAG"
