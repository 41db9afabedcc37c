#!/bin/bash
# rocprofv3 record of the in-HBM dense kernel (1 M hillslopes x 1 day at 1-minute stride, tile layout), run on the GPU
# box from the repo root:  tools/profile_dense.sh <tag>
#   1. --kernel-trace --stats of the exact command (the program directly after "--")
#   2. separate PMC passes: WRITE_SIZE, FETCH_SIZE, clock / wave cycles, instruction mix -- never combined with tracing
set -e
ROOTDIR="${GRAFT_REPO_ROOT:-$PWD}"
TAG=${1:-r03_dense}
OUT="$ROOTDIR/gpurun_out/prof_$TAG"
ARGS="--days 1 --dense-stride 1 --dense-tiled 2 --steps 4 --warmup 1 --no-variants --no-dense --no-cpu-baseline"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
i=0
for C in "WRITE_SIZE" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(out + '/trace/*/*kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        if 'rk45' in r['Name']:
            res['kernel_stats'] = {k: r[k] for k in r}
for f in sorted(glob.glob(out + '/pmc*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'rk45' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        res[k] = sum(v) / len(v)
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
