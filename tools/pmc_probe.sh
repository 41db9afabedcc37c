#!/bin/bash
# ad-hoc PMC probe: tools/pmc_probe.sh <tag> "<bench args>" "<counters pass1>" ["<counters pass2>" ...]
ROOTDIR="${GRAFT_REPO_ROOT:-$PWD}"
TAG=$1; ARGS=$2; shift 2
OUT="$ROOTDIR/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/p$i" -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/bench_$i.json" 2> "$OUT/err_$i.txt"
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for f in sorted(glob.glob(out+'/p*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = 'radau' if 'radau_fallback' in r['Kernel_Name'] else ('rk45' if 'rk45' in r['Kernel_Name'] else None)
        if k: agg[k + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(f.split('/')[-3], k, sum(v)/len(v), len(v))
PY
