#!/bin/bash
# round 4, call 49: rocprofv3 record of the dense launch as (sub-window, tile) items at config 5's share size: 500 000
# hillslopes x 1 day of 1-minute queries, 4 sub-windows (rk45_dense_queue_kernel): kernel trace, WRITE_SIZE, FETCH_SIZE
set -o pipefail
ROOTDIR="$GRAFT_REPO_ROOT"
OUT="$ROOTDIR/gpurun_out/prof_r04_dense_queue"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOTDIR/tools/dense_subwindow_probe.py" --ns 500000 --sub 4 > "$OUT/trace.out" 2> "$OUT/trace.err"; echo "trace rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOTDIR/tools/dense_subwindow_probe.py" --ns 500000 --sub 4 > "$OUT/write.out" 2> "$OUT/write.err"; echo "write rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOTDIR/tools/dense_subwindow_probe.py" --ns 500000 --sub 4 > "$OUT/fetch.out" 2> "$OUT/fetch.err"; echo "fetch rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(out + '/trace/*/*kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        if 'rk45_dense_queue' in r['Name']:
            res['kernel_stats'] = {k: r[k] for k in r}
for f in sorted(glob.glob(out + '/pmc_*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'rk45_dense_queue' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        res[k + '_KiB_per_launch'] = sum(v) / len(v)
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
