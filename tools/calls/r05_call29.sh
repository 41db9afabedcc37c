#!/bin/bash
# round 5, call 29: records on HEAD -- instruction-level PMC of all seven workloads (the Radau kernel is new), rocprofv3 kernel
# stats of the cold C3 with the Radau pass (restart, continue), the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O/profiles
timeout -k 10 900 bash tools/pmc_issue.sh r05 > $O/pmc_issue_r05.log 2>&1; echo "pmc rc=$?"
cp profiles/r05_pmc_issue.json $O/profiles/
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline --tmin -5 --group-tiles 8 --radau"
( cd /tmp && export TMPDIR=/tmp
  for T in restart continue; do
    X=""; [ $T = continue ] && X="--radau-continue"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r05_cold_$T" -- python3 "$GRAFT_REPO_ROOT/bench.py" $B $X > "$GRAFT_REPO_ROOT/$O/bench_cold_$T.json" 2> "$GRAFT_REPO_ROOT/$O/bench_cold_$T.err" || echo "trace $T failed"
    for f in "$GRAFT_REPO_ROOT"/gpurun_out/prof_r05_cold_$T/*/*kernel_stats.csv; do head -10 "$f" > "$GRAFT_REPO_ROOT/$O/profiles/r05_cold_radau_${T}_kernel_stats.csv"; done
  done )
head -4 $O/profiles/r05_cold_radau_restart_kernel_stats.csv | cut -c1-220
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_line.json").read().strip().splitlines()[-1])
c = d.get("variant_survey_c3_cold_cells", {})
print(d["ms_per_step"], {k: (round(v["ms"], 2) if isinstance(v, dict) and "ms" in v else None) for k, v in c.items()})
print(json.dumps(c.get("rk45_plus_radau", {}).get("radau_kernel"))[:400])
PY
