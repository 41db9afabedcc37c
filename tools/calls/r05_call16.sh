#!/bin/bash
# round 5, call 16: rocprofv3 record of the headline kernel (trace + stats, separate PMC passes) and of the dense kernel,
# then the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O/profiles
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; echo "profile_round rc=$?"
python3 tools/summarize_profile.py gpurun_out/prof_r05 r05 > $O/summarize.log 2>&1; echo "summarize rc=$?"; tail -3 $O/summarize.log | cut -c1-300
bash tools/profile_dense.sh r05_dense > $O/profile_dense.log 2>&1; echo "profile_dense rc=$?"
cp gpurun_out/prof_r05_dense/summary.json $O/profiles/r05_dense_pmc.json
cp profiles/r05_kernel_stats.csv profiles/r05_pmc.json profiles/traffic_latest.json $O/profiles/ 2>/dev/null
for f in gpurun_out/prof_r05_dense/trace/*/*kernel_stats.csv; do head -8 "$f" > $O/profiles/r05_dense_kernel_stats.csv; done
timeout -k 10 1000 python bench.py > $O/bench_c.json 2> $O/bench_c.err; echo "bench rc=$?"; cut -c1-600 $O/bench_c.json
