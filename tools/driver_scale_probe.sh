#!/bin/bash
# the C++ driver end to end at scale: N hillslopes from a generated params CSV (rows of data/small_test.csv's
# format with varied Hu / slope / areas), 30 days, constant stub forcings, final.csv only.  Run on the GPU box.
N=${1:-1000000}
cd "$(dirname "$0")/.."
T=${TMPDIR:-/tmp}
python - "$N" "$T/big_params.csv" <<'PY'
import sys, numpy as np
n=int(sys.argv[1]); rng=np.random.default_rng(7)
hdr="stream,next_stream,drainage_area_km2,length_km,area_sqkm,centroid_lon,centroid_lat,hu,i2,i3,sw,ss,n,slope,res_ss,res_gw,melt,t_thres"
a=np.column_stack([np.arange(1,n+1), np.arange(2,n+2), rng.uniform(0.1,2,n), rng.uniform(0.2,1,n), rng.uniform(0.1,0.5,n),
   np.full(n,-75.0), np.full(n,40.3), rng.uniform(100,250,n), np.full(n,4.0), np.full(n,1.6), np.full(n,0.11), np.full(n,0.33),
   np.full(n,0.1), rng.uniform(0.01,0.05,n), np.full(n,2.0), np.full(n,55.0), np.full(n,3.7), np.zeros(n)])
np.savetxt(sys.argv[2], a, delimiter=",", header=hdr, comments="", fmt=["%d","%d"]+["%.6g"]*16)
PY
tools/rk45_solver --model 204 --params $T/big_params.csv --tf 43200 --nq 0 --math fast --initial-step 0.01 \
   --out-final $T/final_big.csv --out-dense $T/dense_big.csv 2>&1 | tail -3
ls -la $T/big_params.csv $T/final_big.csv
