#!/usr/bin/env python3
"""Extract the reference's committed result artifacts into small golden fixtures.

Run ONLY in the build container (needs /root/reference and /opt/conda/bin/h5dump);
the outputs below are committed so that tests run where the reference is absent.
Only DATA files are read (CSV / NetCDF-4 result artifacts and the 10-row sample
parameter table); no reference source text is copied.

  G1  src/final.csv, src/dense.csv            DummyModel, 4 identical systems, nq=10000
  G2  src/final_example.nc, dense_example.nc  Model 204, 10 systems, nq=2881 (1-min grid)
  IN  data/small_test.csv                      the parameter rows that reproduce G2
  IN  data/small_example_pr_lookup.csv         the sample stream -> grid-cell lookup (41 274 streams, 88 cells;
                                               the t2m lookup is byte-identical)
"""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

REF = "/root/reference"
H5DUMP = "/opt/conda/bin/h5dump"
HERE = os.path.dirname(os.path.abspath(__file__))


def h5_f64(path, dset, shape):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "d.bin")
        subprocess.run([H5DUMP, "-d", dset, "-b", "LE", "-o", out, path], check=True,
                       stdout=subprocess.DEVNULL, cwd=td)
        return np.fromfile(out, dtype="<f8").reshape(shape)


def main():
    meta = {}
    # ---- G1 -------------------------------------------------------------------------
    shutil.copyfile(f"{REF}/src/final.csv", f"{HERE}/g1_dummy_final.csv")
    with open(f"{REF}/src/dense.csv") as f:
        header = f.readline().rstrip("\n").split(",")
        rows = [ln.rstrip("\n").split(",") for ln in f if ln.strip()]
    nsys = (len(header) - 1) // 5
    # all systems are textually identical per row -> keep system 0 only (6 columns)
    for r in rows:
        for s in range(1, nsys):
            assert r[1 + 5 * s:6 + 5 * s] == r[1:6]
    with open(f"{HERE}/g1_dummy_dense_sys0.csv", "w") as f:
        f.write(",".join(header[:6]) + "\n")
        for r in rows:
            f.write(",".join(r[:6]) + "\n")
    meta["g1"] = {"source": ["src/final.csv", "src/dense.csv"], "nsys": nsys, "nq": len(rows),
                  "all_systems_identical": True, "header": header,
                  "inputs": "DummyModel rhs (model_dummy_python.ipynb:65-89), y0=1, t in [0,5], "
                            "t_q=5(q+1)/10001, Parameters defaults (initialStep=0.01)"}
    # ---- G2 -------------------------------------------------------------------------
    fin = h5_f64(f"{REF}/src/final_example.nc", "/outputs", (10, 5))
    den = h5_f64(f"{REF}/src/dense_example.nc", "/outputs", (10, 2881, 5))
    tim = h5_f64(f"{REF}/src/dense_example.nc", "/time", (2881,))
    assert np.array_equal(tim, np.arange(2881.0))
    ident = bool(all(np.array_equal(den[s], den[0]) for s in range(10)))
    np.save(f"{HERE}/g2_m204_final.npy", fin)
    if ident:
        np.save(f"{HERE}/g2_m204_dense_sys0.npy", den[0])
    else:
        np.save(f"{HERE}/g2_m204_dense.npy", den)
    shutil.copyfile(f"{REF}/data/small_test.csv", f"{HERE}/small_test.csv")
    # the sample lookup table (reference-held DATA): fixture for LookupMapper and the per-cell column index
    shutil.copyfile(f"{REF}/data/small_example_pr_lookup.csv", f"{HERE}/lookup_small_example.csv")
    assert open(f"{REF}/data/small_example_pr_lookup.csv", "rb").read() == open(f"{REF}/data/small_example_t2m_lookup.csv", "rb").read()
    # the two NetCDF-4 result files themselves (data, written by netcdf 4.9.2 / hdf5 1.14.4):
    # fixtures for the HDF5-based NetCDF reader, include/hlm/netcdf4_io.hpp
    shutil.copyfile(f"{REF}/src/final_example.nc", f"{HERE}/final_example.nc")
    shutil.copyfile(f"{REF}/src/dense_example.nc", f"{HERE}/dense_example.nc")
    meta["g2"] = {"source": ["src/final_example.nc", "src/dense_example.nc", "data/small_test.csv"],
                  "nsys": 10, "nq": 2881, "all_systems_identical": ident,
                  "inputs": "Model 204, params data/small_test.csv via loadSpatialParams, "
                            "y0={0.01,3,0,5,0.2}, t in [0,2880] min, queries 0..2880 step 1, "
                            "rtol=1e-6 atol=1e-9 safety=.9 minScale=.2 maxScale=10 initialStep=1e-6, "
                            "stub double forcings rainfall=0.001 temperature=1.0 (SURVEY 8c)"}
    with open(f"{HERE}/golden_meta.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "header"} for k, v in meta.items()}, indent=1))


if __name__ == "__main__":
    main()
