"""NetCDF-4 I/O on the HDF5 C library (include/hlm/netcdf4_io.hpp): the reader against the
reference's own .nc result files (written by netcdf 4.9.2), the writers by round trip and by
structure comparison with those files.  CPU only; skipped when no HDF5 headers exist."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

HDF5_DIR = os.environ.get("HDF5_DIR", "/opt/conda")
H5DUMP = os.path.join(HDF5_DIR, "bin", "h5dump")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(HDF5_DIR, "include", "hdf5_hl.h")),
                                reason="no HDF5 C library with hdf5_hl in this image")


@pytest.fixture(scope="module")
def nc(tmp_path_factory):
    d = tmp_path_factory.mktemp("nc")
    exe = str(d / "test_netcdf4_io")
    libs = [os.path.join(HDF5_DIR, "lib", n) for n in ("libhdf5_hl.so", "libhdf5.so")]
    subprocess.run(["g++", "-O1", "-std=c++17", "-DHLM_WITH_HDF5", "-I" + os.path.join(ROOT, "include"),
                    "-isystem", os.path.join(HDF5_DIR, "include"), os.path.join(ROOT, "tests", "cpp", "test_netcdf4_io.cpp"),
                    "-o", exe] + libs, check=True)
    env = dict(os.environ, LD_LIBRARY_PATH="")

    def run(*args, ok=True):
        # the HDF5 libs are found through their absolute DT_NEEDED-less paths: preload them
        e = dict(env, LD_PRELOAD=":".join(libs))
        p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=e, timeout=120)
        if ok:
            assert p.returncode == 0, (args, p.stdout, p.stderr)
        return p
    run.dir = d
    return run


def test_reader_on_reference_dense_example(nc):
    """dense_example.nc: outputs(system=10, time=2881, variable=5) double -> float chunk."""
    out = nc.dir / "chunk.bin"
    p = nc("read", os.path.join(GOLDEN, "dense_example.nc"), "outputs", 2, 3, out)
    assert "dims 10 2881 5 loaded=1" in p.stdout and "moved=1 old=0" in p.stdout
    got = np.fromfile(out, dtype=np.float32).reshape(3, 2881, 5)
    gold = np.load(os.path.join(GOLDEN, "g2_m204_dense_sys0.npy")).astype(np.float32)
    for s in range(3):
        assert np.array_equal(got[s], gold)  # all 10 systems of the artifact are identical


def test_reader_errors_match_reference(nc):
    g = os.path.join(GOLDEN, "dense_example.nc")
    o = nc.dir / "x.bin"
    p = nc("read", os.path.join(GOLDEN, "final_example.nc"), "outputs", 0, 1, o, ok=False)
    assert p.returncode == 3 and "Expected 3D variable (time, lat, lon), got 2D" in p.stdout
    p = nc("read", g, "nope", 0, 1, o, ok=False)
    assert p.returncode == 3 and "Variable nope not found in file" in p.stdout
    p = nc("read", str(nc.dir / "absent.nc"), "pr", 0, 1, o, ok=False)
    assert p.returncode == 3 and "runtime_error: Opening file" in p.stdout
    p = nc("read", g, "outputs", 0, 0, o, ok=False)
    assert "invalid_argument: Size of time chunk must be greater than zero" in p.stdout
    p = nc("read", g, "outputs", 10, 1, o, ok=False)
    assert "out_of_range: Start time index out of range" in p.stdout
    p = nc("read", g, "outputs", 8, 3, o, ok=False)
    assert "out_of_range: Requested time steps exceed available data" in p.stdout


def _structure(path):
    """{dataset: (dims, sorted attribute names)} from h5dump -H"""
    txt = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for ln in txt.splitlines():
        m = re.search(r'DATASET "([^"]+)"', ln)
        if m:
            cur = m.group(1)
            out[cur] = [None, []]
        m = re.search(r"DATASPACE\s+SIMPLE \{ \(([^)]*)\)", ln)
        if m and cur and out[cur][0] is None:
            out[cur][0] = tuple(int(x) for x in m.group(1).split(","))
        m = re.search(r'ATTRIBUTE "([^"]+)"', ln)
        if m and cur:
            out[cur][1].append(m.group(1))
    return {k: (v[0], sorted(v[1])) for k, v in out.items()}


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="h5dump not available")
@pytest.mark.parametrize("level", [0, 4])
def test_writers_produce_the_reference_file_structure_and_round_trip(nc, level):
    d, f = nc.dir / f"dense_{level}.nc", nc.dir / f"final_{level}.nc"
    nc("write_dense", d, 10, 2881, 5, level)
    nc("write_final", f, 10, 5, level)
    ref_d, ref_f = _structure(os.path.join(GOLDEN, "dense_example.nc")), _structure(os.path.join(GOLDEN, "final_example.nc"))
    assert _structure(str(d)) == ref_d   # same datasets, dims and attribute names as netcdf-c 4.9.2 wrote
    assert _structure(str(f)) == ref_f
    # values: dense through the product reader, final through h5dump's binary dump
    out = nc.dir / "rt.bin"
    nc("read", d, "outputs", 0, 10, out)
    got = np.fromfile(out, dtype=np.float32).reshape(10, 2881, 5)
    s, q, c = np.meshgrid(np.arange(10), np.arange(2881), np.arange(5), indexing="ij")
    assert np.array_equal(got, (s * 1000.0 + q + 0.125 * c).astype(np.float32))
    fb = nc.dir / "final.bin"
    subprocess.run([H5DUMP, "-d", "/outputs", "-b", "LE", "-o", str(fb), str(f)], check=True, capture_output=True, cwd=nc.dir)
    assert np.array_equal(np.fromfile(fb, dtype="<f8").reshape(10, 5), np.arange(10)[:, None] * 1000.0 + 0.125 * np.arange(5))
    tb = nc.dir / "time.bin"
    subprocess.run([H5DUMP, "-d", "/time", "-b", "LE", "-o", str(tb), str(d)], check=True, capture_output=True, cwd=nc.dir)
    assert np.array_equal(np.fromfile(tb, dtype="<f8"), 60.0 * np.arange(2881))
    if level > 0:
        assert os.path.getsize(d) < 0.6 * 10 * 2881 * 5 * 8  # shuffle + deflate took effect
        hdr = subprocess.run([H5DUMP, "-p", "-H", str(d)], capture_output=True, text=True).stdout
        assert "COMPRESSION DEFLATE { LEVEL 4 }" in hdr and "SHUFFLE" in hdr


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="h5dump not available")
@pytest.mark.parametrize("level,window", [(0, 1), (0, 97), (0, 5000), (4, 97)])
def test_windowed_writer_produces_the_same_file_contents(nc, level, window):
    """DenseNetcdfWriter (the consumer of rk45_api::run_rk45_streamed's windows; BASELINE config 5's output never
    exists whole on the host): same objects, attributes and values as write_dense_netcdf of the whole array."""
    whole, piecewise = nc.dir / f"whole_{level}.nc", nc.dir / f"windows_{level}_{window}.nc"
    nc("write_dense", whole, 7, 500, 3, level)
    nc("write_dense_windows", piecewise, 7, 500, 3, level, window)   # (exit 0: every window written, one past the end refused)
    assert _structure(str(piecewise)) == _structure(str(whole))
    for ds in ("/outputs", "/time", "/system", "/variable"):
        a, b = nc.dir / "a.bin", nc.dir / "b.bin"
        subprocess.run([H5DUMP, "-d", ds, "-b", "LE", "-o", str(a), str(whole)], check=True, capture_output=True, cwd=nc.dir)
        subprocess.run([H5DUMP, "-d", ds, "-b", "LE", "-o", str(b), str(piecewise)], check=True, capture_output=True, cwd=nc.dir)
        assert open(a, "rb").read() == open(b, "rb").read(), ds
