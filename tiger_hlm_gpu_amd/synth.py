"""Deterministic synthetic inputs for the BASELINE.json configurations.

Every value is a pure function of (seed, GLOBAL hillslope index, field), computed with a
vectorised splitmix64 hash, so a rank can build its own contiguous shard
(``start``/``count``) and the union over ranks is identical to the single-GPU input.

No reference data is needed: the named small_example_params.csv is absent from the
reference (.MISSING_LARGE_BLOBS) and the ERA5-Land forcings are not distributed.
"""
import numpy as np

from .models import SP_DTYPE
from .params import C_1

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(seed, idx, field):
    """U[0,1) double for each index in ``idx`` (uint64 array), stream ``field``."""
    with np.errstate(over="ignore"):
        k = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(field))
        z = _splitmix64(np.asarray(idx, dtype=np.uint64) ^ k)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


Y0_204 = (0.01, 3.0, 0.0, 5.0, 0.2)  # main.cpp:376


# Capacities at which the static store really saturates DURING the window (``saturating=True``).  SURVEY's
# "active-surface" draw (hu ~ U[0.005, 0.05], start at h_stat = 3) overflows only in the first minutes: the
# reference's ET stub (model_204.hpp:96-97: Emax = min(0.1 T, h_stat), loss h_stat / Hu * Emax per minute) empties a
# static store of that size within the first hour, after which it sits near sqrt(rain * Hu) -- below Hu whenever the
# rain rate is below Hu.  With the generator's rain rates (1e-4 m/min x Exp(1)) the store therefore overflows in the
# storms only for capacities of that order: log-uniform over [5e-5, 5e-3] m covers hillslopes that saturate in every
# storm, in the heavy ones only, and never.
SATURATING_HU = (5e-5, 5e-3)


def model204_params(start, count, seed=42, active_surface=False, saturating=False):
    """SpatialParams for hillslopes [start, start+count): ranges around data/small_test.csv
    (SURVEY 8d C3).  ``active_surface`` draws hu ~ U[0.005, 0.05] so that the static store
    overflows and the Manning pow() branch is exercised (in the first minutes of a run that starts at
    h_stat = 3: see SATURATING_HU); ``saturating`` draws hu log-uniformly over SATURATING_HU, capacities
    the storms of forcing_tables() fill again and again."""
    idx = np.arange(start, start + count, dtype=np.uint64)
    u = lambda f: uniform(seed, idx, f)
    sp = np.zeros(count, dtype=SP_DTYPE)
    sp["stream"] = 420000000 + idx.astype(np.int64)
    sp["next_stream"] = sp["stream"] + 1
    sp["c1"] = C_1
    sp["Hu"] = (0.005 + 0.045 * u(1)) if active_surface else (50.0 + 250.0 * u(1))
    if saturating:
        lo, hi = np.log(SATURATING_HU[0]), np.log(SATURATING_HU[1])
        sp["Hu"] = np.exp(lo + (hi - lo) * u(1))
    sp["infil"] = 4.0 * C_1
    sp["perco"] = 1.6 * C_1
    sp["n_mann"] = 0.03 + 0.12 * u(2)
    sp["slope"] = 0.002 + 0.098 * u(3)
    sp["L"] = 0.05 + 2.95 * u(4)
    sp["A_h"] = 0.01 + 1.99 * u(5)
    sp["alpha3"] = 2.0 * 24.0 * 60.0
    sp["alpha4"] = 55.0 * 24.0 * 60.0
    sp["melt_f"] = 3.7
    sp["temp_thr"] = 0.0
    sp["sw"] = 0.11
    sp["ss"] = 0.33
    sp["lat"] = 40.0 + u(6)
    return sp


def model204_y0(count, wet_surface=False, settled_for=None):
    """main.cpp:376 initial state.  ``wet_surface`` starts the surface store at 0.01 m: it
    then drains algebraically and never reaches 0, so the Manning pow() term is evaluated
    in every stage of every step (the dry default skips it bit-exactly).
    ``settled_for`` = the hillslopes' SpatialParams: a start that is past the first hour's transient of a small
    capacity -- no snow, the static store half full (min(3, Hu) / 2), the surface store wet.  From main.cpp:376's
    state a capacity of millimetres dumps 3 m of water into a dry surface store at the first step of 0.01 min; the
    reject that follows shrinks the step below (tf - t0) * 1e-6 = 0.043 min of a 30-day window and the hillslope is
    flagged before t = 1 min (rk45_kernel.cu:160: three quarters of SURVEY's active-surface draw)."""
    y0 = np.tile(np.asarray(Y0_204, dtype=np.float64), (count, 1))
    if wet_surface:
        y0[:, 2] = 0.01
    if settled_for is not None:
        y0[:, 0] = 0.0
        y0[:, 1] = 0.5 * np.minimum(3.0, settled_for["Hu"])
        y0[:, 2] = 0.01
    return y0


def cell_of_system(start, count, ncell=1024, run_len=512):
    """Contiguous runs of hillslopes share a forcing grid cell (the sample lookup maps
    41 274 streams to 88 cells)."""
    g = np.arange(start, start + count, dtype=np.int64)
    return ((g // run_len) % ncell).astype(np.int32)


def forcing_tables(days=30, ncell=1024, seed=42, wet_prob=0.15, intensity=1e-4, tmin=0.5, tmax=15.0):
    """Per-cell forcing tables: pr hourly [24*days][ncell] (m/min), t2m daily [days][ncell]
    (deg C), both float32, plus (dt_hours, nT).

    Temperatures default to U[0.5, 15]: the reference's linear ET stub (model_204.hpp:96,
    Emax = 0.1*T) turns negative for T < 0, the static store then grows exponentially
    until it hits Hu, and the kink there flags the system stiff (measured: 28 % flagged
    with U[-5, 15], 0 % with U[0.5, 15]); SURVEY 8d asks for < 1 % flagged."""
    nh, nd = 24 * days, days
    ih = np.arange(nh * ncell, dtype=np.uint64)
    wet = uniform(seed, ih, 101) < wet_prob
    amt = -intensity * np.log1p(-uniform(seed, ih, 102))
    pr = np.where(wet, amt, 0.0).astype(np.float32).reshape(nh, ncell)
    idd = np.arange(nd * ncell, dtype=np.uint64)
    t2m = (tmin + (tmax - tmin) * uniform(seed, idd, 103)).astype(np.float32).reshape(nd, ncell)
    return pr, t2m, (1.0, 24.0), (nh, nd)


def expand_per_stream(pr, t2m, cells):
    """The reference's device layout float[sum nT][ns] (main.cpp:527-549): every stream
    gets its own column.  Host-side version for small cases; bench.py expands on the GPU."""
    return np.concatenate([pr[:, cells], t2m[:, cells]], axis=0).astype(np.float32)


def dummy_y0(start, count, seed=42, spread=0.5):
    """C2: y0_i = 1 + spread*u_i (spread=0 gives the C1 / G1 initial state)."""
    idx = np.arange(start, start + count, dtype=np.uint64)
    cols = [1.0 + spread * uniform(seed, idx, 200 + c) for c in range(5)]
    return np.stack(cols, axis=1)


def dummy_query_times(tf=5.0, nq=10000):
    """t_q = tf (q+1)/(nq+1): the grid of the reference's src/dense.csv."""
    return tf * (np.arange(nq, dtype=np.float64) + 1.0) / (nq + 1.0)


def minute_queries(t0, tf, stride=1.0):
    """for (t = t0; t <= tf; t += stride), main.cpp:653-656."""
    n = int(np.floor((tf - t0) / stride + 1e-9)) + 1
    return t0 + stride * np.arange(n, dtype=np.float64)


def write_params_csv(path, sp):
    """The parameter CSV (data/small_test.csv header; parameters_loader.cpp:35-49) whose load gives back ``sp``
    bit for bit: the loader multiplies i2 / i3 by c = 0.001/60 and res_ss / res_gw by 1440 (:92-101), so the raw
    columns are written as those quotients (exact for the generator's 4, 1.6, 2 and 55) and everything else with
    17 significant digits."""
    cols = ["stream", "next_stream", "drainage_area_km2", "length_km", "area_sqkm", "centroid_lon", "centroid_lat", "hu",
            "i2", "i3", "sw", "ss", "n", "slope", "res_ss", "res_gw", "melt", "t_thres"]
    n = sp.shape[0]
    raw = {
        "drainage_area_km2": sp["A_h"], "length_km": sp["L"], "area_sqkm": sp["A_h"], "centroid_lon": np.zeros(n),
        "centroid_lat": sp["lat"], "hu": sp["Hu"], "i2": np.round(sp["infil"] / C_1, 12), "i3": np.round(sp["perco"] / C_1, 12),
        "sw": sp["sw"], "ss": sp["ss"], "n": sp["n_mann"], "slope": sp["slope"], "res_ss": np.round(sp["alpha3"] / 1440.0, 12),
        "res_gw": np.round(sp["alpha4"] / 1440.0, 12), "melt": sp["melt_f"], "t_thres": sp["temp_thr"]}
    import pandas as pd
    df = pd.DataFrame({"stream": sp["stream"], "next_stream": sp["next_stream"], **{c: raw[c] for c in cols[2:]}})[cols]
    df.to_csv(path, index=False, float_format="%.17g")


def write_lookup_csv(path, streams, cells, nlon=32):
    """stream,lat_index,lon_index (data/small_example_pr_lookup.csv; forcing_loader.cpp:14-64): cell = lat * nlon + lon."""
    import pandas as pd
    cells = np.asarray(cells, dtype=np.int64)
    pd.DataFrame({"stream": np.asarray(streams, dtype=np.int64), "lat_index": cells // nlon, "lon_index": cells % nlon}).to_csv(path, index=False)
