"""ctypes loader for the CPU oracle (oracle/hlm_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HLM_ORACLE_SO: another build of the same text (make -C oracle asan: AddressSanitizer + UBSan)
_SO = os.environ.get("HLM_ORACLE_SO") or os.path.join(_HERE, "libhlm_oracle.so")
_SO_FMA = os.path.join(_HERE, "libhlm_oracle_fma.so")  # conditioning yardstick, see the Makefile

UID_DUMMY = 0
UID_204 = 204
UID_TOY3 = 3   # the build's 3-state example model
NEQ = 5        # the 5-state models; neq(uid) for any uid


def neq(uid):
    return {UID_DUMMY: 5, UID_204: 5, UID_TOY3: 3}.get(uid, NEQ)  # (an unknown uid is the C side's error to report)

# I_O/parameters_loader.hpp:19-37 (136 bytes)
SP_DTYPE = np.dtype(
    [("stream", "<i8"), ("next_stream", "<i8")]
    + [(n, "<f8") for n in ("c1", "infil", "perco", "Hu", "lat", "sw", "ss", "n_mann", "slope",
                            "L", "A_h", "alpha3", "alpha4", "melt_f", "temp_thr")]
)
assert SP_DTYPE.itemsize == 136

DEFAULT_PARAMS = (0.01, 1e-6, 1e-9, 0.9, 0.2, 10.0)  # models/model_204.hpp:22-30


class _SolverParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("initialStep", "rtol", "atol", "safety", "minScale", "maxScale")]


class _Job(C.Structure):
    _fields_ = [
        ("uid", C.c_int), ("ns", C.c_int), ("nq", C.c_int),
        ("t0", C.c_double), ("tf", C.c_double),
        ("y0", C.c_void_p), ("query", C.c_void_p), ("sp", C.c_void_p),
        ("forc", C.c_void_p), ("forc_dt_h", C.c_void_p), ("forc_nT", C.c_void_p),
        ("nforc", C.c_int), ("forc_stub_f64", C.c_void_p),
        ("p", _SolverParams), ("fixed_step", C.c_double),
        ("attempt_cap", C.c_uint32), ("nthreads", C.c_int),
        ("y_final", C.c_void_p), ("dense_raw", C.c_void_p), ("stiff", C.c_void_p),
        ("n_attempt", C.c_void_p), ("n_accept", C.c_void_p), ("t_reached", C.c_void_p),
        ("y_reached", C.c_void_p), ("radau_continue", C.c_int),
        ("start_t", C.c_void_p), ("start_h", C.c_void_p), ("use_t_stop", C.c_int), ("t_stop", C.c_double), ("h_reached", C.c_void_p),
        ("radau_trace", C.c_void_p), ("radau_trace_cap", C.c_int64),
    ]


def build(force=False):
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("hlm_oracle.c", "hlm_oracle_radau.c", "hlm_oracle.h"))
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < newest for so in (_SO, _SO_FMA)):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


_lib = None
_lib_fma = None


def _declare(_lib):
    if True:
        _lib.hlm_oracle_rk45_run.argtypes = [C.POINTER(_Job)]
        _lib.hlm_oracle_rk45_run.restype = C.c_int
        _lib.hlm_oracle_radau_run.argtypes = [C.POINTER(_Job)]
        _lib.hlm_oracle_radau_run.restype = C.c_int
        _lib.hlm_oracle_reorder_dense.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib.hlm_oracle_reorder_dense.restype = None
        _lib.hlm_oracle_reorder_dense_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.hlm_oracle_reorder_dense_n.restype = None
        _lib.hlm_oracle_rhs.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.hlm_oracle_rhs.restype = None
        _lib.hlm_oracle_jacobian.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.hlm_oracle_jacobian.restype = None
        _lib.hlm_oracle_rk45_step.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.hlm_oracle_rk45_step.restype = None
        _lib.hlm_oracle_rk45_dense.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        _lib.hlm_oracle_rk45_dense.restype = None
        for n in ("c", "a", "b", "balt", "P"):
            getattr(_lib, "hlm_oracle_tableau_" + n).restype = C.POINTER(C.c_double)
    return _lib


def lib(fma=False):
    """The oracle library; fma=True: the FMA-contracted build of the same text (yardstick only)."""
    global _lib, _lib_fma
    if _lib is None:
        build()
        _lib = _declare(C.CDLL(_SO))
    if fma and _lib_fma is None:
        _lib_fma = _declare(C.CDLL(_SO_FMA))  # loaded only on request: a yardstick, not the checker
    return _lib_fma if fma else _lib


def _p(a):
    return None if a is None else a.ctypes.data


def tableau():
    L = lib()
    g = lambda n, k: np.array([getattr(L, "hlm_oracle_tableau_" + n)()[i] for i in range(k)])
    return {"c": g("c", 7), "a": g("a", 42).reshape(7, 6), "b": g("b", 7), "b_alt": g("balt", 7),
            "P": g("P", 28).reshape(7, 4)}


def rhs(uid, y, sp=None, F=(), t=0.0):
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(neq(uid))
    Fd = np.ascontiguousarray(np.asarray(F, dtype=np.float32).astype(np.float64))
    spa = None if sp is None else np.ascontiguousarray(sp, dtype=SP_DTYPE)
    lib().hlm_oracle_rhs(uid, t, _p(y), _p(out), _p(spa), _p(Fd) if len(Fd) else None, len(Fd))
    return out


def jacobian(uid, y, sp=None, F=()):
    """d(rhs)/dy, J[i][j] = d dydt_i / d y_j: the exact one-sided derivative the Radau pass uses."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros((NEQ, NEQ))
    Fd = np.ascontiguousarray(np.asarray(F, dtype=np.float32).astype(np.float64))
    spa = None if sp is None else np.ascontiguousarray(sp, dtype=SP_DTYPE)
    lib().hlm_oracle_jacobian(uid, _p(y), _p(out), _p(spa), _p(Fd) if len(Fd) else None, len(Fd))
    return out


def rk45_step(uid, t, y, h, rtol, atol, sp=None, F=()):
    y = np.ascontiguousarray(y, dtype=np.float64)
    Fd = np.ascontiguousarray(np.asarray(F, dtype=np.float32).astype(np.float64))
    spa = None if sp is None else np.ascontiguousarray(sp, dtype=SP_DTYPE)
    k = np.zeros((7, NEQ))
    k[0] = rhs(uid, y, sp, F, t)
    yout = np.zeros(NEQ)
    err = C.c_double(0.0)
    lib().hlm_oracle_rk45_step(uid, t, _p(y), h, rtol, atol, _p(spa), _p(Fd) if len(Fd) else None, len(Fd),
                               _p(k), _p(yout), C.addressof(err))
    return k, yout, err.value


def rk45_dense(y, k, h, theta):
    y = np.ascontiguousarray(y, dtype=np.float64)
    k = np.ascontiguousarray(k, dtype=np.float64)
    out = np.zeros(NEQ)
    lib().hlm_oracle_rk45_dense(_p(y), _p(k), h, theta, _p(out))
    return out


def run(uid, y0, t0, tf, query=None, sp=None, forc=None, forc_dt_h=None, forc_nT=None,
        params=DEFAULT_PARAMS, forc_stub_f64=None, fixed_step=0.0, attempt_cap=0, nthreads=0,
        reorder=True, radau=False, radau_only=False, fma=False, radau_continue=False, start_t=None, start_h=None, t_stop=None,
        continue_from=None, radau_trace_cap=0):
    """Integrate; returns dict(y_final[ns,5], dense[ns,nq,5] (or raw [ns,5,nq]), stiff, n_attempt, n_accept).

    radau=True re-integrates the systems RK45 flagged with the Radau-IIA fallback
    (hlm_oracle_radau_run); radau_only=True skips RK45 and sends every system through it.
    radau_continue=True (with radau): the flagged systems continue from the state RK45 reached instead of
    restarting at t0 (hlm_oracle_job.radau_continue).
    fma=True runs the FMA-contracted build instead: NOT a reference, only the yardstick for how far
    two valid roundings of the same computation separate on this input.
    continue_from = (t[ns], y[ns, N_EQ]) with radau_only: the Radau pass takes every system over at the GIVEN state (what
    another implementation's RK45 pass left when it flagged the system), as radau_continue does with the oracle's own.
    start_t / start_h ([ns]) and t_stop: segment replay (hlm_oracle.h) -- every system enters the loop at (start_t, y0)
    with the step start_h and leaves it paused (stiff = PAUSED = 3, state in t_reached / y_reached / h_reached) before
    the first attempt that starts at or after t_stop.
    radau_trace_cap > 0: the result carries "radau_trace" [ns, cap] uint8, one byte per implicit attempt (hlm_oracle.h)."""
    n = neq(uid)
    y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1, n)
    ns = y0.shape[0]
    q = np.zeros(0) if query is None else np.ascontiguousarray(query, dtype=np.float64)
    nq = q.shape[0]
    spa = None if sp is None else np.ascontiguousarray(sp, dtype=SP_DTYPE)
    if spa is not None:
        assert spa.shape == (ns,)
    j = _Job()
    j.uid, j.ns, j.nq, j.t0, j.tf = uid, ns, nq, float(t0), float(tf)
    j.y0, j.query, j.sp = _p(y0), _p(q) if nq else None, _p(spa)
    keep = []
    if forc_stub_f64 is not None:
        stub = np.ascontiguousarray(forc_stub_f64, dtype=np.float64)
        j.forc_stub_f64, j.nforc = _p(stub), stub.shape[0]
        keep.append(stub)
    elif forc is not None:
        f = np.ascontiguousarray(forc, dtype=np.float32)
        dt = np.ascontiguousarray(forc_dt_h, dtype=np.float64)
        nT = np.ascontiguousarray(forc_nT, dtype=np.uint64)
        assert f.size == int(nT.sum()) * ns, (f.size, nT, ns)
        j.forc, j.forc_dt_h, j.forc_nT, j.nforc = _p(f), _p(dt), _p(nT), dt.shape[0]
        keep += [f, dt, nT]
    j.p = _SolverParams(*[float(x) for x in params])
    j.fixed_step, j.attempt_cap, j.nthreads = float(fixed_step), int(attempt_cap), int(nthreads)
    yf = np.empty((ns, n))
    raw = np.empty((ns, n, nq)) if nq else None
    stiff = np.empty(ns, dtype=np.int32)
    na = np.zeros(ns, dtype=np.uint32)
    nacc = np.zeros(ns, dtype=np.uint32)
    treach = np.zeros(ns)
    j.y_final, j.dense_raw, j.stiff, j.n_attempt, j.n_accept = _p(yf), _p(raw), _p(stiff), _p(na), _p(nacc)
    j.t_reached = _p(treach)
    yreach = np.zeros((ns, n))
    j.y_reached = _p(yreach)
    hreach = np.zeros(ns)
    j.h_reached = _p(hreach)
    if start_t is not None:
        st = np.ascontiguousarray(start_t, dtype=np.float64)
        sh = np.ascontiguousarray(start_h, dtype=np.float64)
        assert st.shape == (ns,) and sh.shape == (ns,)
        j.start_t, j.start_h = _p(st), _p(sh)
        keep += [st, sh]
    if t_stop is not None:
        j.use_t_stop, j.t_stop = 1, float(t_stop)
    j.radau_continue = 1 if (radau_continue and not radau_only) else 0
    if continue_from is not None:
        assert radau_only, "continue_from hands given states to the Radau pass alone"
        treach[:] = np.asarray(continue_from[0], dtype=np.float64)
        yreach[:] = np.asarray(continue_from[1], dtype=np.float64).reshape(ns, n)
        j.radau_continue = 1
    trace = None
    if radau_trace_cap > 0:
        trace = np.zeros((ns, int(radau_trace_cap)), dtype=np.uint8)
        j.radau_trace, j.radau_trace_cap = _p(trace), int(radau_trace_cap)
    if radau_only:
        stiff[:] = 1
        yf[:] = 0.0
        rc = 0
    else:
        rc = lib(fma).hlm_oracle_rk45_run(C.byref(j))
    if rc != 0:
        raise ValueError("hlm_oracle_rk45_run: bad arguments")
    if radau or radau_only:
        if lib(fma).hlm_oracle_radau_run(C.byref(j)) != 0:
            raise ValueError("hlm_oracle_radau_run: bad arguments")
    dense = raw
    if nq and reorder:
        dense = np.empty((ns, nq, n))
        lib().hlm_oracle_reorder_dense_n(_p(raw), _p(dense), ns, nq, n)
    return {"y_final": yf, "dense": dense, "stiff": stiff, "n_attempt": na, "n_accept": nacc, "t_reached": treach,
            "y_reached": yreach, "h_reached": hreach, "radau_trace": trace}
