"""Python mirror of the reference's host API (namespace rk45_api), calling libhlm_hip.so
through the C ABI only.

  set_model_parameters(Model, p)      model_registry.hpp:12-13 / model_registry.cpp:18-60
  setup_gpu_buffers(Model, y0, q)     rk45_api.hpp:63-116
  launch_rk45_kernel(Model, bufs, ..) rk45_api.hpp:120-154
  retrieve_and_free(Model, bufs)      rk45_api.hpp:159-270
  run_rk45(Model, y0, t0, tf, q, d_sp) rk45_api.hpp:273-313

plus the device-resident entry (integrate_device) that bench.py and the multi-GPU path
use with torch-owned HBM buffers.  torch is plumbing only (device memory, streams).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DeviceJob, GpuBuffers, HlmError, check, lib
from .models import SP_DTYPE, Parameters


def _uid(model):
    return model if isinstance(model, int) else model.UID


def _ptr(a):
    return None if a is None else a.ctypes.data


def device_count():
    n = C.c_int(0)
    rc = lib().hlm_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def set_device(dev):
    check(lib().hlm_set_device(int(dev)))


class DeviceBuffer:
    """Caller-owned device allocation (what main.cpp:392-404 does for d_sp)."""

    def __init__(self, nbytes):
        p = C.c_void_p(0)
        check(lib().hlm_device_malloc(C.byref(p), int(nbytes)))
        self.ptr = p.value or 0
        self.nbytes = int(nbytes)

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        check(lib().hlm_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes))
        return b

    def to_host(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib().hlm_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(lib().hlm_device_free(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def upload_spatial_params(sp):
    """H2D of the 136-byte AoS records (main.cpp:392-404)."""
    sp = np.ascontiguousarray(sp, dtype=SP_DTYPE)
    return DeviceBuffer.from_host(sp)


# ---- registry ------------------------------------------------------------------------------
def set_model_parameters(model, p):
    """rk45_api::setModelParameters<Model>(p)."""
    if isinstance(p, Parameters):
        p = p.as_tuple()
    arr = (C.c_double * 6)(*[float(x) for x in p])
    check(lib().hlm_set_model_parameters(_uid(model), arr))


def get_model_parameters(model):
    arr = (C.c_double * 6)()
    check(lib().hlm_get_model_parameters(_uid(model), arr))
    return Parameters(*list(arr))


def set_fixed_step(model, h):
    check(lib().hlm_set_fixed_step(_uid(model), float(h)))


def set_attempt_cap(cap):
    check(lib().hlm_set_attempt_cap(int(cap)))


def set_math_mode(mode):
    check(lib().hlm_set_math_mode(int(mode)))


def set_time_pieces(pieces):
    """Launch shape of the host API's runs without dense output (hlm_set_time_pieces): > 1 = a persistent grid over a
    queue of (time piece, tile) items; 0 / 1 = one wave per tile; negative = automatic (from 4 096 hillslopes
    on, one piece per twelve simulated hours, at most 32, none below 8)."""
    check(lib().hlm_set_time_pieces(int(pieces)))


def set_queue_shape(run_pieces=0, group_tiles=0):
    """hlm_set_queue_shape: queue_run_pieces / queue_group_tiles of the host API's launches in time pieces (0, 0 =
    automatic runs, an item is a tile; group_tiles = 8 for domains that lose hillslopes to the stiff flag)."""
    check(lib().hlm_set_queue_shape(int(run_pieces), int(group_tiles)))


def set_queue_hybrid(mode=0):
    """hlm_set_queue_hybrid: 1 = waves holding saturated hillslopes run one block of general stages (domains whose static
    stores the storms fill), -1 / 0 = a block per RHS form (the default)."""
    check(lib().hlm_set_queue_hybrid(int(mode)))


def time_pieces_for(job):
    """The number of time pieces integrate_device(job) launches with (0 = one wave per tile)."""
    return int(lib().hlm_time_pieces_for(C.byref(job)))


def set_radau_fallback(enabled):
    """Host API (run_rk45 / launch_rk45_kernel): re-integrate flagged systems with Radau IIA.  False / 0: off
    (the reference's observable behaviour); True / 1: every flagged system restarts at t0 (radau_kernel.cu:44-52);
    2 or "continue": it continues from the state it had reached when RK45 flagged it."""
    mode = 2 if enabled == "continue" else int(enabled)
    check(lib().hlm_set_radau_fallback(mode))


def set_radau_pieces(pieces=0):
    """hlm_set_radau_pieces: the Radau pass's work queue in `pieces` time pieces (1 = whole systems, at most 16); 0 =
    automatic (16 restarting at t0; continuing from the flag states: whole systems up to the last six of 16 pieces).
    Results do not depend on it."""
    check(lib().hlm_set_radau_pieces(int(pieces)))


def set_output_states(comps=None):
    """Output selection of the host API (hlm_set_output_states; the reference's ``output.states``,
    data/config.yaml:44-52): the dense rows of setup_gpu_buffers / run_rk45 carry only the state components
    ``comps`` (ascending, distinct).  None or () = all components."""
    comps = [] if comps is None else [int(c) for c in comps]
    arr = (C.c_int * max(len(comps), 1))(*comps)
    check(lib().hlm_set_output_states(len(comps), arr))


class Forcings:
    """Device-resident forcing block + metadata (the reference's c_forc_dt / c_forc_nT /
    d_forc_data / nForc symbols, forcing_data.cu:5-10)."""

    def __init__(self, data, dt_hours, nT, ld, col=None):
        data = np.ascontiguousarray(data, dtype=np.float32)
        self.dt_hours = [float(x) for x in dt_hours]
        self.nT = [int(x) for x in nT]
        self.ld = int(ld)
        assert data.size == sum(self.nT) * self.ld, (data.size, self.nT, self.ld)
        self.buf = DeviceBuffer.from_host(data)
        self.col = None if col is None else DeviceBuffer.from_host(np.ascontiguousarray(col, dtype=np.int32))

    def register(self):
        n = len(self.nT)
        dt = (C.c_double * n)(*self.dt_hours)
        nt = (C.c_uint64 * n)(*self.nT)
        check(lib().hlm_set_forcings(self.buf.ptr, dt, nt, n, self.ld, self.col.ptr if self.col else None))

    def free(self):
        check(lib().hlm_clear_forcings())
        self.buf.free()
        if self.col:
            self.col.free()


def set_forcings(data, dt_hours, nT, ld, col=None):
    f = Forcings(data, dt_hours, nT, ld, col)
    f.register()
    return f


class HostForcings:
    """Forcing series kept in HOST memory and streamed through HBM in time chunks by
    launch_rk45_kernel / run_rk45 (hlm_set_forcings_host; BASELINE config 3 "forcings
    streamed").  ``tables``: one float32 array [nT_j][ld] per forcing; the object keeps them
    alive until free()."""

    def __init__(self, tables, dt_hours, ld, chunk_minutes, col=None):
        self.tables = [np.ascontiguousarray(t, dtype=np.float32) for t in tables]
        self.dt_hours = [float(x) for x in dt_hours]
        self.ld = int(ld)
        assert all(t.ndim == 2 and t.shape[1] == self.ld for t in self.tables)
        self.chunk_minutes = float(chunk_minutes)
        self.col = None if col is None else DeviceBuffer.from_host(np.ascontiguousarray(col, dtype=np.int32))

    def register(self):
        n = len(self.tables)
        ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in self.tables])
        dt = (C.c_double * n)(*self.dt_hours)
        nt = (C.c_uint64 * n)(*[t.shape[0] for t in self.tables])
        check(lib().hlm_set_forcings_host(ptrs, dt, nt, n, self.ld, self.col.ptr if self.col else None, self.chunk_minutes))

    def free(self):
        check(lib().hlm_clear_forcings())
        self.tables = []
        if self.col:
            self.col.free()


def set_forcings_host(tables, dt_hours, ld, chunk_minutes, col=None):
    f = HostForcings(tables, dt_hours, ld, chunk_minutes, col)
    f.register()
    return f


def clear_forcings():
    check(lib().hlm_clear_forcings())


# ---- three-phase API ---------------------------------------------------------------------------
def setup_gpu_buffers(model, h_y0, h_query_times):
    y0 = np.ascontiguousarray(h_y0, dtype=np.float64).reshape(-1)
    q = np.ascontiguousarray(h_query_times if h_query_times is not None else [], dtype=np.float64).reshape(-1)
    ns = y0.size // model.N_EQ
    b = GpuBuffers()
    check(lib().hlm_setup_gpu_buffers_model(_uid(model), _ptr(y0) if ns else None, ns, _ptr(q) if q.size else None, q.size,
                                            C.byref(b)))
    return b


def launch_rk45_kernel(model, bufs, t0, tf, d_sp):
    sp_ptr = d_sp.ptr if isinstance(d_sp, DeviceBuffer) else d_sp
    check(lib().hlm_launch_rk45_kernel(_uid(model), C.byref(bufs), float(t0), float(tf), sp_ptr))


def retrieve_and_free(model, bufs, with_info=False, dense_out=None):
    """``dense_out``: optional float64 array / pinned torch tensor of ns*nq*N_EQ elements to receive the dense
    rows (pinned, device-mapped memory is written by the transposing kernel directly)."""
    ns, nq, neq = bufs.num_systems, bufs.num_queries, model.N_EQ
    width = bufs.dense_ncomp if bufs.dense_ncomp > 0 else neq  # doubles per dense row (output selection)
    final = np.empty(ns * neq)
    if dense_out is None:
        dense = np.empty(ns * nq * width)
    else:
        dense = dense_out.numpy().reshape(-1) if hasattr(dense_out, "numpy") else np.asarray(dense_out).reshape(-1)
        assert dense.dtype == np.float64 and dense.size == ns * nq * width and dense.flags.c_contiguous
    stiff = np.empty(ns, dtype=np.int32)
    na = np.empty(ns, dtype=np.uint32)
    nacc = np.empty(ns, dtype=np.uint32)
    check(lib().hlm_retrieve_and_free(C.byref(bufs), _ptr(final), _ptr(dense) if dense.size else None,
                                      _ptr(stiff), _ptr(na), _ptr(nacc)))
    if with_info:
        return final, dense, {"stiff": stiff, "n_attempt": na, "n_accept": nacc}
    return final, dense


def run_rk45(model, h_y0, t0, tf, h_query_times, d_sp, with_info=False, dense_width=None):
    """std::pair<FinalType, DenseType> run_rk45<Model>(h_y0, t0, tf, h_query_times, d_sp):
    first = y_final[s*N_EQ+i]; second = dense[(s*nq+q)*N_EQ+c].  With a registered output selection
    (set_output_states) pass its length as ``dense_width``: second = dense[(s*nq+q)*K+k]."""
    y0 = np.ascontiguousarray(h_y0, dtype=np.float64).reshape(-1)
    q = np.ascontiguousarray(h_query_times if h_query_times is not None else [], dtype=np.float64).reshape(-1)
    neq = model.N_EQ
    ns, nq = y0.size // neq, q.size
    final = np.empty(ns * neq)
    dense = np.empty(ns * nq * (neq if dense_width is None else int(dense_width)))
    stiff = np.empty(ns, dtype=np.int32)
    na = np.empty(ns, dtype=np.uint32)
    nacc = np.empty(ns, dtype=np.uint32)
    sp_ptr = d_sp.ptr if isinstance(d_sp, DeviceBuffer) else d_sp
    check(lib().hlm_run_rk45(_uid(model), _ptr(y0) if ns else None, ns, float(t0), float(tf),
                             _ptr(q) if nq else None, nq, sp_ptr, _ptr(final) if ns else None,
                             _ptr(dense) if dense.size else None, _ptr(stiff), _ptr(na), _ptr(nacc)))
    if with_info:
        return final, dense, {"stiff": stiff, "n_attempt": na, "n_accept": nacc}
    return final, dense


def run_rk45_streamed(model, h_y0, t0, tf, h_query_times, d_sp, window_queries, sink=None, with_info=False):
    """run_rk45 for dense output that fits no host array (hlm_run_rk45_streamed): the rows reach
    ``sink(query_begin, query_end, rows[ns, W, K])`` window by window (a view valid until the sink returns);
    returns the final states (and flags / counters with ``with_info``)."""
    y0 = np.ascontiguousarray(h_y0, dtype=np.float64).reshape(-1)
    q = np.ascontiguousarray(h_query_times, dtype=np.float64).reshape(-1)
    neq = model.N_EQ
    ns, nq = y0.size // neq, q.size
    width = [neq]
    err = []

    def trampoline(_user, qb, qe, ptr):
        try:
            r = sink(qb, qe, np.ctypeslib.as_array(ptr, shape=(ns, qe - qb, width[0])))
            return int(r) if r else 0
        except BaseException as ex:  # never let an exception cross the C frame
            err.append(ex)
            return -1
    # the registered output selection decides the row width: ask the library through a zero-size setup
    probe = GpuBuffers()
    check(lib().hlm_setup_gpu_buffers_model(_uid(model), None, 0, None, 0, C.byref(probe)))
    width[0] = probe.dense_ncomp if probe.dense_ncomp > 0 else neq
    check(lib().hlm_retrieve_and_free(C.byref(probe), None, None, None, None, None))
    cb = _lib.WINDOW_SINK(trampoline) if sink is not None else None
    final = np.empty(ns * neq)
    stiff = np.empty(ns, dtype=np.int32)
    na = np.empty(ns, dtype=np.uint32)
    nacc = np.empty(ns, dtype=np.uint32)
    sp_ptr = d_sp.ptr if isinstance(d_sp, DeviceBuffer) else d_sp
    try:
        check(lib().hlm_run_rk45_streamed(_uid(model), _ptr(y0) if ns else None, ns, float(t0), float(tf), _ptr(q), nq, sp_ptr,
                                          int(window_queries), C.cast(cb, C.c_void_p) if cb else None, None,
                                          _ptr(final) if ns else None, _ptr(stiff), _ptr(na), _ptr(nacc)))
    except HlmError:
        if err:
            raise err[0]
        raise
    if with_info:
        return final, {"stiff": stiff, "n_attempt": na, "n_accept": nacc}
    return final


# ---- device-resident path (torch tensors as HBM buffers) ------------------------------------------
def make_device_job(model, *, y0, t0, tf, query=None, sp=None, forc=None, forc_dt_hours=(), forc_nT=(),
                    forc_ld=0, forc_col=None, params=None, fixed_step=0.0, attempt_cap=0,
                    math_mode=_lib.MATH_FAST, y_final, dense=None, stiff, n_attempt=None, n_accept=None,
                    dense_states=None):
    """Fill an hlm_device_job from torch CUDA tensors (or raw int device pointers)."""
    def dp(t):
        if t is None:
            return None
        return t if isinstance(t, int) else t.data_ptr()

    j = DeviceJob()
    j.uid = _uid(model)
    ns = y0.numel() // model.N_EQ if not isinstance(y0, int) else None
    j.num_systems = ns
    j.num_queries = 0 if query is None else query.numel()
    j.t0, j.tf = float(t0), float(tf)
    j.d_y0, j.d_query, j.d_sp = dp(y0), dp(query), dp(sp)
    j.d_forc, j.d_forc_col, j.forc_ld = dp(forc), dp(forc_col), int(forc_ld)
    j.nforc = len(forc_nT)
    for k, (dt, nt) in enumerate(zip(forc_dt_hours, forc_nT)):
        j.forc_dt_hours[k] = float(dt)
        j.forc_nT[k] = int(nt)
    p = (params if params is not None else Parameters())
    p = p.as_tuple() if isinstance(p, Parameters) else tuple(float(x) for x in p)
    for k in range(6):
        j.params[k] = p[k]
    j.fixed_step, j.attempt_cap, j.math_mode = float(fixed_step), int(attempt_cap), int(math_mode)
    j.d_y_final, j.d_dense, j.d_stiff = dp(y_final), dp(dense), dp(stiff)
    j.d_n_attempt, j.d_n_accept = dp(n_attempt), dp(n_accept)
    if dense_states is not None:  # output selection: rows of len(dense_states) components (hlm_device_job.dense_ncomp)
        j.dense_ncomp = len(dense_states)
        for k, c in enumerate(dense_states):
            j.dense_comp[k] = int(c)
    return j


def integrate_device(job, stream=None):
    """Enqueue on ``stream`` (a torch.cuda.Stream, a raw hipStream_t int, or None)."""
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    check(lib().hlm_rk45_integrate_device(C.byref(job), s))


def radau_fallback_device(job, stream=None):
    """Enqueue the Radau-IIA pass over the systems the RK45 pass of ``job`` flagged
    (hlm_radau_fallback_device); same stream, after integrate_device(job)."""
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    check(lib().hlm_radau_fallback_device(C.byref(job), s))


def integrate_streamed(job, window_queries, h_dense, stream=None):
    """Dense output streamed to the host in query windows with overlapped D2H
    (hlm_rk45_integrate_streamed).  ``h_dense``: a C-contiguous float64 host array of
    ns*nq*5 elements (numpy array, ideally backed by pinned memory) or a raw pointer.
    Returns the hlm_stream_stats as a dict."""
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    ptr = h_dense if isinstance(h_dense, int) else h_dense.ctypes.data
    st = _lib.StreamStats()
    check(lib().hlm_rk45_integrate_streamed(C.byref(job), int(window_queries), ptr, s, C.byref(st)))
    return {"wall_ms": st.wall_ms, "kernel_ms": st.kernel_ms, "copy_ms": st.copy_ms, "windows": st.windows,
            "dense_bytes": st.dense_bytes}


def integrate_streamed_sink(job, window_queries, sink=None, stream=None):
    """Dense output streamed window by window through pinned staging buffers of the library into ``sink``
    (hlm_rk45_integrate_streamed_sink): ``sink(query_begin, query_end, rows)`` is called with a numpy VIEW
    [ns, query_end - query_begin, K] of the staging buffer, valid until it returns (copy what is kept); it may
    return a non-zero int to stop the run.  sink=None discards the rows.  Returns the stats dict."""
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    ns = job.num_systems
    width = job.dense_ncomp if job.dense_ncomp > 0 else model_info(job.uid)[0]
    err = []

    def trampoline(_user, qb, qe, ptr):
        try:
            rows = np.ctypeslib.as_array(ptr, shape=(ns, qe - qb, width))
            r = sink(qb, qe, rows)
            return int(r) if r else 0
        except BaseException as ex:  # never let an exception cross the C frame
            err.append(ex)
            return -1
    cb = _lib.WINDOW_SINK(trampoline) if sink is not None else None
    st = _lib.StreamStats()
    try:
        check(lib().hlm_rk45_integrate_streamed_sink(C.byref(job), int(window_queries), C.cast(cb, C.c_void_p) if cb else None,
                                                     None, s, C.byref(st)))
    except HlmError:
        if err:
            raise err[0]
        raise
    return {"wall_ms": st.wall_ms, "kernel_ms": st.kernel_ms, "copy_ms": st.copy_ms, "windows": st.windows,
            "dense_bytes": st.dense_bytes}


def resume_bytes(ns, model=None):
    """Bytes of the integrator-state record of ``ns`` systems ((N_EQ + 2) doubles + 5 int32 each)."""
    if model is None:
        return int(lib().hlm_resume_bytes(int(ns)))
    return int(lib().hlm_resume_bytes_model(_uid(model), int(ns)))


def model_info(model):
    """(N_EQ, number of forcings rhs reads) of a registered model, as the library has it."""
    n, f = C.c_int(0), C.c_int(0)
    check(lib().hlm_model_info(_uid(model), C.byref(n), C.byref(f)))
    return n.value, f.value


def time_device(job, iters, stream=None):
    """Per-launch kernel milliseconds measured with HIP events on the launch stream."""
    s = None if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    ms = (C.c_float * iters)()
    check(lib().hlm_rk45_time_device(C.byref(job), s, iters, ms))
    return list(ms)


def kernel_info(model, dense=False, math_mode=_lib.MATH_FAST):
    v, s, l, m = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    check(lib().hlm_kernel_info(_uid(model), int(dense), int(math_mode), C.byref(v), C.byref(s), C.byref(l), C.byref(m)))
    return {"vgprs": v.value, "scratch_bytes": s.value, "lds_bytes": l.value, "max_threads": m.value}


# ---- one process, the GPUs of a node ---------------------------------------------------------------------
def shard_bounds(num_systems, num_shards, shard):
    """(start, count) of shard ``shard``: the reference's split rule (main.cpp:275-307), hlm_shard_bounds."""
    a, b = C.c_int(0), C.c_int(0)
    check(lib().hlm_shard_bounds(int(num_systems), int(num_shards), int(shard), C.byref(a), C.byref(b)))
    return a.value, b.value


def gather_plan(num_systems, num_shards, num_devices, neq=_lib.HLM_NEQ):
    """hlm_gather_plan: the end-of-run gather of hlm_run_rk45_sharded as a list of dicts (shard, from_rank, array,
    dst_offset_bytes, bytes) -- no device is touched."""
    n = C.c_int(0)
    check(lib().hlm_gather_plan(int(num_systems), int(num_shards), int(num_devices), int(neq), None, C.byref(n)))
    arr = (_lib.GatherPiece * max(n.value, 1))()
    check(lib().hlm_gather_plan(int(num_systems), int(num_shards), int(num_devices), int(neq), arr, C.byref(n)))
    return [{f: getattr(arr[k], f) for f, _ in _lib.GatherPiece._fields_} for k in range(n.value)]


def set_sharded_upload(mode):
    """hlm_set_sharded_upload: "threads" (default: one host thread per device issues its copies), "single" (round 4's
    path: one thread for all devices) or "staged" (threads packing through a pair of pinned buffers each)."""
    check(lib().hlm_set_sharded_upload({"single": _lib.UPLOAD_SINGLE_THREAD, "threads": _lib.UPLOAD_THREADS, "staged": _lib.UPLOAD_STAGED_THREADS}[mode]))


def _sharded_stats(st):
    d = {f: getattr(st, f) for f, _ in _lib.ShardedStats._fields_}
    nd = st.num_devices
    for k in ("device_ids", "upload_ms_device", "integrate_ms_device"):
        d[k] = [d[k][i] for i in range(nd)]
    return d


def _sharded_run_struct(model, y0, t0, tf, sp, tables, dt_hours, cells, devices, num_devices):
    uid, neq = _uid(model), model.N_EQ
    y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1, neq)
    ns = y0.shape[0]
    r = _lib.ShardedRun()
    r.uid, r.num_systems, r.t0, r.tf = uid, ns, float(t0), float(tf)
    r.h_y0 = _ptr(y0)
    keep = [y0]
    if sp is not None:
        sp = np.ascontiguousarray(sp, dtype=SP_DTYPE)
        assert sp.shape == (ns,)
        r.h_sp = _ptr(sp)
        keep.append(sp)
    tabs = [np.ascontiguousarray(t, dtype=np.float32) for t in tables]
    if tabs:
        ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
        r.h_forc_tables = C.cast(ptrs, C.POINTER(C.c_void_p))
        r.nforc, r.forc_ld = len(tabs), int(tabs[0].shape[1])
        for k, t in enumerate(tabs):
            assert t.shape[1] == r.forc_ld
            r.forc_dt_hours[k], r.forc_nT[k] = float(dt_hours[k]), int(t.shape[0])
        keep += [tabs, ptrs]
    if cells is not None:
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        assert cells.shape == (ns,)
        r.h_forc_col = _ptr(cells)
        keep.append(cells)
    if devices is None:
        nd, dev_arr = int(num_devices if num_devices is not None else 1), None
    else:
        nd = len(devices)
        dev_arr = (C.c_int * nd)(*[int(d) for d in devices])
    return r, keep, ns, neq, nd, dev_arr


def run_rk45_sharded_streamed(model, y0, t0, tf, query, window_queries, sp=None, tables=(), dt_hours=(), cells=None, num_shards=1,
                              devices=None, num_devices=None, sink=None):
    """hlm_run_rk45_sharded_streamed: BASELINE config 5 on the GPUs of a node in one call -- every device streams the dense
    rows of its shards through its own pinned windows into ``sink(shard, sys_begin, sys_count, query_begin, query_end,
    rows[sys_count, W, K])`` (called from the device's host thread; None discards).  Returns (results dict, stats dict)."""
    r, keep, ns, neq, nd, dev_arr = _sharded_run_struct(model, y0, t0, tf, sp, tables, dt_hours, cells, devices, num_devices)
    q = np.ascontiguousarray(query, dtype=np.float64)
    errors = []

    def trampoline(_user, shard, s0, cnt, qb, qe, ptr):
        try:
            if sink is not None:
                k = neq   # (output selection: the registered one; rows are [cnt][qe - qb][K])
                rows = np.ctypeslib.as_array(ptr, shape=(cnt * (qe - qb) * k,)).reshape(cnt, qe - qb, k)
                sink(shard, s0, cnt, qb, qe, rows)
            return 0
        except Exception as e:  # noqa: BLE001 -- an exception must not cross the C boundary
            errors.append(e)
            return 1
    cb = _lib.SHARD_WINDOW_SINK(trampoline) if sink is not None else None   # (NULL: the library discards the rows)
    yf = np.empty((ns, neq))
    stiff = np.empty(ns, dtype=np.int32)
    na, nacc = np.empty(ns, dtype=np.uint32), np.empty(ns, dtype=np.uint32)
    st = _lib.ShardedStats()
    rc = lib().hlm_run_rk45_sharded_streamed(C.byref(r), int(num_shards), dev_arr, nd, _ptr(q), int(q.shape[0]), int(window_queries),
                                             C.cast(cb, C.c_void_p) if cb else None, None, _ptr(yf), _ptr(stiff), _ptr(na), _ptr(nacc), C.byref(st))
    if errors:
        raise errors[0]
    check(rc)
    del keep
    return {"y_final": yf, "stiff": stiff, "n_attempt": na, "n_accept": nacc}, _sharded_stats(st)


def run_rk45_sharded(model, y0, t0, tf, sp=None, tables=(), dt_hours=(), cells=None, num_shards=1, devices=None,
                     num_devices=None, gather="rccl"):
    """hlm_run_rk45_sharded: the whole domain from host arrays, split by the reference's rule over ``num_shards`` shards
    on ``devices`` (default: 0 .. num_devices - 1) in THIS process, end-of-run gather over RCCL (``gather`` = "rccl")
    or by per-device copies ("host").  ``tables``: one float32 array [nT_j][ld] per forcing; ``cells``: int32 column of
    each hillslope (per-cell tables), None = per-stream tables (ld == ns).  Returns (results dict, stats dict)."""
    uid, neq = _uid(model), model.N_EQ
    y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1, neq)
    ns = y0.shape[0]
    r = _lib.ShardedRun()
    r.uid, r.num_systems, r.t0, r.tf = uid, ns, float(t0), float(tf)
    r.h_y0 = _ptr(y0)
    keep = [y0]
    if sp is not None:
        sp = np.ascontiguousarray(sp, dtype=SP_DTYPE)
        assert sp.shape == (ns,)
        r.h_sp = _ptr(sp)
        keep.append(sp)
    tabs = [np.ascontiguousarray(t, dtype=np.float32) for t in tables]
    if tabs:
        ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
        r.h_forc_tables = C.cast(ptrs, C.POINTER(C.c_void_p))
        r.nforc, r.forc_ld = len(tabs), int(tabs[0].shape[1])
        for k, t in enumerate(tabs):
            assert t.shape[1] == r.forc_ld
            r.forc_dt_hours[k], r.forc_nT[k] = float(dt_hours[k]), int(t.shape[0])
        keep += [tabs, ptrs]
    if cells is not None:
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        assert cells.shape == (ns,)
        r.h_forc_col = _ptr(cells)
        keep.append(cells)
    if devices is None:
        nd, dev_arr = int(num_devices if num_devices is not None else 1), None
    else:
        nd = len(devices)
        dev_arr = (C.c_int * nd)(*[int(d) for d in devices])
    yf = np.empty((ns, neq))
    stiff = np.empty(ns, dtype=np.int32)
    na, nacc = np.empty(ns, dtype=np.uint32), np.empty(ns, dtype=np.uint32)
    st = _lib.ShardedStats()
    mode = {"rccl": _lib.GATHER_RCCL, "host": _lib.GATHER_HOST}[gather]
    check(lib().hlm_run_rk45_sharded(C.byref(r), int(num_shards), dev_arr, nd, mode, _ptr(yf), _ptr(stiff), _ptr(na), _ptr(nacc), C.byref(st)))
    del keep
    return {"y_final": yf, "stiff": stiff, "n_attempt": na, "n_accept": nacc, "dense": None}, _sharded_stats(st)
