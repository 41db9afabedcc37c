"""ctypes binding of libhlm_hip.so (include/hlm_hip.h).

The library is built in-tree by ``tiger_hlm_gpu_amd/csrc/Makefile`` (see
``__graft_entry__.build``).  There is no fallback: if the shared object is missing the
import of any compute entry point raises, and every call that needs a device fails with
``HlmError`` when no HIP device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HLM_HIP_LIB: another build of the same library (kernel experiments: tools/dense_probe.sh)
LIB_PATH = os.environ.get("HLM_HIP_LIB") or os.path.join(_HERE, "lib", "libhlm_hip.so")

HLM_NEQ = 5
HLM_MAX_NEQ = 8
HLM_MAX_FORCINGS = 16
UID_DUMMY = 0
UID_204 = 204
UID_TOY3 = 3
MATH_STRICT = 0
MATH_FAST = 1
SYS_OK, SYS_STIFF, SYS_CAPPED = 0, 1, 2
PATH_COUNTERS = 7  # HLM_PATH_COUNTERS (hlm_device_job.d_path_counts)


class HlmError(RuntimeError):
    """Raised for a non-zero hlm_status (the reference throws std::runtime_error,
    rk45_api.hpp:86-108,150-153; model_registry.cpp:33-38)."""

    def __init__(self, code, msg):
        super().__init__(f"hlm_status {code}: {msg}")
        self.code = code


class GpuBuffers(C.Structure):
    """hlm_gpu_buffers -- the tuple setup_gpu_buffers returns (rk45_api.hpp:110-115)."""
    _fields_ = [
        ("d_y0_all", C.c_void_p), ("d_y_final_all", C.c_void_p), ("d_query_times", C.c_void_p),
        ("d_dense_all", C.c_void_p), ("d_stiff", C.c_void_p), ("d_n_attempt", C.c_void_p),
        ("d_n_accept", C.c_void_p), ("num_systems", C.c_int), ("num_queries", C.c_int), ("neq", C.c_int), ("dense_layout", C.c_int),
        ("dense_ncomp", C.c_int), ("dense_comp", C.c_int * HLM_MAX_NEQ),
    ]


class DeviceJob(C.Structure):
    """hlm_device_job"""
    _fields_ = [
        ("uid", C.c_int), ("num_systems", C.c_int), ("num_queries", C.c_int),
        ("t0", C.c_double), ("tf", C.c_double),
        ("d_y0", C.c_void_p), ("d_query", C.c_void_p), ("d_sp", C.c_void_p),
        ("d_forc", C.c_void_p), ("d_forc_col", C.c_void_p), ("forc_ld", C.c_int64),
        ("forc_dt_hours", C.c_double * HLM_MAX_FORCINGS), ("forc_nT", C.c_uint64 * HLM_MAX_FORCINGS),
        ("nforc", C.c_int),
        ("params", C.c_double * 6), ("fixed_step", C.c_double), ("attempt_cap", C.c_uint32),
        ("math_mode", C.c_int),
        ("d_y_final", C.c_void_p), ("d_dense", C.c_void_p), ("d_stiff", C.c_void_p),
        ("d_n_attempt", C.c_void_p), ("d_n_accept", C.c_void_p),
        ("query_begin", C.c_int), ("query_end", C.c_int), ("d_resume", C.c_void_p), ("resume_in", C.c_int),
        ("dense_tiled", C.c_int), ("use_t_stop", C.c_int), ("t_stop", C.c_double),
        ("forc_first_index", C.c_uint64 * HLM_MAX_FORCINGS), ("forc_rows", C.c_uint64 * HLM_MAX_FORCINGS),
        ("dense_ncomp", C.c_int), ("dense_comp", C.c_int * HLM_MAX_NEQ),
        ("d_flag_state", C.c_void_p), ("time_pieces", C.c_int), ("queue_run_pieces", C.c_int), ("queue_group_tiles", C.c_int), ("dense_sub_windows", C.c_int),
        ("d_path_counts", C.c_void_p), ("queue_hybrid", C.c_int),
    ]


class StreamStats(C.Structure):
    """hlm_stream_stats"""
    _fields_ = [("wall_ms", C.c_double), ("kernel_ms", C.c_double), ("copy_ms", C.c_double),
                ("windows", C.c_int), ("dense_bytes", C.c_uint64)]


class ShardedRun(C.Structure):
    """hlm_sharded_run"""
    _fields_ = [("uid", C.c_int), ("num_systems", C.c_int), ("t0", C.c_double), ("tf", C.c_double),
                ("h_y0", C.c_void_p), ("h_sp", C.c_void_p), ("h_forc_tables", C.POINTER(C.c_void_p)),
                ("forc_dt_hours", C.c_double * HLM_MAX_FORCINGS), ("forc_nT", C.c_uint64 * HLM_MAX_FORCINGS),
                ("nforc", C.c_int), ("forc_ld", C.c_int64), ("h_forc_col", C.c_void_p)]


class ShardedStats(C.Structure):
    """hlm_sharded_stats"""
    _fields_ = [("upload_ms", C.c_double), ("integrate_ms", C.c_double), ("wall_to_integrated_ms", C.c_double), ("gather_ms", C.c_double), ("download_ms", C.c_double),
                ("comm_init_ms", C.c_double), ("gather_bytes", C.c_uint64), ("rccl_version", C.c_int), ("num_shards", C.c_int),
                ("num_devices", C.c_int), ("rccl_nranks", C.c_int), ("upload_mode", C.c_int), ("device_ids", C.c_int * 16),
                ("upload_ms_device", C.c_double * 16), ("integrate_ms_device", C.c_double * 16), ("dense_bytes", C.c_uint64),
                ("dense_kernel_ms", C.c_double)]


class GatherPiece(C.Structure):
    """hlm_gather_piece"""
    _fields_ = [("shard", C.c_int), ("from_rank", C.c_int), ("array", C.c_int), ("dst_offset_bytes", C.c_uint64), ("bytes", C.c_uint64)]


UPLOAD_SINGLE_THREAD, UPLOAD_THREADS, UPLOAD_STAGED_THREADS = 0, 1, 2
# hlm_shard_window_sink: int (*)(void* user, int shard, int sys_begin, int sys_count, int query_begin, int query_end, const double* h_rows)
SHARD_WINDOW_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double))


GATHER_HOST, GATHER_RCCL = 0, 1

# hlm_window_sink: int (*)(void* user, int query_begin, int query_end, const double* h_rows)
WINDOW_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double))

# name -> (restype, argtypes); every symbol include/hlm_hip.h declares
SIGNATURES = {
    "hlm_abi_version": (C.c_int, []),
    "hlm_last_error": (C.c_char_p, []),
    "hlm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "hlm_set_device": (C.c_int, [C.c_int]),
    "hlm_device_synchronize": (C.c_int, []),
    "hlm_device_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "hlm_device_free": (C.c_int, [C.c_void_p]),
    "hlm_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "hlm_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "hlm_set_model_parameters": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "hlm_get_model_parameters": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "hlm_set_fixed_step": (C.c_int, [C.c_int, C.c_double]),
    "hlm_set_attempt_cap": (C.c_int, [C.c_uint32]),
    "hlm_set_math_mode": (C.c_int, [C.c_int]),
    "hlm_set_radau_fallback": (C.c_int, [C.c_int]),
    "hlm_set_radau_pieces": (C.c_int, [C.c_int]),
    "hlm_set_output_states": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "hlm_set_time_pieces": (C.c_int, [C.c_int]),
    "hlm_set_queue_shape": (C.c_int, [C.c_int, C.c_int]),
    "hlm_set_queue_hybrid": (C.c_int, [C.c_int]),
    "hlm_time_pieces_for": (C.c_int, [C.POINTER(DeviceJob)]),
    "hlm_set_forcings": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int,
                                   C.c_int64, C.c_void_p]),
    "hlm_set_forcings_host": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int,
                                        C.c_int64, C.c_void_p, C.c_double]),
    "hlm_clear_forcings": (C.c_int, []),
    "hlm_setup_gpu_buffers": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(GpuBuffers)]),
    "hlm_setup_gpu_buffers_model": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(GpuBuffers)]),
    "hlm_model_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hlm_launch_rk45_kernel": (C.c_int, [C.c_int, C.POINTER(GpuBuffers), C.c_double, C.c_double, C.c_void_p]),
    "hlm_retrieve_and_free": (C.c_int, [C.POINTER(GpuBuffers), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "hlm_run_rk45": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hlm_rk45_integrate_device": (C.c_int, [C.POINTER(DeviceJob), C.c_void_p]),
    "hlm_radau_fallback_device": (C.c_int, [C.POINTER(DeviceJob), C.c_void_p]),
    "hlm_resume_bytes": (C.c_size_t, [C.c_int]),
    "hlm_resume_bytes_model": (C.c_size_t, [C.c_int, C.c_int]),
    "hlm_rk45_integrate_streamed": (C.c_int, [C.POINTER(DeviceJob), C.c_int, C.c_void_p, C.c_void_p,
                                              C.POINTER(StreamStats)]),
    "hlm_rk45_integrate_streamed_sink": (C.c_int, [C.POINTER(DeviceJob), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.POINTER(StreamStats)]),
    "hlm_run_rk45_streamed": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hlm_rk45_integrate_forcing_streamed": (C.c_int, [C.POINTER(DeviceJob), C.POINTER(C.c_void_p), C.c_double,
                                                      C.c_void_p, C.POINTER(C.c_int)]),
    "hlm_set_stream_path": (C.c_int, [C.c_int]),
    "hlm_host_alloc_pinned": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "hlm_host_free_pinned": (C.c_int, [C.c_void_p]),
    "hlm_rk45_time_device": (C.c_int, [C.POINTER(DeviceJob), C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "hlm_selftest_math": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "hlm_selftest_rhs": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p]),
    "hlm_selftest_attempt": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "hlm_run_rk45_sharded": (C.c_int, [C.POINTER(ShardedRun), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.POINTER(ShardedStats)]),
    "hlm_shard_bounds": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hlm_gather_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(GatherPiece), C.POINTER(C.c_int)]),
    "hlm_set_sharded_upload": (C.c_int, [C.c_int]),
    "hlm_run_rk45_sharded_streamed": (C.c_int, [C.POINTER(ShardedRun), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(ShardedStats)]),
    "hlm_kernel_info": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib = None


def lib():
    """Load libhlm_hip.so (once).  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C tiger_hlm_gpu_amd/csrc)")
        # torch bundles its own libamdhip64/libhsa-runtime64.  Whichever HIP runtime is
        # mapped first serves every later DT_NEEDED of that soname, and two runtimes in one
        # process cannot both own the device: load torch's first so that the library and
        # torch (device memory, streams, torch.distributed) share ONE runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise HlmError(rc, lib().hlm_last_error().decode("utf-8", "replace"))
