// libhlm_hip.so: the C ABI of include/hlm_hip.h on top of the gfx950 RK45 kernels.
// Host side mirrors the reference's rk45_api (src/solver/rk45_api.hpp:53-315) and model
// registry (src/model_registry.cpp:18-60); no oracle / CPU fallback exists in here --
// without a HIP device every compute entry point fails with HLM_ERR_NO_DEVICE/HIP.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "hlm_hip.h"
#include "hlm_radau_kernel.h"
#include "hlm_rk45_dense_kernel.h"
#include "hlm_rk45_kernel.h"
#include "hlm_rk45_queue_kernel.h"
#include "hlm_gather_rccl.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HLM_HIP_TRY(expr)                                                                 \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(HLM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

// ---- process-global registry: the reference's devParams (model_204_global.cu:42) and
// forcing symbols (forcing_data.cu:5-10), kept host-side and passed as kernel arguments.
struct ModelState {
    int uid;
    double p[6];
    double fixed_step;
};
std::mutex g_mu;
ModelState g_models[3] = {
    {HLM_UID_DUMMY, {0.01, 1e-6, 1e-9, 0.9, 0.2, 10.0}, 0.0},  // model_204.hpp:22-30 defaults
    {HLM_UID_204, {0.01, 1e-6, 1e-9, 0.9, 0.2, 10.0}, 0.0},
    {HLM_UID_TOY3, {0.01, 1e-6, 1e-9, 0.9, 0.2, 10.0}, 0.0},
};

// ---- the registered device models (hlm_models.h).  Adding a model = a struct there + a line here +
// a host-side struct under include/hlm/models/ (the reference: README.md "Add a New Model",
// model_registry.cpp:18-60).  f is called with a value of the model's type; R() for an unknown UID.
template <class R, class F>
R with_model(int uid, F&& f)
{
    switch (uid) {
        case HLM_UID_204: return f(hlm::Model204Dev{});
        case HLM_UID_DUMMY: return f(hlm::DummyDev{});
        case HLM_UID_TOY3: return f(hlm::ToyDev{});
        default: return R();
    }
}
int model_neq(int uid) { return with_model<int>(uid, [](auto m) { return decltype(m)::N_EQ; }); }   // 0: unknown UID
int model_nf(int uid) { return with_model<int>(uid, [](auto m) { return decltype(m)::NF; }); }
bool model_needs_sp(int uid) { return uid == HLM_UID_204; }
uint32_t g_cap = 1u << 24;
int g_stream_path = 0;  // hlm_set_stream_path(): 0 copy engine (default), 1 / 2 transposing transfer kernel
constexpr int g_stream_layout = hlm::DENSE_TILE_COMP;  // device layout of the tiled stream path
// Default arithmetic of the host API: FAST (the measured-more-accurate and 2.4x faster of the two,
// DESIGN.md "Math modes").  Until hlm_set_math_mode() is called the choice is automatic: FAST wherever
// its one precondition holds (1e-60 <= atol <= 1e50), STRICT otherwise -- a reference-style caller
// with any tolerances the reference accepts gets a result, and with the usual ones the fast kernel.
int g_math = HLM_MATH_FAST;
bool g_math_explicit = false;
int g_time_pieces = -1;  // hlm_set_time_pieces(): < 0 = automatic (auto_time_pieces)
int g_queue_run_pieces = 0, g_queue_group_tiles = 0;  // hlm_set_queue_shape(): hlm_device_job.queue_run_pieces / queue_group_tiles of the host API's launches
int g_queue_hybrid = 0;  // hlm_set_queue_hybrid(): hlm_device_job.queue_hybrid of the host API's launches
constexpr int kMaxTimePieces = 4096;
// The automatic launch shape of a run without dense output: from 4 096 hillslopes on, one piece per twelve simulated
// hours, at most 32 -- and none at all when that makes fewer than 8: FEW pieces cost more than the drain they remove
// (measured, 1 M / 100 000 hillslopes: 2 pieces +3 / +17 %, 4 pieces -1 / +5 %, 8 pieces -3 / -3 %, 16 pieces -4 /
// -7 %, 32 pieces -5 / -10 % on 8- to 30-day windows; profiles/r03_launch_shape_probe.txt).  Since the no-rain path
// takes its linear stores in closed form and every lane picks its path by itself, the pieces pay at every size
// measured (4 096 ... 1 M hillslopes: -4 ... -13 %): they re-align the lanes of a wave in simulated time, and waves
// whose lanes sit on both sides of a storm's edge run both paths.
int auto_time_pieces(int num_systems, double t0, double tf)
{
    if (num_systems < 4096 || !(tf > t0)) return 0;
    const double p = (tf - t0) / 720.0;
    return p >= 32.0 ? 32 : (p >= 8.0 ? (int)p : 0);
}
// The automatic number of sub-windows of a dense launch in the tile layout (hlm_device_job.dense_sub_windows = 0):
// four when the launch has fewer than five rounds of waves (2 048 resident two-wave-per-SIMD slots on an MI355X; 640 000
// hillslopes) and at least 64 queries per sub-window, else one wave per tile.  Measured, 1 440 queries x 1 day, same box
// (profiles/r04_dense_subwindows_probe.txt): 200 000 hillslopes 3.74 -> 3.22 ms (n = 5), 500 000 (config 5's share per
// GPU) 7.39 -> 7.21 ms = 0.487 -> 0.500 of the HBM spec, 1 000 000 13.67 -> 13.87 ms: every item pays a prologue, an
// epilogue and a drained ring, which only a launch with few rounds earns back at its end.
int auto_dense_sub_windows(int num_systems, int window_queries)
{
    const long long tiles = ((long long)num_systems + hlm::WAVE - 1) / hlm::WAVE;
    if (tiles >= 5ll * 2048ll || window_queries < 4 * 64) return 0;
    return 4;
}
int g_sharded_upload = HLM_UPLOAD_THREADS;  // hlm_set_sharded_upload()
int g_radau_pieces = 0;  // hlm_set_radau_pieces(): time pieces of the Radau pass's work queue, 0 = automatic
int g_radau = 0;  // hlm_set_radau_fallback(): host API re-integrates flagged systems with Radau IIA
// hlm_set_output_states(): the dense rows of the host API carry only these state components (0 = all)
int g_out_ncomp = 0;
int g_out_comp[HLM_MAX_NEQ] = {};

struct ForcingState {
    const float* d_forc = nullptr;
    const int32_t* d_col = nullptr;
    int64_t ld = 0;
    double dt_h[HLM_MAX_FORCINGS] = {};
    uint64_t nT[HLM_MAX_FORCINGS] = {};
    int nforc = 0;
    const float* h_tables[HLM_MAX_FORCINGS] = {};  // hlm_set_forcings_host: series kept on the host
    double chunk_minutes = 0.0;                     // > 0: stream them in time chunks
};
// One registration per HIP device (d_forc is a device pointer, and a host process may drive several
// GPUs, one thread each, through the host-buffer API) -- and, like the reference's global symbols,
// ONE host-API integration in flight per device: hlm_launch_rk45_kernel marks the slot busy while it
// runs and a registration change on a busy slot is refused (HLM_ERR_STATE) instead of racing with it.
constexpr int kMaxDevices = 64;
struct ForcingSlot {
    ForcingState f;
    int in_flight = 0;
};
ForcingSlot g_forc_dev[kMaxDevices];
// slot of the calling thread's current device; NULL (with the error text set) when there is none
ForcingSlot* forc_slot()
{
    int d = 0;
    const hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(HLM_ERR_HIP, "hipGetDevice failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    if (d < 0 || d >= kMaxDevices) {
        fail(HLM_ERR_HIP, "device index %d outside the registry (%d slots)", d, kMaxDevices);
        return nullptr;
    }
    return &g_forc_dev[d];
}

ModelState* find_model(int uid)
{
    for (auto& m : g_models)
        if (m.uid == uid) return &m;
    return nullptr;
}

using KernelFn = void (*)(const hlm::Rk45Args);

template <class Model, int MATH, bool RES>
KernelFn pick_kernel2(bool dense, bool fixed)
{
    using namespace hlm;
    if (dense) return fixed ? rk45_multi_kernel<Model, MATH, true, true, RES> : rk45_multi_kernel<Model, MATH, true, false, RES>;
    return fixed ? rk45_multi_kernel<Model, MATH, false, true, RES> : rk45_multi_kernel<Model, MATH, false, false, RES>;
}

template <class Model>
KernelFn pick_kernel(int math, bool dense, bool fixed, bool resumable)
{
    using namespace hlm;
    if (math == HLM_MATH_FAST)
        return resumable ? pick_kernel2<Model, MATH_FAST, true>(dense, fixed) : pick_kernel2<Model, MATH_FAST, false>(dense, fixed);
    return resumable ? pick_kernel2<Model, MATH_STRICT, true>(dense, fixed) : pick_kernel2<Model, MATH_STRICT, false>(dense, fixed);
}

// Output selection (hlm_device_job.dense_ncomp): the SEL kernel variants, resumable form only (it also serves a
// plain one-launch job: no resume buffer, t_stop = +inf)
template <class Model>
KernelFn pick_selecting(int math, bool fixed, bool ring)
{
    using namespace hlm;
    constexpr int R = HLM_DENSE_RING_ROWS;
    if (ring) {
        if (math == HLM_MATH_FAST) return fixed ? rk45_dense_ring_kernel<Model, MATH_FAST, true, true, R, true> : rk45_dense_ring_kernel<Model, MATH_FAST, false, true, R, true>;
        return fixed ? rk45_dense_ring_kernel<Model, MATH_STRICT, true, true, R, true> : rk45_dense_ring_kernel<Model, MATH_STRICT, false, true, R, true>;
    }
    if (math == HLM_MATH_FAST) return fixed ? rk45_multi_kernel<Model, MATH_FAST, true, true, true, true> : rk45_multi_kernel<Model, MATH_FAST, true, false, true, true>;
    return fixed ? rk45_multi_kernel<Model, MATH_STRICT, true, true, true, true> : rk45_multi_kernel<Model, MATH_STRICT, true, false, true, true>;
}
KernelFn pick_selecting(int uid, int math, bool fixed, bool ring)
{
    return with_model<KernelFn>(uid, [&](auto m) { return pick_selecting<decltype(m)>(math, fixed, ring); });
}

// doubles per dense row of a job: the selected components, or all N_EQ
int job_row_width(const hlm_device_job* j) { return j->dense_ncomp > 0 ? j->dense_ncomp : model_neq(j->uid); }
// a selection that really leaves components out (listing all of them is the plain kernel)
bool job_selects(const hlm_device_job* j) { return j->dense_ncomp > 0 && j->dense_ncomp < model_neq(j->uid); }

int validate_selection(int uid, int n, const int* comps)
{
    const int neq = model_neq(uid);
    if (n < 0 || n > neq) return fail(HLM_ERR_INVALID_ARGUMENT, "output selection: %d components, model %d has %d", n, uid, neq);
    for (int k = 0; k < n; ++k) {
        if (comps[k] < 0 || comps[k] >= neq)
            return fail(HLM_ERR_INVALID_ARGUMENT, "output selection: %d is not a state of model %d (N_EQ = %d)", comps[k], uid, neq);
        if (k > 0 && comps[k] <= comps[k - 1])
            return fail(HLM_ERR_INVALID_ARGUMENT, "output selection must be ascending and distinct (position %d: %d after %d)", k, comps[k], comps[k - 1]);
    }
    return HLM_SUCCESS;
}

// the dense kernels with wave-synchronous stores (hlm_rk45_dense_kernel.h), layout [tile][q][comp][lane]
template <class Model>
KernelFn pick_dense_ring(int math, bool fixed, bool res)
{
    using namespace hlm;
    constexpr int R = HLM_DENSE_RING_ROWS;
    if (math == HLM_MATH_FAST) {
        if (res) return fixed ? rk45_dense_ring_kernel<Model, MATH_FAST, true, true, R> : rk45_dense_ring_kernel<Model, MATH_FAST, false, true, R>;
        return fixed ? rk45_dense_ring_kernel<Model, MATH_FAST, true, false, R> : rk45_dense_ring_kernel<Model, MATH_FAST, false, false, R>;
    }
    if (res) return fixed ? rk45_dense_ring_kernel<Model, MATH_STRICT, true, true, R> : rk45_dense_ring_kernel<Model, MATH_STRICT, false, true, R>;
    return fixed ? rk45_dense_ring_kernel<Model, MATH_STRICT, true, false, R> : rk45_dense_ring_kernel<Model, MATH_STRICT, false, false, R>;
}

KernelFn pick_dense_ring(int uid, int math, bool fixed, bool res)
{
    return with_model<KernelFn>(uid, [&](auto m) { return pick_dense_ring<decltype(m)>(math, fixed, res); });
}

// the dense ring kernel as a persistent grid over (sub-window, tile) items (hlm_rk45_dense_kernel.h): adaptive steps
using DenseQueueFn = void (*)(const hlm::Rk45Args, unsigned int*, int);
template <class Model>
DenseQueueFn pick_dense_queue(int math, bool sel)
{
    using namespace hlm;
    constexpr int R = HLM_DENSE_RING_ROWS;
    if (math == HLM_MATH_FAST) return sel ? rk45_dense_queue_kernel<Model, MATH_FAST, false, R, true> : rk45_dense_queue_kernel<Model, MATH_FAST, false, R, false>;
    return sel ? rk45_dense_queue_kernel<Model, MATH_STRICT, false, R, true> : rk45_dense_queue_kernel<Model, MATH_STRICT, false, R, false>;
}
DenseQueueFn pick_dense_queue(int uid, int math, bool sel)
{
    return with_model<DenseQueueFn>(uid, [&](auto m) { return pick_dense_queue<decltype(m)>(math, sel); });
}

// the persistent kernels over a queue of (time piece, tile) items (hlm_rk45_queue_kernel.h): no dense output
using QueueFn = void (*)(const hlm::Rk45Args, unsigned int*, int, const hlm::QueueRuns);
template <class Model>
QueueFn pick_queue(int math, bool fixed, bool retile, bool count = false, bool hybrid = false)
{
    using namespace hlm;
    if (hybrid && !count && !fixed && math == HLM_MATH_FAST) {   // (FAST, adaptive, a model with per-hillslope RHS forms; else the plain instances)
        if constexpr (Model::HAS_NORAINPATH && Model::NORAIN_LINEAR > 0)
            return retile ? rk45_queue_kernel<Model, MATH_FAST, false, true, false, true> : rk45_queue_kernel<Model, MATH_FAST, false, false, false, true>;
    }
    // (the counting instance: FAST, adaptive, tiles as items -- validate_job refuses d_path_counts on anything else)
    if (count) {   // (only a model with per-hillslope RHS forms has anything to count: Model 204)
        if constexpr (Model::HAS_NORAINPATH && Model::NORAIN_LINEAR > 0) return rk45_queue_kernel<Model, MATH_FAST, false, false, true>;
        else return nullptr;
    }
    if (retile && !fixed)   // (the re-tiling instances: adaptive steps only -- a fixed-step run never flags)
        return math == HLM_MATH_FAST ? rk45_queue_kernel<Model, MATH_FAST, false, true> : rk45_queue_kernel<Model, MATH_STRICT, false, true>;
    if (math == HLM_MATH_FAST) return fixed ? rk45_queue_kernel<Model, MATH_FAST, true> : rk45_queue_kernel<Model, MATH_FAST, false>;
    return fixed ? rk45_queue_kernel<Model, MATH_STRICT, true> : rk45_queue_kernel<Model, MATH_STRICT, false>;
}
QueueFn pick_queue(int uid, int math, bool fixed, bool retile, bool count = false, bool hybrid = false)
{
    return with_model<QueueFn>(uid, [&](auto m) { return pick_queue<decltype(m)>(math, fixed, retile, count, hybrid); });
}

KernelFn pick(int uid, int math, bool dense, bool fixed, bool resumable = false)
{
    return with_model<KernelFn>(uid, [&](auto m) { return pick_kernel<decltype(m)>(math, dense, fixed, resumable); });
}

// a job needs the resumable kernel variant when it carries state between launches or holds only
// a time chunk of a forcing series
bool job_is_resumable(const hlm_device_job* j)
{
    if (j->d_resume || j->use_t_stop || j->resume_in) return true;
    if (j->query_end != 0 && (j->query_begin != 0 || j->query_end != j->num_queries)) return true;
    for (int k = 0; k < j->nforc; ++k)
        if (j->forc_first_index[k] != 0 || (j->forc_rows[k] != 0 && j->forc_rows[k] != j->forc_nT[k])) return true;
    return false;
}

int validate_job(const hlm_device_job* j)
{
    if (!j) return fail(HLM_ERR_INVALID_ARGUMENT, "job is NULL");
    if (model_neq(j->uid) == 0) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", j->uid);
    if (j->num_systems < 0 || j->num_queries < 0)
        return fail(HLM_ERR_INVALID_ARGUMENT, "negative num_systems/num_queries");
    if (j->num_systems == 0) return HLM_SUCCESS;
    if (!j->d_y0 || !j->d_y_final || !j->d_stiff)
        return fail(HLM_ERR_INVALID_ARGUMENT, "d_y0 / d_y_final / d_stiff must not be NULL");
    if (j->num_queries > 0 && (!j->d_query || !j->d_dense))
        return fail(HLM_ERR_INVALID_ARGUMENT, "num_queries > 0 needs d_query and d_dense");
    if (model_needs_sp(j->uid) && !j->d_sp)
        return fail(HLM_ERR_INVALID_ARGUMENT, "Model 204 needs d_sp (SpatialParams)");
    if (j->nforc < 0 || j->nforc > HLM_MAX_FORCINGS)
        return fail(HLM_ERR_INVALID_ARGUMENT, "nforc %d outside [0,%d]", j->nforc, HLM_MAX_FORCINGS);
    // A model reads its first NF forcings; further registered ones are ignored, as the reference's rhs
    // ignores them (model_204.hpp:82-83 reads F[0], F[1] and nothing else), and missing ones read as 0.
    const bool reads_forcings = model_nf(j->uid) > 0;
    if (j->nforc > 0 && reads_forcings) {
        if (!j->d_forc) return fail(HLM_ERR_INVALID_ARGUMENT, "nforc > 0 needs d_forc");
        if (j->forc_ld <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "forc_ld must be > 0");
        if (!j->d_forc_col && j->forc_ld < j->num_systems)
            return fail(HLM_ERR_INVALID_ARGUMENT,
                        "forc_ld (%lld) < num_systems (%d) without a column map",
                        (long long)j->forc_ld, j->num_systems);
        for (int k = 0; k < j->nforc; ++k) {
            if (j->forc_nT[k] == 0) return fail(HLM_ERR_INVALID_ARGUMENT, "forc_nT[%d] == 0", k);
            if (!(j->forc_dt_hours[k] > 0.0))
                return fail(HLM_ERR_INVALID_ARGUMENT, "forc_dt_hours[%d] must be > 0", k);
        }
    }
    if (j->math_mode != HLM_MATH_STRICT && j->math_mode != HLM_MATH_FAST)
        return fail(HLM_ERR_INVALID_ARGUMENT, "math_mode %d", j->math_mode);
    if (j->fixed_step < 0.0) return fail(HLM_ERR_INVALID_ARGUMENT, "fixed_step < 0");
    // FAST forms the error norm with ONE reciprocal of the product of the five per-component
    // tolerances (>= atol each): that product must stay inside the double range
    if (j->math_mode == HLM_MATH_FAST && j->fixed_step == 0.0 && !(j->params[2] >= 1e-60 && j->params[2] <= 1e50))
        return fail(HLM_ERR_INVALID_ARGUMENT, "math mode FAST needs atol >= 1e-60 and <= 1e50 (got %g); use HLM_MATH_STRICT", j->params[2]);
    if (j->query_end != 0 || j->query_begin != 0) {
        if (j->query_begin < 0 || j->query_end <= j->query_begin || j->query_end > j->num_queries)
            return fail(HLM_ERR_INVALID_ARGUMENT, "query window [%d,%d) outside [0,%d)", j->query_begin,
                        j->query_end, j->num_queries);
        if ((j->query_begin != 0 || j->query_end != j->num_queries) && !j->d_resume)
            return fail(HLM_ERR_INVALID_ARGUMENT, "a partial query window needs d_resume");
    }
    if (j->resume_in && !j->d_resume) return fail(HLM_ERR_INVALID_ARGUMENT, "resume_in without d_resume");
    if (j->dense_tiled < 0 || j->dense_tiled > 2) return fail(HLM_ERR_INVALID_ARGUMENT, "dense_tiled must be 0, 1 or 2");
    if (const int rcs = validate_selection(j->uid, j->dense_ncomp, j->dense_comp); rcs != HLM_SUCCESS) return rcs;
    if (j->dense_tiled == 1) return fail(HLM_ERR_INVALID_ARGUMENT, "dense_tiled = 1 (the tile-row layout of rounds 1-4) was removed in ABI 5: use 0 or 2");
    if (j->time_pieces > kMaxTimePieces) return fail(HLM_ERR_INVALID_ARGUMENT, "time_pieces must be at most %d (negative = automatic)", kMaxTimePieces);
    // (the queue kernel numbers its (piece, tile) items with 32 bits)
    if (j->time_pieces > 1 && ((long long)j->num_systems + 63) / 64 * (long long)j->time_pieces >= (1ll << 32) - 1)
        return fail(HLM_ERR_INVALID_ARGUMENT, "time_pieces %d x %d hillslopes: more than 2^32 (piece, tile) items", j->time_pieces, j->num_systems);
    if (j->queue_run_pieces < -kMaxTimePieces) return fail(HLM_ERR_INVALID_ARGUMENT, "queue_run_pieces must be >= %d (0 = automatic, -n = automatic with runs of at most n pieces)", -kMaxTimePieces);
    if (j->dense_sub_windows < 0 || j->dense_sub_windows > 4096) return fail(HLM_ERR_INVALID_ARGUMENT, "dense_sub_windows must be within [0, 4096]");
    if (j->queue_group_tiles < 0 || j->queue_group_tiles > hlm::QUEUE_MAX_GROUP)
        return fail(HLM_ERR_INVALID_ARGUMENT, "queue_group_tiles must be within [0, %d]", hlm::QUEUE_MAX_GROUP);
    if (j->d_path_counts &&
        (j->math_mode != HLM_MATH_FAST || j->fixed_step > 0.0 || j->queue_group_tiles > 1 || j->num_queries > 0))
        return fail(HLM_ERR_INVALID_ARGUMENT, "d_path_counts: the counting instance is the queue kernel in FAST arithmetic with adaptive "
                                              "steps, no dense output and tiles as items (queue_group_tiles <= 1)");
    if (j->queue_hybrid < -1 || j->queue_hybrid > 1) return fail(HLM_ERR_INVALID_ARGUMENT, "queue_hybrid must be -1, 0 or 1");
    if (j->use_t_stop && !j->d_resume) return fail(HLM_ERR_INVALID_ARGUMENT, "t_stop needs d_resume");
    if (j->use_t_stop && !(j->t_stop == j->t_stop)) return fail(HLM_ERR_INVALID_ARGUMENT, "t_stop is NaN");
    for (int k = 0; reads_forcings && k < j->nforc; ++k)
        if (j->forc_first_index[k] + (j->forc_rows[k] ? j->forc_rows[k] : j->forc_nT[k]) > j->forc_nT[k] ||
            j->forc_first_index[k] >= j->forc_nT[k])
            return fail(HLM_ERR_INVALID_ARGUMENT, "forcing %d: chunk [%llu, +%llu) outside its %llu samples", k,
                        (unsigned long long)j->forc_first_index[k], (unsigned long long)j->forc_rows[k],
                        (unsigned long long)j->forc_nT[k]);
    return HLM_SUCCESS;
}

hlm::Rk45Args make_args(const hlm_device_job* j)
{
    hlm::Rk45Args a;
    std::memset(&a, 0, sizeof(a));
    a.y0 = j->d_y0;
    a.query = j->d_query;
    a.sp = j->d_sp;
    a.forc = j->d_forc;
    a.forc_col = j->d_forc_col;
    a.forc_ld = j->forc_ld;
    unsigned long long base = 0;
    for (int k = 0; k < j->nforc; ++k) {
        const unsigned long long rows = j->forc_rows[k] ? j->forc_rows[k] : j->forc_nT[k];
        if (k < hlm::HLM_KERNEL_MAX_NF) {
            a.forc_dtmin[k] = j->forc_dt_hours[k] * 60.0;  // rk45_kernel.cu:90
            a.forc_nT[k] = j->forc_nT[k];
            a.forc_base[k] = base;
            a.forc_first[k] = j->forc_first_index[k];
            a.forc_rows[k] = rows;
        }
        base += rows * (unsigned long long)j->forc_ld;
    }
    a.t_stop = j->use_t_stop ? j->t_stop : __builtin_inf();
    a.nforc = j->nforc;
    a.ns = j->num_systems;
    a.nq = j->num_queries;
    a.t0 = j->t0;
    a.tf = j->tf;
    a.h0 = j->params[0];
    a.rtol = j->params[1];
    a.atol = j->params[2];
    a.safety = j->params[3];
    a.min_scale = j->params[4];
    a.max_scale = j->params[5];
    a.fixed_step = j->fixed_step;
    a.cap = j->attempt_cap ? j->attempt_cap : (1u << 24);
    a.y_final = j->d_y_final;
    a.dense = j->d_dense;
    a.dense_tiled = j->dense_tiled;
    a.q_begin = j->query_end ? j->query_begin : 0;
    a.q_end = j->query_end ? j->query_end : j->num_queries;
    a.resume = static_cast<double*>(j->d_resume);
    a.resume_in = j->resume_in;
    a.stiff = j->d_stiff;
    a.n_attempt = j->d_n_attempt;
    a.n_accept = j->d_n_accept;
    a.flag_state = j->d_flag_state;
    a.path_counts = reinterpret_cast<unsigned long long*>(j->d_path_counts);
    if (job_selects(j)) {
        a.dense_k = j->dense_ncomp;
        for (int c = 0; c < HLM_MAX_NEQ; ++c) a.dense_pos[c] = -1;
        for (int k = 0; k < j->dense_ncomp; ++k) a.dense_pos[j->dense_comp[k]] = k;
    }
    return a;
}

#ifdef HLM_QUEUE_DEBUG
// experiment builds only (not in include/hlm_hip.h): where the last queue launch keeps its state, for a look at a
// kernel that does not finish (tools/queue_hang_probe.py)
char* g_debug_queue_pool = nullptr;
size_t g_debug_queue_rbytes = 0, g_debug_queue_qbytes = 0;
#endif

// The runs of pieces a queue launch hands out (hlm_device_job.queue_run_pieces; hlm_rk45_queue_kernel.h "RUNS").
// Automatic: with more tiles than resident waves every run takes half of what is left of the window, at most a
// quarter and at least a sixteenth of it (32 pieces: 8, 8, 8, 4, 2, 2): the launch's tail is that of its last
// items, and five hand-overs instead of 31 leave a third of the HBM traffic (profiles/r04_queue_runs_probe.txt).  The
// cap is what workloads with a spread of tile costs need: a tile's pieces are serial and the queue is in order, so
// a first run of half the window (round 4's first rule: 16, 8, 4, 2, 2) starts the tiles late in the order late --
// headline 129.8 ms either way, wet surface 217.7 -> 210.3 ms, SURVEY's cold cells 122.5 -> 118.9 ms (113.0 with every
// piece handed over: queue_run_pieces = 1; -n = this schedule with runs of at most n pieces).  A launch with no more
// tiles than resident waves hands every piece over: its waves sit unevenly on the SIMDs (1 563 waves on 1 024
// SIMDs), the queue evens that out piece by piece, and its bytes are few (100 000 hillslopes, cold cells: 20.8 ms
// against 23.1 ms as one run per tile).
hlm::QueueRuns make_queue_runs(int pieces, long long tiles, long long resident, int run_pieces)
{
    hlm::QueueRuns r;
    std::memset(&r, 0, sizeof(r));
#ifdef HLM_QUEUE_PROF
    if (const char* e = std::getenv("HLM_QUEUE_RUNS")) {   // experiment builds: run lengths "16,8,4,2,2"
        int b = 0;
        for (const char* q = e; *q && r.nruns < hlm::QUEUE_MAX_RUNS;) {
            const int len = std::atoi(q);
            if (len <= 0 || b >= pieces) break;
            r.begin_[r.nruns++] = b;
            b += len;
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
        if (b < pieces && r.nruns < hlm::QUEUE_MAX_RUNS) r.begin_[r.nruns++] = b;
        r.begin_[r.nruns] = pieces;
        return r;
    }
#endif
    // run_pieces < 0: the automatic schedule with no run longer than |run_pieces| pieces (uniform runs where the
    // shrinking schedule would need more than QUEUE_MAX_RUNS entries)
    const int cap = run_pieces < 0 ? -run_pieces : (pieces / 4 > 1 ? pieces / 4 : 1);
    const int least = pieces / 16 > 1 ? pieces / 16 : 1;
    if (run_pieces < 0 && (cap <= least || (pieces + cap - 1) / cap + 4 > hlm::QUEUE_MAX_RUNS - 1)) run_pieces = cap;
    if (run_pieces > 0) {
        r.uniform = run_pieces < pieces ? run_pieces : pieces;
        r.nruns = (pieces + r.uniform - 1) / r.uniform;
        return r;
    }
    if (tiles <= resident) {
        r.uniform = 1;
        r.nruns = pieces;
        return r;
    }
    int b = 0;
    while (b < pieces && r.nruns < hlm::QUEUE_MAX_RUNS - 1) {
        const int rem = pieces - b;
        int len = rem / 2 > least ? rem / 2 : least;
        if (len > cap) len = cap;
        if (len > rem) len = rem;
        r.begin_[r.nruns++] = b;
        b += len;
    }
    if (b < pieces) r.begin_[r.nruns++] = b;   // (never with least = pieces / 16: at most 6 runs)
    r.begin_[r.nruns] = pieces;
    return r;
}

// Persistent launch over a queue of (run of time pieces, tile) items (hlm_rk45_queue_kernel.h): as many one-wave
// workgroups as the device holds at once; the lane states between the runs live in a stream-ordered scratch allocation
// -- or, for a launch that integrates the window [w0, w1] of a longer run (w1 < tf: lanes pause there; resume_in: they
// come from an earlier window), in the caller's state record j->d_resume, where the next window's launch finds them.
// forc_first: the forcing block holds the sample rows from forc_first[k] on (a time chunk of the series): the block's
// base is biased so that the plain row arithmetic of the kernel lands in it.
int launch_queue(const hlm_device_job* j, int pieces, hipStream_t stream, double w0, double w1, bool windowed)
{
    const bool fixed = j->fixed_step > 0.0;
    const int group = j->queue_group_tiles > 1 && !fixed ? j->queue_group_tiles : 1;
    // queue_hybrid = 0 (automatic) is "a block per RHS form": the hybrid instances need 252 VGPRs for their four blocks (two
    // waves per SIMD instead of three; capped at 168 they spill 117 and lose everywhere), which costs the workloads whose
    // static stores never overflow 18 % (headline 127.4 -> 149.9 ms, wet surface +6 %) and gains 14-19 % where the storms
    // fill them (369.8 -> 317.0 ms, 261.4 -> 212.2 ms with 8 tiles packed per piece: profiles/r05_hybrid_ab.txt) -- a choice
    // only the caller, who knows the domain, can make (hlm_set_queue_hybrid / rk45_solver --hybrid).
    QueueFn fn = pick_queue(j->uid, j->math_mode, fixed, group > 1, j->d_path_counts != nullptr, j->queue_hybrid > 0);
    if (!fn) return fail(HLM_ERR_UNKNOWN_MODEL, "no kernel for UID %d%s", j->uid, j->d_path_counts ? " (d_path_counts: the model has no per-hillslope RHS forms to count)" : "");
    if (windowed && !j->d_resume) return fail(HLM_ERR_INVALID_ARGUMENT, "a window launch needs d_resume");
    int dev = 0, cus = 0, per_cu = 0;
    HLM_HIP_TRY(hipGetDevice(&dev));
    HLM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HLM_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), hlm::WAVE, 0));
    const long long resident = (long long)cus * (per_cu > 0 ? per_cu : 1);
    const long long tiles = ((long long)j->num_systems + hlm::WAVE - 1) / hlm::WAVE;
    const unsigned blocks = (unsigned)(tiles < resident ? tiles : resident);
    const size_t rbytes = windowed ? 0 : (hlm_resume_bytes_model(j->uid, j->num_systems) + 15u) & ~(size_t)15u;
    const size_t qbytes = sizeof(unsigned int) * ((size_t)tiles + hlm::QUEUE_MAX_GROUP + 2);  // next item | finished runs per tile (passes per group) | abort word (behind the padded tiles)
    char* pool = nullptr;
    HLM_HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&pool), rbytes + qbytes, stream));
    unsigned int* queue = reinterpret_cast<unsigned int*>(pool + rbytes);
    {
        const unsigned int words = (unsigned int)(qbytes / sizeof(unsigned int));
        hipLaunchKernelGGL(hlm::queue_init_kernel, dim3((words + 255u) / 256u), dim3(256), 0, stream, queue, words);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) {
            (void)hipFreeAsync(pool, stream);
            return fail(HLM_ERR_HIP, "queue_init_kernel failed: %s", hipGetErrorString(e));
        }
    }
    hlm::Rk45Args a = make_args(j);
    if (windowed) {
        a.resume = static_cast<double*>(j->d_resume);
        a.resume_in = j->resume_in;
        a.t_stop = w1 < j->tf ? w1 : __builtin_inf();
        unsigned long long base = 0;   // the chunk's rows: bias every forcing's base by its first row (unsigned wrap-around)
        for (int k = 0; k < j->nforc; ++k) {
            const unsigned long long rows = j->forc_rows[k] ? j->forc_rows[k] : j->forc_nT[k];
            if (k < hlm::HLM_KERNEL_MAX_NF) a.forc_base[k] = base - j->forc_first_index[k] * (unsigned long long)j->forc_ld;
            base += rows * (unsigned long long)j->forc_ld;
        }
    } else {
        a.resume = reinterpret_cast<double*>(pool);
        a.resume_in = 0;
        a.t_stop = __builtin_inf();
    }
#ifdef HLM_QUEUE_DEBUG
    g_debug_queue_pool = pool;
    g_debug_queue_rbytes = rbytes;
    g_debug_queue_qbytes = qbytes;
#endif
    // groups of tiles packed at the start of every run (queue_group_tiles): every piece handed over unless the job
    // names a run length or a cap of its own
    hlm::QueueRuns runs = make_queue_runs(pieces, tiles, resident, group > 1 && j->queue_run_pieces == 0 ? 1 : j->queue_run_pieces);
    runs.group = group;
    if (group > 1)   // test hook: sibling passes of a group staggered in time (hlm_rk45_queue_kernel.h)
        if (const char* e = std::getenv("HLM_QUEUE_STAGGER_TICKS")) {
            const unsigned long v = std::strtoul(e, nullptr, 10);
            runs.stagger_ticks = (unsigned int)(v < 1000000ul ? v : 1000000ul);   // (at most 10 ms per pass index: a stray variable cannot hang a launch)
        }
    runs.t_begin = w0;
    runs.piece_len = (w1 - w0) / (double)pieces;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(hlm::WAVE), 0, stream, a, queue, pieces, runs);
    const hipError_t launched = hipGetLastError();
    HLM_HIP_TRY(hipFreeAsync(pool, stream));  // stream-ordered: after the kernel
    HLM_HIP_TRY(launched);
    return HLM_SUCCESS;
}
int launch_queue(const hlm_device_job* j, int pieces, hipStream_t stream) { return launch_queue(j, pieces, stream, j->t0, j->tf, false); }

// pieces of the queue kernel this job launches with; 0 = the fixed map
int job_time_pieces(const hlm_device_job* j)
{
    if (j->num_queries > 0 || job_is_resumable(j)) return 0;
    const int pieces = j->time_pieces < 0 ? auto_time_pieces(j->num_systems, j->t0, j->tf) : j->time_pieces;
    return pieces > 1 ? pieces : 0;
}

// sub-windows of a dense launch in the tile layout (hlm_device_job.dense_sub_windows; 0 = the automatic rule)
int job_dense_sub_windows(const hlm_device_job* j)
{
    if (j->num_queries <= 0 || j->dense_tiled != hlm::DENSE_TILE_COMP || j->fixed_step > 0.0) return 0;
    if (j->use_t_stop || j->forc_first_index[0] != 0) return 0;   // (time chunks / forcing chunks: one wave per tile)
    for (int k = 0; k < j->nforc; ++k)
        if (j->forc_first_index[k] != 0 || (j->forc_rows[k] != 0 && j->forc_rows[k] != j->forc_nT[k])) return 0;
    const int qb = j->query_end ? j->query_begin : 0, qe = j->query_end ? j->query_end : j->num_queries;
    int n = j->dense_sub_windows;
    if (n == 0) n = auto_dense_sub_windows(j->num_systems, qe - qb);
    if (n > qe - qb) n = qe - qb;
    return n > 1 ? n : 0;
}

// The dense ring kernel as a persistent grid over (sub-window, tile) items.  The lane states between the sub-windows
// live in the caller's record when the job has one (a query window of a streamed run: the launch also starts from it
// and leaves the paused lanes in it), else in a stream-ordered scratch allocation.
int launch_dense_queue(const hlm_device_job* j, int nsub, hipStream_t stream)
{
    DenseQueueFn fn = pick_dense_queue(j->uid, j->math_mode, job_selects(j));
    if (!fn) return fail(HLM_ERR_UNKNOWN_MODEL, "no kernel for UID %d", j->uid);
    int dev = 0, cus = 0, per_cu = 0;
    HLM_HIP_TRY(hipGetDevice(&dev));
    HLM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    HLM_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), hlm::WAVE, 0));
    const long long resident = (long long)cus * (per_cu > 0 ? per_cu : 1);
    const long long tiles = ((long long)j->num_systems + hlm::WAVE - 1) / hlm::WAVE;
    const unsigned blocks = (unsigned)(tiles < resident ? tiles : resident);
    const size_t rbytes = j->d_resume ? 0 : (hlm_resume_bytes_model(j->uid, j->num_systems) + 15u) & ~(size_t)15u;
    const size_t qbytes = sizeof(unsigned int) * ((size_t)tiles + 2);  // next item | finished sub-windows per tile | abort word
    char* pool = nullptr;
    HLM_HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&pool), rbytes + qbytes, stream));
    unsigned int* queue = reinterpret_cast<unsigned int*>(pool + rbytes);
    {
        const unsigned int words = (unsigned int)(qbytes / sizeof(unsigned int));
        hipLaunchKernelGGL(hlm::queue_init_kernel, dim3((words + 255u) / 256u), dim3(256), 0, stream, queue, words);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) {
            (void)hipFreeAsync(pool, stream);
            return fail(HLM_ERR_HIP, "queue_init_kernel failed: %s", hipGetErrorString(e));
        }
    }
    hlm::Rk45Args a = make_args(j);
    if (!j->d_resume) {
        a.resume = reinterpret_cast<double*>(pool);
        a.resume_in = 0;
    }
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(hlm::WAVE), 0, stream, a, queue, nsub);
    const hipError_t launched = hipGetLastError();
    HLM_HIP_TRY(hipFreeAsync(pool, stream));  // stream-ordered: after the kernel
    HLM_HIP_TRY(launched);
    return HLM_SUCCESS;
}

int launch(const hlm_device_job* j, hipStream_t stream)
{
    const bool dense = j->num_queries > 0;
    const bool fixed = j->fixed_step > 0.0;
    if (j->d_path_counts && job_time_pieces(j) <= 1)
        return fail(HLM_ERR_INVALID_ARGUMENT, "d_path_counts: this job does not launch in time pieces (set time_pieces > 1)");
    if (const int nsub = job_dense_sub_windows(j); nsub > 1) return launch_dense_queue(j, nsub, stream);
    if (const int pieces = job_time_pieces(j); pieces > 1) return launch_queue(j, pieces, stream);
    KernelFn fn = dense && job_selects(j) ? pick_selecting(j->uid, j->math_mode, fixed, j->dense_tiled == hlm::DENSE_TILE_COMP)
                  : dense && j->dense_tiled == hlm::DENSE_TILE_COMP
                      ? pick_dense_ring(j->uid, j->math_mode, fixed, job_is_resumable(j))
                      : pick(j->uid, j->math_mode, dense, fixed, job_is_resumable(j));
    if (!fn) return fail(HLM_ERR_UNKNOWN_MODEL, "no kernel for UID %d", j->uid);
    const hlm::Rk45Args a = make_args(j);
    const unsigned blocks = (unsigned)((j->num_systems + hlm::WAVE - 1) / hlm::WAVE);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(hlm::WAVE), 0, stream, a);
    HLM_HIP_TRY(hipGetLastError());
    return HLM_SUCCESS;
}

// tile-major window -> host order, for the layout and N_EQ at hand (hlm_rk45_kernel.h: dense_tile_to_host)
void launch_tile_to_host(int /*layout: DENSE_TILE_COMP*/, int neq, dim3 grid, hipStream_t stream, const double* src, double* dst, int ns, int nq,
                         int q_begin, int W)
{
    auto go = [&](auto n) {
        constexpr int N = decltype(n)::value;
        hipLaunchKernelGGL((hlm::dense_tile_to_host<N>), grid, dim3(256), 0, stream, src, dst, ns, nq, q_begin, W);
    };
    switch (neq) {  // neq = doubles per row: a model's N_EQ, or the width of an output selection
        case 1: go(std::integral_constant<int, 1>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 3: go(std::integral_constant<int, 3>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 5: go(std::integral_constant<int, 5>{}); break;
        case 6: go(std::integral_constant<int, 6>{}); break;
        case 7: go(std::integral_constant<int, 7>{}); break;
        default: go(std::integral_constant<int, 8>{}); break;
    }
    static_assert(HLM_MAX_NEQ == 8, "one instance per possible row width");
}

// Radau-IIA pass over the systems the RK45 launch of the same job flagged (stream-ordered after it)
int launch_radau(const hlm_device_job* j, hipStream_t stream)
{
    if (job_is_resumable(j))
        return fail(HLM_ERR_INVALID_ARGUMENT,
                    "the Radau fallback restarts from t0: it needs the whole query range and forcing series "
                    "in one launch (no d_resume / t_stop / query window / forcing chunk)");
    if (j->fixed_step > 0.0) return HLM_SUCCESS;  // fixed-step runs never flag
    const int ns = j->num_systems;
    // the flagged systems, compacted IN ORDER: neighbours share a forcing cell and take similar numbers of
    // implicit attempts, so the lanes of a Radau wave finish together (an atomic append scatters them)
    const int nb = (ns + hlm::COMPACT_BLOCK - 1) / hlm::COMPACT_BLOCK;
    // the pass's work queue (hlm_radau_kernel.h "TIME PIECES"): items (piece of the window, list entry), kRadauPieces
    // equal pieces; a system paused at a piece's end waits in a record by list entry (y, t, h | two counters, the next query)
    constexpr int kRadauPieces = 16;
    const int neq = model_neq(j->uid);
    const bool by_flag_time = j->d_flag_state != nullptr;
    // ints: [0 .. ns) the list | [nb] block counts | count, base, next item | [kRadauPieces] entries per piece | [ns] done
    //       | [3][ns] record | ([ns] time bins);  doubles behind them: [neq + 3 + NF][ns] record
    const size_t n_int = (size_t)ns + (size_t)nb + 3 + (size_t)kRadauPieces + (size_t)ns + 3 * (size_t)ns + (by_flag_time ? (size_t)ns : 0);
    const size_t int_bytes = (n_int * sizeof(int) + 15) / 16 * 16;
    char* scratch_raw = nullptr;
    HLM_HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&scratch_raw), int_bytes + sizeof(double) * (size_t)(neq + 3 + hlm::HLM_KERNEL_MAX_NF) * (size_t)ns, stream));
    int* const scratch = reinterpret_cast<int*>(scratch_raw);
    int* const list = scratch;
    int* const block_count = scratch + ns;
    int* const count = block_count + nb;
    int* const next_item = count + 2;   // (count + 1 = the appending passes' base)
    int* const cum = next_item + 1;
    int* const done = cum + kRadauPieces;
    int* const rec_i32 = done + ns;
    int* const bin = rec_i32 + 3 * (size_t)ns;
    double* const rec_f64 = reinterpret_cast<double*>(scratch_raw + int_bytes);
    // count, base, next item, the entries per piece and every list entry's counter of finished pieces start at 0
    if (const hipError_t ez = hipMemsetAsync(count, 0, sizeof(int) * (3 + (size_t)kRadauPieces + (size_t)ns), stream); ez != hipSuccess) {
        (void)hipFreeAsync(scratch_raw, stream);
        return fail(HLM_ERR_HIP, "Radau launch set-up failed: %s", hipGetErrorString(ez));
    }
    if (!by_flag_time) {
        hipLaunchKernelGGL(hlm::compact_count_kernel, dim3((unsigned)nb), dim3(hlm::COMPACT_BLOCK), 0, stream, j->d_stiff, (int)HLM_SYS_STIFF, ns, block_count);
        hipLaunchKernelGGL(hlm::compact_scan_kernel, dim3(1), dim3(hlm::COMPACT_BLOCK), 0, stream, block_count, nb, count, (int*)nullptr, (int*)nullptr);
        hipLaunchKernelGGL(hlm::compact_scatter_kernel, dim3((unsigned)nb), dim3(hlm::COMPACT_BLOCK), 0, stream, j->d_stiff, (int)HLM_SYS_STIFF, ns, block_count, list);
    } else {
        // "continue" mode: the work left is tf - t_flag.  The list is built piece by piece of the flag time (earliest
        // first: the longest jobs start first), in index order within a piece -- kRadauPieces stable compaction passes
        // appending to one list, no host round trip; cum[p] = the entries flagged in the pieces up to p, i.e. the list
        // entries that exist in piece p of the queue
        int* const base = count + 1;
        hipLaunchKernelGGL(hlm::radau_flag_bin_kernel, dim3((unsigned)nb), dim3(hlm::COMPACT_BLOCK), 0, stream, j->d_stiff, j->d_flag_state,
                           neq, ns, j->t0, j->tf, kRadauPieces, bin);
        for (int b = 0; b < kRadauPieces; ++b) {
            hipLaunchKernelGGL(hlm::compact_count_kernel, dim3((unsigned)nb), dim3(hlm::COMPACT_BLOCK), 0, stream, bin, b, ns, block_count);
            hipLaunchKernelGGL(hlm::compact_scan_kernel, dim3(1), dim3(hlm::COMPACT_BLOCK), 0, stream, block_count, nb, count, base, cum + b);
            hipLaunchKernelGGL(hlm::compact_scatter_kernel, dim3((unsigned)nb), dim3(hlm::COMPACT_BLOCK), 0, stream, bin, b, ns, block_count, list);
        }
    }
    hlm::Rk45Args a = make_args(j);
    // an implicit attempt costs ~30 us of a wave: the default cap of this pass is 2^20 attempts per
    // system (half a minute), not the RK45 loop's 2^24, so a hopeless system cannot hold a wave for long
    if (!j->attempt_cap) a.cap = 1u << 20;
    using RadauFn = void (*)(const hlm::Rk45Args, const int*, const int*, int*, const hlm::RadauQueue);
    const bool fast = j->math_mode == HLM_MATH_FAST;
    RadauFn rf = with_model<RadauFn>(j->uid, [&](auto m) -> RadauFn {
        using M = decltype(m);
        return fast ? hlm::radau_fallback_kernel<M, hlm::MATH_FAST> : hlm::radau_fallback_kernel<M, hlm::MATH_STRICT>;
    });
    if (!rf) { (void)hipFreeAsync(scratch_raw, stream); return fail(HLM_ERR_UNKNOWN_MODEL, "no Radau kernel for UID %d", j->uid); }
    // a persistent grid: as many one-wave workgroups as the device holds (or as there could be full waves of flagged
    // systems); every lane draws its items from next_item
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t eq = hipGetDevice(&dev);
    if (eq == hipSuccess) eq = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (eq == hipSuccess) eq = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(rf), hlm::WAVE, 0);
    const long long resident = (long long)cus * (per_cu > 0 ? per_cu : 1), waves = ((long long)ns + hlm::WAVE - 1) / hlm::WAVE;
    const unsigned blocks = (unsigned)(waves < resident ? waves : resident);
    if (eq != hipSuccess) { (void)hipFreeAsync(scratch_raw, stream); return fail(HLM_ERR_HIP, "Radau launch set-up failed: %s", hipGetErrorString(eq)); }
    hlm::RadauQueue q{};
    // Automatic: 16 pieces when every flagged system restarts at t0 (all items are equally long: whole systems leave
    // a fifth of the launch to its tail).  When they continue from their flag states the list is ordered longest job
    // first and only the END of the launch needs small items: whole systems up to the last six of the 16 pieces, those
    // one by one (hlm_radau_kernel.h: "merged_to"; measured on the cold C3, RK45 + pass: 1 piece at the end 170.5 ms, 2: 169.0,
    // 3: 165.3, 4: 164.4, 6: 161.8, 9: 162.5, all 15: 169.6).  hlm_set_radau_pieces() forces a count for both: 1 = whole systems,
    // n > 1 = every piece an item of its own ("continue" mode then always takes the 16 bins its list is ordered by).
    static const int kRadauTailPieces = [] {   // (experiment knob: HLM_RADAU_TAIL_PIECES = 1 ... 15)
        const char* e = std::getenv("HLM_RADAU_TAIL_PIECES");
        const int v = e ? std::atoi(e) : 0;
        return v >= 1 && v <= 15 ? v : 6;
    }();
    q.npieces = g_radau_pieces > 0 ? g_radau_pieces : kRadauPieces;
    if (by_flag_time && q.npieces > 1) q.npieces = kRadauPieces;
    q.merged_to = by_flag_time && g_radau_pieces == 0 ? kRadauPieces - 1 - kRadauTailPieces : 0;
    if ((long long)ns > (1ll << hlm::RADAU_ITEM_PIECE_SHIFT)) q.npieces = 1;   // (an item's slot has 26 bits beside its piece; 16 x ns items would not fit an int either)
    if (q.npieces == 1) q.merged_to = 0;
    q.cum = by_flag_time && q.npieces > 1 ? cum : nullptr;
    q.rec_f64 = rec_f64;
    q.rec_i32 = rec_i32;
    q.done = done;
    q.cap = ns;
    hipLaunchKernelGGL(rf, dim3(blocks), dim3(hlm::WAVE), 0, stream, a, list, count, next_item, q);
    const hipError_t launched = hipGetLastError();
    HLM_HIP_TRY(hipFreeAsync(scratch_raw, stream));  // stream-ordered: after the kernels above
    HLM_HIP_TRY(launched);
    return HLM_SUCCESS;
}

// Edges t0 = e_0 < e_1 < ... < e_n = tf of a window cut into pieces of `step` minutes, as t0 + k * step (never by
// repeated addition: a step below the ulp of t would not advance, and the piece count is bounded up front).
constexpr int kMaxPieces = 1 << 20;
constexpr int kMaxTimedWindows = 4096;  // windowed pipelines keep four HIP events per window up to here
int chunk_edges(double t0, double tf, double step, std::vector<double>* edge, const char* what)
{
    edge->assign(1, t0);
    if (!(tf > t0)) return HLM_SUCCESS;  // nothing to integrate: no pieces
    const double n = std::ceil((tf - t0) / step);
    if (!(n >= 1.0) || n > (double)kMaxPieces)
        return fail(HLM_ERR_INVALID_ARGUMENT, "%s: %g minutes per piece cuts [%g, %g] into %g pieces (at most %d)", what, step, t0, tf, n, kMaxPieces);
    const int np = (int)n;
    edge->reserve((size_t)np + 1);
    for (int k = 1; k < np; ++k) {
        const double e = t0 + (double)k * step;
        if (e > edge->back() && e < tf) edge->push_back(e);
    }
    edge->push_back(tf);
    return HLM_SUCCESS;
}

int fill_job_from_registry(hlm_device_job* j, int uid, const ForcingState& F)
{
    ModelState* m = find_model(uid);
    if (!m) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    j->uid = uid;
    std::memcpy(j->params, m->p, sizeof(m->p));
    j->fixed_step = m->fixed_step;
    j->attempt_cap = g_cap;
    j->time_pieces = g_time_pieces;  // (< 0: resolved by the caller once it knows the number of systems)
    j->queue_run_pieces = g_queue_run_pieces;
    j->queue_group_tiles = g_queue_group_tiles;
    j->queue_hybrid = g_queue_hybrid;
    j->math_mode = g_math;
    if (!g_math_explicit && !(m->p[2] >= 1e-60 && m->p[2] <= 1e50)) j->math_mode = HLM_MATH_STRICT;
    j->d_forc = F.d_forc;
    j->d_forc_col = F.d_col;
    j->forc_ld = F.ld;
    j->nforc = F.nforc;
    std::memcpy(j->forc_dt_hours, F.dt_h, sizeof(F.dt_h));
    std::memcpy(j->forc_nT, F.nT, sizeof(F.nT));
    return HLM_SUCCESS;
}

// replace the calling device's registration (under g_mu); refused while an integration uses it
int replace_forcings(const ForcingState& next)
{
    ForcingSlot* slot = forc_slot();
    if (!slot) return HLM_ERR_HIP;
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot->in_flight)
        return fail(HLM_ERR_STATE, "the forcing registration of this device is in use by a running "
                                   "hlm_launch_rk45_kernel / hlm_run_rk45 (one host-API integration per device)");
    slot->f = next;
    return HLM_SUCCESS;
}

}  // namespace

extern "C" {

int hlm_abi_version(void) { return HLM_ABI_VERSION; }
const char* hlm_last_error(void) { return g_err; }

int hlm_device_count(int* count)
{
    if (!count) return fail(HLM_ERR_INVALID_ARGUMENT, "count is NULL");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(HLM_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return HLM_SUCCESS;
}

int hlm_set_device(int device)
{
    HLM_HIP_TRY(hipSetDevice(device));
    return HLM_SUCCESS;
}

int hlm_device_synchronize(void)
{
    HLM_HIP_TRY(hipDeviceSynchronize());
    return HLM_SUCCESS;
}

int hlm_device_malloc(void** dptr, size_t bytes)
{
    if (!dptr) return fail(HLM_ERR_INVALID_ARGUMENT, "dptr is NULL");
    *dptr = nullptr;
    if (bytes == 0) return HLM_SUCCESS;
    HLM_HIP_TRY(hipMalloc(dptr, bytes));
    return HLM_SUCCESS;
}

int hlm_device_free(void* dptr)
{
    if (dptr) HLM_HIP_TRY(hipFree(dptr));
    return HLM_SUCCESS;
}

int hlm_memcpy_h2d(void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return HLM_SUCCESS;
    if (!dst || !src) return fail(HLM_ERR_INVALID_ARGUMENT, "NULL pointer in h2d copy");
    HLM_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return HLM_SUCCESS;
}

int hlm_memcpy_d2h(void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return HLM_SUCCESS;
    if (!dst || !src) return fail(HLM_ERR_INVALID_ARGUMENT, "NULL pointer in d2h copy");
    HLM_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return HLM_SUCCESS;
}

// ---- registry ------------------------------------------------------------------------
int hlm_set_model_parameters(int uid, const double p[6])
{
    if (!p) return fail(HLM_ERR_INVALID_ARGUMENT, "p is NULL");
    std::lock_guard<std::mutex> lk(g_mu);
    ModelState* m = find_model(uid);
    if (!m) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    if (!(p[0] > 0.0)) return fail(HLM_ERR_INVALID_ARGUMENT, "initialStep must be > 0");
    if (!(p[1] >= 0.0) || !(p[2] >= 0.0) || !(p[1] + p[2] > 0.0))
        return fail(HLM_ERR_INVALID_ARGUMENT, "rtol/atol must be >= 0 and not both 0");
    if (!(p[4] > 0.0) || !(p[5] >= p[4]))
        return fail(HLM_ERR_INVALID_ARGUMENT, "need 0 < minScale <= maxScale");
    std::memcpy(m->p, p, sizeof(m->p));
    return HLM_SUCCESS;
}

int hlm_get_model_parameters(int uid, double p[6])
{
    if (!p) return fail(HLM_ERR_INVALID_ARGUMENT, "p is NULL");
    std::lock_guard<std::mutex> lk(g_mu);
    ModelState* m = find_model(uid);
    if (!m) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    std::memcpy(p, m->p, sizeof(m->p));
    return HLM_SUCCESS;
}

int hlm_set_fixed_step(int uid, double h)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ModelState* m = find_model(uid);
    if (!m) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    if (!(h >= 0.0)) return fail(HLM_ERR_INVALID_ARGUMENT, "fixed step must be >= 0");
    m->fixed_step = h;
    return HLM_SUCCESS;
}

int hlm_set_attempt_cap(uint32_t cap)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_cap = cap ? cap : (1u << 24);
    return HLM_SUCCESS;
}

int hlm_set_stream_path(int path)
{
    if (path != 0 && path != 2) return fail(HLM_ERR_INVALID_ARGUMENT, "stream path %d (0 copy engine, 2 transposing transfer kernel; 1 was removed in ABI 5)", path);
    std::lock_guard<std::mutex> lk(g_mu);
    g_stream_path = path ? 1 : 0;
    return HLM_SUCCESS;
}

int hlm_set_math_mode(int mode)
{
    if (mode != HLM_MATH_STRICT && mode != HLM_MATH_FAST)
        return fail(HLM_ERR_INVALID_ARGUMENT, "math mode %d", mode);
    std::lock_guard<std::mutex> lk(g_mu);
    g_math = mode;
    g_math_explicit = true;
    return HLM_SUCCESS;
}

int hlm_set_forcings(const float* d_forc, const double* dt_hours, const uint64_t* nT, int nforc,
                     int64_t ld, const int32_t* d_col)
{
    if (nforc < 0 || nforc > HLM_MAX_FORCINGS)
        return fail(HLM_ERR_INVALID_ARGUMENT, "nforc %d outside [0,%d]", nforc, HLM_MAX_FORCINGS);
    if (nforc > 0 && (!d_forc || !dt_hours || !nT || ld <= 0))
        return fail(HLM_ERR_INVALID_ARGUMENT, "forcings need d_forc, dt_hours, nT and ld > 0");
    ForcingState F;
    F.d_forc = d_forc;
    F.d_col = d_col;
    F.ld = ld;
    F.nforc = nforc;
    for (int k = 0; k < nforc; ++k) {
        F.dt_h[k] = dt_hours[k];
        F.nT[k] = nT[k];
    }
    return replace_forcings(F);
}

int hlm_set_forcings_host(const float* const* h_tables, const double* dt_hours, const uint64_t* nT, int nforc,
                          int64_t ld, const int32_t* d_col, double chunk_minutes)
{
    if (nforc <= 0 || nforc > HLM_MAX_FORCINGS || !h_tables || !dt_hours || !nT || ld <= 0 || !(chunk_minutes > 0.0 || chunk_minutes < 0.0))
        return fail(HLM_ERR_INVALID_ARGUMENT, "set_forcings_host: need 1..%d host tables, dt, nT, ld > 0, chunk > 0", HLM_MAX_FORCINGS);
    ForcingState F;
    F.d_col = d_col;
    F.ld = ld;
    F.nforc = nforc;
    F.chunk_minutes = chunk_minutes;
    for (int k = 0; k < nforc; ++k) {
        if (!h_tables[k]) return fail(HLM_ERR_INVALID_ARGUMENT, "h_tables[%d] is NULL", k);
        F.h_tables[k] = h_tables[k];
        F.dt_h[k] = dt_hours[k];
        F.nT[k] = nT[k];
    }
    return replace_forcings(F);
}

int hlm_clear_forcings(void) { return replace_forcings(ForcingState{}); }

// ---- device-resident core --------------------------------------------------------------
int hlm_rk45_integrate_device(const hlm_device_job* job, void* hip_stream)
{
    int rc = validate_job(job);
    if (rc != HLM_SUCCESS) return rc;
    if (job->num_systems == 0) return HLM_SUCCESS;
    return launch(job, static_cast<hipStream_t>(hip_stream));
}

int hlm_radau_fallback_device(const hlm_device_job* job, void* hip_stream)
{
    int rc = validate_job(job);
    if (rc != HLM_SUCCESS) return rc;
    if (job->num_systems == 0) return HLM_SUCCESS;
    return launch_radau(job, static_cast<hipStream_t>(hip_stream));
}

int hlm_set_output_states(int n, const int* comps)
{
    if (n < 0 || n > HLM_MAX_NEQ || (n > 0 && !comps)) return fail(HLM_ERR_INVALID_ARGUMENT, "output states: n = %d (0..%d) with a list", n, HLM_MAX_NEQ);
    for (int k = 0; k < n; ++k) {
        if (comps[k] < 0 || comps[k] >= HLM_MAX_NEQ) return fail(HLM_ERR_INVALID_ARGUMENT, "output states: component %d", comps[k]);
        if (k > 0 && comps[k] <= comps[k - 1]) return fail(HLM_ERR_INVALID_ARGUMENT, "output states must be ascending and distinct");
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_out_ncomp = n;
    for (int k = 0; k < HLM_MAX_NEQ; ++k) g_out_comp[k] = k < n ? comps[k] : 0;
    return HLM_SUCCESS;
}

int hlm_time_pieces_for(const hlm_device_job* job)
{
    if (!job) return 0;
    return job_time_pieces(job);
}

int hlm_set_time_pieces(int pieces)
{
    if (pieces > kMaxTimePieces) return fail(HLM_ERR_INVALID_ARGUMENT, "time pieces %d (at most %d)", pieces, kMaxTimePieces);
    std::lock_guard<std::mutex> lk(g_mu);
    g_time_pieces = pieces < 0 ? -1 : pieces;
    return HLM_SUCCESS;
}

int hlm_set_queue_shape(int run_pieces, int group_tiles)
{
    if (run_pieces < -kMaxTimePieces || run_pieces > kMaxTimePieces)
        return fail(HLM_ERR_INVALID_ARGUMENT, "queue_run_pieces %d outside [%d, %d]", run_pieces, -kMaxTimePieces, kMaxTimePieces);
    if (group_tiles < 0 || group_tiles > hlm::QUEUE_MAX_GROUP)
        return fail(HLM_ERR_INVALID_ARGUMENT, "queue_group_tiles must be within [0, %d]", hlm::QUEUE_MAX_GROUP);
    std::lock_guard<std::mutex> lk(g_mu);
    g_queue_run_pieces = run_pieces;
    g_queue_group_tiles = group_tiles;
    return HLM_SUCCESS;
}

int hlm_set_queue_hybrid(int mode)
{
    if (mode < -1 || mode > 1) return fail(HLM_ERR_INVALID_ARGUMENT, "queue_hybrid must be -1, 0 or 1");
    std::lock_guard<std::mutex> lk(g_mu);
    g_queue_hybrid = mode;
    return HLM_SUCCESS;
}

int hlm_set_radau_pieces(int pieces)
{
    if (pieces < 0 || pieces > 16) return fail(HLM_ERR_INVALID_ARGUMENT, "radau pieces %d (0 = automatic, 1 ... 16)", pieces);
    g_radau_pieces = pieces;
    return HLM_SUCCESS;
}

int hlm_set_radau_fallback(int enabled)
{
    if (enabled < 0 || enabled > 2) return fail(HLM_ERR_INVALID_ARGUMENT, "radau fallback mode %d (0 off, 1 restart at t0, 2 continue from the flag state)", enabled);
    std::lock_guard<std::mutex> lk(g_mu);
    g_radau = enabled;
    return HLM_SUCCESS;
}

size_t hlm_resume_bytes_model(int uid, int ns)
{
    const int neq = model_neq(uid);
    return ns <= 0 || neq == 0 ? 0 : (size_t)ns * (hlm::resume_f64(neq) * sizeof(double) + hlm::RESUME_I32 * sizeof(int32_t));
}
size_t hlm_resume_bytes(int ns) { return hlm_resume_bytes_model(HLM_UID_204, ns); }

int hlm_model_info(int uid, int* n_eq, int* n_forcings)
{
    if (model_neq(uid) == 0) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    if (n_eq) *n_eq = model_neq(uid);
    if (n_forcings) *n_forcings = model_nf(uid);
    return HLM_SUCCESS;
}

int hlm_host_alloc_pinned(void** hptr, size_t bytes)
{
    if (!hptr) return fail(HLM_ERR_INVALID_ARGUMENT, "hptr is NULL");
    *hptr = nullptr;
    if (bytes == 0) return HLM_SUCCESS;
    HLM_HIP_TRY(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return HLM_SUCCESS;
}

int hlm_host_free_pinned(void* hptr)
{
    if (hptr) HLM_HIP_TRY(hipHostFree(hptr));
    return HLM_SUCCESS;
}

int hlm_rk45_integrate_streamed(const hlm_device_job* job, int window_queries, double* h_dense,
                                void* hip_stream, hlm_stream_stats* stats)
{
    if (!job) return fail(HLM_ERR_INVALID_ARGUMENT, "job is NULL");
    hlm_device_job probe = *job;  // d_dense / window fields of the caller's job are ignored
    probe.d_dense = reinterpret_cast<double*>(sizeof(double));
    probe.query_begin = probe.query_end = probe.resume_in = 0;
    probe.d_resume = nullptr;
    int rc = validate_job(&probe);
    if (rc != HLM_SUCCESS) return rc;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const int ns = job->num_systems, nq = job->num_queries;
    if (nq <= 0 || ns <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "streamed dense output needs num_systems, num_queries > 0");
    if (window_queries <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "window_queries must be > 0");
    if (!h_dense) return fail(HLM_ERR_INVALID_ARGUMENT, "h_dense is NULL");
    const int W = window_queries < nq ? window_queries : nq;
    const int nwin = (nq + W - 1) / W;
    const int neq = job_row_width(job);                // doubles per row: N_EQ, or the selected components only
    const size_t row = (size_t)neq * sizeof(double);  // 40 B per (system, query) for the 5-state models

    // Path 0 (default): linear device windows [ns][W][5] + hipMemcpy2DAsync on the copy engine.
    // Path 1 (hlm_set_stream_path(1), needs pinned h_dense): the RK45 kernel writes tile-major
    // windows (coalesced; 19 ms instead of 27 ms per 57.6 GB) and dense_tile_to_host
    // transposes them straight into host memory.  Measured end to end the copy engine wins
    // (54.8 vs 51.5 GB/s): the pipeline is PCIe-bound and a PCIe-bound kernel that holds CUs
    // and memory queues slows the concurrent integration, the SDMA engine does not.
    double* h_dense_dev = nullptr;
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, h_dense) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer)
            h_dense_dev = static_cast<double*>(at.devicePointer);
        else
            (void)hipGetLastError();  // pageable memory: not an error, just the copy-engine path
    }
    const bool tiled = h_dense_dev != nullptr && g_stream_path == 1;
    const int nblk = (ns + hlm::WAVE - 1) / hlm::WAVE;
    hipStream_t compute = static_cast<hipStream_t>(hip_stream), copy = nullptr;
    double* buf[2] = {nullptr, nullptr};
    void* resume = nullptr;
    hipEvent_t done_k[2] = {nullptr, nullptr}, done_c[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev;  // per-window timing events
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };

    {
        int lo = 0, hi = 0;  // the transfer kernel must not queue behind 15 000 integration workgroups
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        ok(hipStreamCreateWithPriority(&copy, hipStreamNonBlocking, hi));
    }
    const size_t win_rows = tiled ? (size_t)nblk * hlm::WAVE : (size_t)ns;
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        if (b == 0 || nwin > 1) ok(hipMalloc((void**)&buf[b], win_rows * W * row));
        ok(hipEventCreateWithFlags(&done_k[b], hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&done_c[b], hipEventDisableTiming));
    }
    ok(hipMalloc(&resume, hlm_resume_bytes_model(job->uid, ns)));
    // per-window timing events: for up to kMaxTimedWindows windows (a run cut into more gets wall time only)
    const bool timed = nwin <= kMaxTimedWindows;
    ev.resize(timed ? (size_t)nwin * 4 : 0, nullptr);
    for (auto& x : ev) ok(hipEventCreate(&x));

    hipEvent_t wall0 = nullptr, wall1 = nullptr;
    ok(hipEventCreate(&wall0));
    ok(hipEventCreate(&wall1));
    if (e == hipSuccess) ok(hipEventRecord(wall0, compute));
    if (e == hipSuccess) ok(hipStreamSynchronize(compute));  // setup done: the clock starts here
    const auto host_t0 = std::chrono::steady_clock::now();

    for (int k = 0; k < nwin && e == hipSuccess && rc == HLM_SUCCESS; ++k) {
        const int b = k & 1, qb = k * W, qe = (qb + W < nq) ? qb + W : nq;
        hlm_device_job j = *job;
        j.d_dense = buf[b];
        j.dense_tiled = tiled ? g_stream_layout : 0;
        j.query_begin = qb;
        j.query_end = qe;
        j.d_resume = resume;
        j.resume_in = k > 0;
        if (k >= 2) ok(hipStreamWaitEvent(compute, done_c[b], 0));  // buffer b is free again
        if (timed) ok(hipEventRecord(ev[4 * k + 0], compute));
        if (e == hipSuccess) rc = launch(&j, compute);
        if (timed) ok(hipEventRecord(ev[4 * k + 1], compute));
        ok(hipEventRecord(done_k[b], compute));
        ok(hipStreamWaitEvent(copy, done_k[b], 0));
        if (timed) ok(hipEventRecord(ev[4 * k + 2], copy));
        if (tiled) {
            const int Wk = qe - qb;
            const dim3 grid((unsigned)nblk, (unsigned)((Wk + hlm::TILE_TQ - 1) / hlm::TILE_TQ));
            launch_tile_to_host(g_stream_layout, neq, grid, copy, buf[b], h_dense_dev, ns, nq, qb, Wk);
            ok(hipGetLastError());
        } else {
            // device [ns][qe-qb][K] -> host rows of nq*8K B, columns qb..qe
            ok(hipMemcpy2DAsync(reinterpret_cast<char*>(h_dense) + (size_t)qb * row, (size_t)nq * row, buf[b],
                                (size_t)(qe - qb) * row, (size_t)(qe - qb) * row, (size_t)ns,
                                hipMemcpyDeviceToHost, copy));
        }
        if (timed) ok(hipEventRecord(ev[4 * k + 3], copy));
        ok(hipEventRecord(done_c[b], copy));
    }
    if (e == hipSuccess) ok(hipStreamSynchronize(compute));
    if (e == hipSuccess) ok(hipStreamSynchronize(copy));
    const auto host_t1 = std::chrono::steady_clock::now();  // last byte is in host memory
    if (e == hipSuccess && rc == HLM_SUCCESS && stats) {
        float ms = 0.f;
        stats->windows = nwin;
        stats->dense_bytes = (uint64_t)ns * (uint64_t)nq * row;
        for (int k = 0; timed && k < nwin; ++k) {
            if (hipEventElapsedTime(&ms, ev[4 * k + 0], ev[4 * k + 1]) == hipSuccess) stats->kernel_ms += ms;
            if (hipEventElapsedTime(&ms, ev[4 * k + 2], ev[4 * k + 3]) == hipSuccess) stats->copy_ms += ms;
        }
        stats->wall_ms = std::chrono::duration<double, std::milli>(host_t1 - host_t0).count();
    }
    for (auto& x : ev) if (x) (void)hipEventDestroy(x);
    for (int b = 0; b < 2; ++b) {
        if (buf[b]) (void)hipFree(buf[b]);
        if (done_k[b]) (void)hipEventDestroy(done_k[b]);
        if (done_c[b]) (void)hipEventDestroy(done_c[b]);
    }
    if (wall0) (void)hipEventDestroy(wall0);
    if (wall1) (void)hipEventDestroy(wall1);
    if (resume) (void)hipFree(resume);
    if (copy) (void)hipStreamDestroy(copy);
    if (e != hipSuccess) return fail(HLM_ERR_HIP, "integrate_streamed: %s", hipGetErrorString(e));
    return rc;
}

// The two device windows, the two pinned staging buffers and the state record of hlm_rk45_integrate_streamed_sink, kept
// by a caller that runs several such pipelines one after the other on one device (the shards of a device in
// hlm_run_rk45_sharded_streamed: pinned allocations cost ~0.1 s per GB, 0.6 s of a 1.65 s run of 8 shards) -- they grow
// to the largest request and are freed by release().
struct StreamScratch {
    double* buf[2] = {nullptr, nullptr};
    double* hbuf[2] = {nullptr, nullptr};
    void* resume = nullptr;
    size_t win_bytes = 0, resume_bytes = 0;
    bool two = false;
    hipError_t ensure(size_t need_win, bool need_two, size_t need_resume)
    {
        hipError_t e = hipSuccess;
        if (need_win > win_bytes || (need_two && !two)) {
            const size_t nb = need_win > win_bytes ? need_win : win_bytes;
            for (int b = 0; b < 2; ++b) {
                if (buf[b]) (void)hipFree(buf[b]);
                if (hbuf[b]) (void)hipHostFree(hbuf[b]);
                buf[b] = hbuf[b] = nullptr;
            }
            two = two || need_two;
            win_bytes = nb;
            for (int b = 0; b < (two ? 2 : 1) && e == hipSuccess; ++b) {
                e = hipMalloc((void**)&buf[b], nb);
                if (e == hipSuccess) e = hipHostMalloc((void**)&hbuf[b], nb, hipHostMallocDefault);
            }
        }
        if (e == hipSuccess && need_resume > resume_bytes) {
            if (resume) (void)hipFree(resume);
            resume = nullptr;
            e = hipMalloc(&resume, need_resume);
            resume_bytes = e == hipSuccess ? need_resume : 0;
        }
        return e;
    }
    void release()
    {
        for (int b = 0; b < 2; ++b) {
            if (buf[b]) (void)hipFree(buf[b]);
            if (hbuf[b]) (void)hipHostFree(hbuf[b]);
            buf[b] = hbuf[b] = nullptr;
        }
        if (resume) (void)hipFree(resume);
        resume = nullptr;
        win_bytes = resume_bytes = 0;
        two = false;
    }
};
static int streamed_sink_impl(const hlm_device_job* job, int window_queries, hlm_window_sink sink, void* user, void* hip_stream,
                              hlm_stream_stats* stats, StreamScratch* keep);

int hlm_rk45_integrate_streamed_sink(const hlm_device_job* job, int window_queries, hlm_window_sink sink, void* user,
                                     void* hip_stream, hlm_stream_stats* stats)
{
    return streamed_sink_impl(job, window_queries, sink, user, hip_stream, stats, nullptr);
}

static int streamed_sink_impl(const hlm_device_job* job, int window_queries, hlm_window_sink sink, void* user, void* hip_stream,
                              hlm_stream_stats* stats, StreamScratch* keep)
{
    if (!job) return fail(HLM_ERR_INVALID_ARGUMENT, "job is NULL");
    hlm_device_job probe = *job;  // d_dense / window fields of the caller's job are ignored
    probe.d_dense = reinterpret_cast<double*>(sizeof(double));
    probe.query_begin = probe.query_end = probe.resume_in = 0;
    probe.d_resume = nullptr;
    int rc = validate_job(&probe);
    if (rc != HLM_SUCCESS) return rc;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const int ns = job->num_systems, nq = job->num_queries;
    if (nq <= 0 || ns <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "streamed dense output needs num_systems, num_queries > 0");
    if (window_queries <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "window_queries must be > 0");
    const int W = window_queries < nq ? window_queries : nq;
    const int nwin = (nq + W - 1) / W;
    const size_t row = (size_t)job_row_width(job) * sizeof(double);
    const size_t win_bytes = (size_t)ns * (size_t)W * row;

    hipStream_t compute = static_cast<hipStream_t>(hip_stream), copy = nullptr;
    StreamScratch own;                      // (a caller's scratch outlives this call; the call's own does not)
    StreamScratch& sc = keep ? *keep : own;
    hipEvent_t done_k[2] = {nullptr, nullptr}, done_c[2] = {nullptr, nullptr};
    const bool timed = nwin <= kMaxTimedWindows;  // per-window timing events (a run cut into more windows gets wall time only)
    std::vector<hipEvent_t> ev(timed ? (size_t)nwin * 4 : 0, nullptr);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    ok(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    ok(sc.ensure(win_bytes, nwin > 1, hlm_resume_bytes_model(job->uid, ns)));
    double* const* buf = sc.buf;    // device windows [ns][W][K] (host order: one contiguous copy each)
    double* const* hbuf = sc.hbuf;  // pinned host staging
    void* const resume = sc.resume;
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        ok(hipEventCreateWithFlags(&done_k[b], hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&done_c[b], hipEventDisableTiming));
    }
    for (auto& x : ev) ok(hipEventCreate(&x));
    if (e == hipSuccess) ok(hipStreamSynchronize(compute));  // setup done: the clock starts here
    const auto host_t0 = std::chrono::steady_clock::now();
    uint64_t delivered = 0;
    int sink_rc = 0;
    auto deliver = [&](int k) {  // window k is in hbuf[k & 1] once its copy has finished
        const int qb = k * W, qe = (qb + W < nq) ? qb + W : nq;
        if (!ok(hipEventSynchronize(done_c[k & 1]))) return;
        if (sink) sink_rc = sink(user, qb, qe, hbuf[k & 1]);
        delivered += (uint64_t)ns * (uint64_t)(qe - qb) * row;
    };
    for (int k = 0; k < nwin && e == hipSuccess && rc == HLM_SUCCESS && sink_rc == 0; ++k) {
        const int b = k & 1, qb = k * W, qe = (qb + W < nq) ? qb + W : nq;
        hlm_device_job j = *job;
        j.d_dense = buf[b];
        j.dense_tiled = 0;
        j.query_begin = qb;
        j.query_end = qe;
        j.d_resume = resume;
        j.resume_in = k > 0;
        if (k >= 2) ok(hipStreamWaitEvent(compute, done_c[b], 0));  // device window b has been copied out
        if (timed) ok(hipEventRecord(ev[4 * k + 0], compute));
        if (e == hipSuccess) rc = launch(&j, compute);
        if (timed) ok(hipEventRecord(ev[4 * k + 1], compute));
        ok(hipEventRecord(done_k[b], compute));
        ok(hipStreamWaitEvent(copy, done_k[b], 0));
        if (timed) ok(hipEventRecord(ev[4 * k + 2], copy));
        // (hbuf[b] is free: the sink of window k - 2 returned during iteration k - 1)
        ok(hipMemcpyAsync(hbuf[b], buf[b], (size_t)ns * (size_t)(qe - qb) * row, hipMemcpyDeviceToHost, copy));
        if (timed) ok(hipEventRecord(ev[4 * k + 3], copy));
        ok(hipEventRecord(done_c[b], copy));
        if (k >= 1 && e == hipSuccess && rc == HLM_SUCCESS) deliver(k - 1);
    }
    if (e == hipSuccess && rc == HLM_SUCCESS && sink_rc == 0) deliver(nwin - 1);
    (void)hipStreamSynchronize(compute);
    (void)hipStreamSynchronize(copy);
    const auto host_t1 = std::chrono::steady_clock::now();
    if (e == hipSuccess && rc == HLM_SUCCESS && stats) {
        float ms = 0.f;
        stats->windows = nwin;
        stats->dense_bytes = delivered;
        for (int k = 0; timed && k < nwin; ++k) {
            if (hipEventElapsedTime(&ms, ev[4 * k + 0], ev[4 * k + 1]) == hipSuccess) stats->kernel_ms += ms;
            if (hipEventElapsedTime(&ms, ev[4 * k + 2], ev[4 * k + 3]) == hipSuccess) stats->copy_ms += ms;
        }
        (void)hipGetLastError();
        stats->wall_ms = std::chrono::duration<double, std::milli>(host_t1 - host_t0).count();
    }
    for (auto& x : ev) if (x) (void)hipEventDestroy(x);
    for (int b = 0; b < 2; ++b) {
        if (done_k[b]) (void)hipEventDestroy(done_k[b]);
        if (done_c[b]) (void)hipEventDestroy(done_c[b]);
    }
    if (!keep) own.release();
    if (copy) (void)hipStreamDestroy(copy);
    if (e != hipSuccess) return fail(HLM_ERR_HIP, "integrate_streamed_sink: %s", hipGetErrorString(e));
    if (sink_rc != 0) return fail(HLM_ERR_STATE, "integrate_streamed_sink: the sink returned %d", sink_rc);
    return rc;
}

int hlm_rk45_integrate_forcing_streamed(const hlm_device_job* job, const float* const* h_tables,
                                        double chunk_minutes, void* hip_stream, int* chunks)
{
    if (!job || !h_tables) return fail(HLM_ERR_INVALID_ARGUMENT, "job / h_tables is NULL");
    // chunk_minutes < 0: GROWING chunks -- the first one |chunk_minutes| long, every next one twice as long up to eight
    // times that (1, 2, 4, 8, 8, ... days for -1440): the first upload, which nothing overlaps, stays short, every later
    // one (PCIe: ~1.8 ms per day of a 1 M-hillslope per-stream series) hides behind the previous chunk's integration
    // (~4.3 ms per day), and the long chunks run as launches of the queue kernel (its pieces cut from the chunk)
    // instead of one fixed-map launch per day: 6 launches instead of 30 for a 30-day window.  Device memory: two
    // buffers of the longest chunk.
    const bool growing = chunk_minutes < 0.0;
    if (growing) chunk_minutes = -chunk_minutes;
    if (!(chunk_minutes > 0.0)) return fail(HLM_ERR_INVALID_ARGUMENT, "chunk_minutes must be > 0 (or < 0: growing chunks)");
    if (job->nforc <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "no forcings to stream");
    for (int k = 0; k < job->nforc; ++k)
        if (!h_tables[k]) return fail(HLM_ERR_INVALID_ARGUMENT, "h_tables[%d] is NULL", k);
    hlm_device_job probe = *job;  // the caller's d_forc / chunk / resume fields are ignored
    probe.d_forc = reinterpret_cast<const float*>(sizeof(float));
    probe.d_resume = nullptr;
    probe.resume_in = probe.use_t_stop = 0;
    std::memset(probe.forc_first_index, 0, sizeof(probe.forc_first_index));
    std::memset(probe.forc_rows, 0, sizeof(probe.forc_rows));
    int rc = validate_job(&probe);
    if (rc != HLM_SUCCESS) return rc;
    if (chunks) *chunks = 0;
    if (job->num_systems == 0) return HLM_SUCCESS;
    const int nf = job->nforc;
    const size_t ld = (size_t)job->forc_ld;

    // chunk edges and, per chunk and forcing, the sample rows any t in [ta, tb] can ask for
    std::vector<double> edge;
    if (const int rce = chunk_edges(job->t0, job->tf, chunk_minutes, &edge, "integrate_forcing_streamed"); rce != HLM_SUCCESS) return rce;
    if (growing && edge.size() > 2) {  // keep the edges 0, 1, 3, 7, 15, 23, ... of the uniform grid
        std::vector<double> g(1, edge[0]);
        size_t at = 0, len = 1;
        while (at + len < edge.size() - 1) {
            at += len;
            g.push_back(edge[at]);
            if (len < 8) len *= 2;
        }
        g.push_back(edge.back());
        edge.swap(g);
    }
    const int nchunk = (int)edge.size() - 1;
    if (nchunk <= 0) return launch(&probe, static_cast<hipStream_t>(hip_stream)) /* tf <= t0: nothing to integrate */;
    auto span = [&](int j, double ta, double tb, uint64_t* first, uint64_t* rows) {
        const double dtm = job->forc_dt_hours[j] * 60.0;
        auto idx = [&](double t) {
            const double r = std::floor(t / dtm);
            return r < 0.0 ? (uint64_t)0 : (r >= (double)job->forc_nT[j] ? job->forc_nT[j] - 1 : (uint64_t)r);
        };
        *first = idx(ta);
        *rows = idx(tb) - *first + 1;
    };
    size_t max_rows = 0;
    for (int j = 0; j < nf; ++j) {
        uint64_t worst = 0, f, r;
        for (int k = 0; k < nchunk; ++k) { span(j, edge[k], edge[k + 1], &f, &r); worst = r > worst ? r : worst; }
        max_rows += worst;
    }

    hipStream_t compute = static_cast<hipStream_t>(hip_stream), copy = nullptr;
    float* buf[2] = {nullptr, nullptr};
    void* resume = nullptr;
    hipEvent_t uploaded[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    ok(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        if (b == 0 || nchunk > 1) ok(hipMalloc((void**)&buf[b], max_rows * ld * sizeof(float)));
        ok(hipEventCreateWithFlags(&uploaded[b], hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
    }
    ok(hipMalloc(&resume, hlm_resume_bytes_model(job->uid, job->num_systems)));
    for (int k = 0; k < nchunk && e == hipSuccess && rc == HLM_SUCCESS; ++k) {
        const int b = k & 1;
        hlm_device_job j = *job;
        if (k >= 2) ok(hipStreamWaitEvent(copy, consumed[b], 0));  // the kernel that read buffer b is done
        size_t off = 0;
        for (int f = 0; f < nf; ++f) {
            span(f, edge[k], edge[k + 1], &j.forc_first_index[f], &j.forc_rows[f]);
            ok(hipMemcpyAsync(buf[b] + off, h_tables[f] + (size_t)j.forc_first_index[f] * ld,
                              (size_t)j.forc_rows[f] * ld * sizeof(float), hipMemcpyHostToDevice, copy));
            off += (size_t)j.forc_rows[f] * ld;
        }
        ok(hipEventRecord(uploaded[b], copy));
        ok(hipStreamWaitEvent(compute, uploaded[b], 0));
        j.d_forc = buf[b];
        j.d_resume = resume;
        j.resume_in = k > 0;
        j.use_t_stop = edge[k + 1] < job->tf;
        j.t_stop = edge[k + 1];
        // a chunk long enough for the queue kernel's pieces (the automatic rule on the chunk: from 4 096 hillslopes and
        // four days on) runs in its shape, lanes from and to the same state record; shorter ones one wave per tile
        // (growing chunks: every chunk, also the short first ones, in 16 pieces -- measured on the headline workload:
        // 134.5 ms against 135.2 ms with the twelve-hour rule and 135.4 ms with 8; profiles/r04_streamed_probe.txt)
        int pieces = 0;
        if (j.num_queries == 0) {
            if (j.time_pieces >= 0) pieces = growing ? j.time_pieces : 0;
            else if (growing) pieces = j.num_systems >= 4096 ? 16 : 0;
            else pieces = auto_time_pieces(j.num_systems, edge[k], edge[k + 1]);
        }
        if (e == hipSuccess) rc = pieces > 1 ? launch_queue(&j, pieces, compute, edge[k], edge[k + 1], true) : launch(&j, compute);
        ok(hipEventRecord(consumed[b], compute));
    }
    if (e == hipSuccess) ok(hipStreamSynchronize(compute));
    if (e == hipSuccess) ok(hipStreamSynchronize(copy));
    for (int b = 0; b < 2; ++b) {
        if (buf[b]) (void)hipFree(buf[b]);
        if (uploaded[b]) (void)hipEventDestroy(uploaded[b]);
        if (consumed[b]) (void)hipEventDestroy(consumed[b]);
    }
    if (resume) (void)hipFree(resume);
    if (copy) (void)hipStreamDestroy(copy);
    if (e != hipSuccess) return fail(HLM_ERR_HIP, "integrate_forcing_streamed: %s", hipGetErrorString(e));
    if (chunks && rc == HLM_SUCCESS) *chunks = nchunk;
    return rc;
}

int hlm_rk45_time_device(const hlm_device_job* job, void* hip_stream, int iters, float* ms)
{
    if (iters < 0 || (iters > 0 && !ms)) return fail(HLM_ERR_INVALID_ARGUMENT, "iters/ms");
    int rc = validate_job(job);
    if (rc != HLM_SUCCESS) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    for (int i = 0; i < iters && e == hipSuccess && rc == HLM_SUCCESS; ++i) {
        e = hipEventRecord(e0, s);
        if (e == hipSuccess && job->num_systems > 0) rc = launch(job, s);
        if (e == hipSuccess && rc == HLM_SUCCESS) e = hipEventRecord(e1, s);
        if (e == hipSuccess && rc == HLM_SUCCESS) e = hipEventSynchronize(e1);
        if (e == hipSuccess && rc == HLM_SUCCESS) e = hipEventElapsedTime(&ms[i], e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);  // on every exit path
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(HLM_ERR_HIP, "rk45_time_device: %s", hipGetErrorString(e));
    return rc;
}

int hlm_selftest_math(int which, const double* d_in, double* d_out, int n, void* hip_stream)
{
    if (which < 0 || which > 9) return fail(HLM_ERR_INVALID_ARGUMENT, "selftest function %d", which);
    if (n < 0 || (n > 0 && (!d_in || !d_out))) return fail(HLM_ERR_INVALID_ARGUMENT, "selftest buffers");
    if (n == 0) return HLM_SUCCESS;
    hipLaunchKernelGGL(hlm::selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), which, d_in, d_out, n);
    HLM_HIP_TRY(hipGetLastError());
    return HLM_SUCCESS;
}

int hlm_selftest_rhs(int uid, int math_mode, int fast_path, const double* d_y, const void* d_sp, const float* d_F,
                     double* d_out, int32_t* d_broke, int n, void* hip_stream)
{
    if (n < 0 || (n > 0 && (!d_y || !d_F || !d_out))) return fail(HLM_ERR_INVALID_ARGUMENT, "selftest_rhs buffers");
    if (math_mode != HLM_MATH_STRICT && math_mode != HLM_MATH_FAST) return fail(HLM_ERR_INVALID_ARGUMENT, "math mode %d", math_mode);
    if (model_needs_sp(uid) && n > 0 && !d_sp) return fail(HLM_ERR_INVALID_ARGUMENT, "Model 204 needs d_sp");
    if (n == 0) return HLM_SUCCESS;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    using namespace hlm;
    const bool known = with_model<bool>(uid, [&](auto m) {
        using M = decltype(m);
        if (math_mode == HLM_MATH_FAST)
            hipLaunchKernelGGL((selftest_rhs_kernel<M, MATH_FAST>), grid, block, 0, s, d_y, d_sp, d_F, d_out, d_broke, M::HAS_FASTPATH ? fast_path : 0, n);
        else
            hipLaunchKernelGGL((selftest_rhs_kernel<M, MATH_STRICT>), grid, block, 0, s, d_y, d_sp, d_F, d_out, d_broke, M::HAS_FASTPATH ? fast_path : 0, n);
        return true;
    });
    if (!known) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    HLM_HIP_TRY(hipGetLastError());
    return HLM_SUCCESS;
}

int hlm_selftest_attempt(int uid, int math_mode, int path, int closed, const double* d_y, const void* d_sp, const float* d_F,
                         const double* d_h, double rtol, double atol, double* d_ynew, double* d_err_jump, int32_t* d_broke, int n,
                         void* hip_stream)
{
    if (n < 0 || (n > 0 && (!d_y || !d_F || !d_h || !d_ynew || !d_err_jump))) return fail(HLM_ERR_INVALID_ARGUMENT, "selftest_attempt buffers");
    if (math_mode != HLM_MATH_STRICT && math_mode != HLM_MATH_FAST) return fail(HLM_ERR_INVALID_ARGUMENT, "math mode %d", math_mode);
    if (path < HLM_ATTEMPT_SELECT_DENSE || path > 3) return fail(HLM_ERR_INVALID_ARGUMENT, "attempt path %d", path);
    if (!(rtol >= 0.0) || !(atol >= 0.0)) return fail(HLM_ERR_INVALID_ARGUMENT, "selftest_attempt tolerances");
    if (model_needs_sp(uid) && n > 0 && !d_sp) return fail(HLM_ERR_INVALID_ARGUMENT, "Model 204 needs d_sp");
    if (n == 0) return HLM_SUCCESS;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    using namespace hlm;
    const bool known = with_model<bool>(uid, [&](auto m) {
        using M = decltype(m);
        if (math_mode == HLM_MATH_FAST)
            hipLaunchKernelGGL((selftest_attempt_kernel<M, MATH_FAST>), grid, block, 0, s, d_y, d_sp, d_F, d_h, rtol, atol, path, closed, d_ynew, d_err_jump, d_broke, n);
        else
            hipLaunchKernelGGL((selftest_attempt_kernel<M, MATH_STRICT>), grid, block, 0, s, d_y, d_sp, d_F, d_h, rtol, atol, path, closed, d_ynew, d_err_jump, d_broke, n);
        return true;
    });
    if (!known) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    HLM_HIP_TRY(hipGetLastError());
    return HLM_SUCCESS;
}

int hlm_kernel_info(int uid, int dense, int math_mode, int* vgprs, int* scratch_bytes, int* lds_bytes,
                    int* max_threads)
{
    KernelFn fn = pick(uid, math_mode, dense != 0, false);
    if (!fn) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    hipFuncAttributes at;
    HLM_HIP_TRY(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(fn)));
    if (vgprs) *vgprs = at.numRegs;
    if (scratch_bytes) *scratch_bytes = (int)at.localSizeBytes;
    if (lds_bytes) *lds_bytes = (int)at.sharedSizeBytes;
    if (max_threads) *max_threads = at.maxThreadsPerBlock;
    return HLM_SUCCESS;
}

#ifdef HLM_QUEUE_MARKS
// pinned host memory for the waves' progress marks (NULL: none); returns the host pointer
int hlm_debug_queue_marks(unsigned int** host, size_t words)
{
    unsigned int* h = nullptr;
    unsigned int* d = nullptr;
    HLM_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h), words * sizeof(unsigned int), hipHostMallocMapped));
    std::memset(h, 0, words * sizeof(unsigned int));
    HLM_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0));
    HLM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(hlm::g_queue_marks), &d, sizeof(d)));
    *host = h;
    return HLM_SUCCESS;
}
#endif
#ifdef HLM_QUEUE_PROF
// experiment builds only: device buffer of 8 x 64-bit words per block for the waves' accounting (NULL: none)
int hlm_debug_queue_prof(unsigned long long* d_words)
{
    HLM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(hlm::g_queue_prof), &d_words, sizeof(d_words)));
    return HLM_SUCCESS;
}
#endif
#ifdef HLM_QUEUE_DEBUG
int hlm_debug_queue_pool(void** pool, size_t* rbytes, size_t* qbytes)
{
    *pool = g_debug_queue_pool;
    *rbytes = g_debug_queue_rbytes;
    *qbytes = g_debug_queue_qbytes;
    return 0;
}
#endif

#ifdef HLM_RADAU_COUNTERS
// experiment builds only (not in include/hlm_hip.h): read and clear the Radau pass's divergence counters
int hlm_debug_radau_counters(unsigned long long out[8])
{
    HLM_HIP_TRY(hipDeviceSynchronize());
    HLM_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(hlm::g_radau_ctr), 8 * sizeof(unsigned long long)));
    const unsigned long long zero[8] = {};
    HLM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(hlm::g_radau_ctr), zero, sizeof(zero)));
    return HLM_SUCCESS;
}
#endif

#ifdef HLM_PHASE_TIMING
// experiment builds only (not in include/hlm_hip.h): read (and optionally clear) the phase-cycle sums
int hlm_debug_phase_cycles(unsigned long long out[8], int reset)
{
    HLM_HIP_TRY(hipDeviceSynchronize());
    if (out) HLM_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(hlm::g_phase_cycles), 8 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero[8] = {};
        HLM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(hlm::g_phase_cycles), zero, sizeof(zero)));
    }
    return HLM_SUCCESS;
}
#endif

// ---- three-phase host API ----------------------------------------------------------------
int hlm_setup_gpu_buffers(const double* h_y0, int ns, const double* h_q, int nq, hlm_gpu_buffers* out)
{
    return hlm_setup_gpu_buffers_model(HLM_UID_204, h_y0, ns, h_q, nq, out);
}

int hlm_setup_gpu_buffers_model(int uid, const double* h_y0, int ns, const double* h_q, int nq, hlm_gpu_buffers* out)
{
    if (!out) return fail(HLM_ERR_INVALID_ARGUMENT, "out is NULL");
    std::memset(out, 0, sizeof(*out));
    const int neq = model_neq(uid);
    if (neq == 0) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
    out->neq = neq;
    if (ns < 0 || nq < 0) return fail(HLM_ERR_INVALID_ARGUMENT, "negative sizes");
    if ((ns > 0 && !h_y0) || (nq > 0 && !h_q)) return fail(HLM_ERR_INVALID_ARGUMENT, "NULL host input");
    out->num_systems = ns;
    out->num_queries = nq;
    const size_t b_state = sizeof(double) * (size_t)ns * (size_t)neq;
    const size_t b_q = sizeof(double) * (size_t)nq;
    // dense rows live tile-major in HBM ([ceil(ns/64)][nq][5][64], whole cache lines per store:
    // hlm_rk45_dense_kernel.h); hlm_retrieve_and_free delivers the host order
    out->dense_layout = hlm::DENSE_TILE_COMP;
    {
        std::lock_guard<std::mutex> lk(g_mu);  // the registered output selection (hlm_set_output_states)
        out->dense_ncomp = g_out_ncomp;
        std::memcpy(out->dense_comp, g_out_comp, sizeof(g_out_comp));
    }
    if (const int rcs = validate_selection(uid, out->dense_ncomp, out->dense_comp); rcs != HLM_SUCCESS) {
        std::memset(out, 0, sizeof(*out));
        return rcs;
    }
    const int width = out->dense_ncomp > 0 ? out->dense_ncomp : neq;
    const size_t b_dense = sizeof(double) * (size_t)((ns + hlm::WAVE - 1) / hlm::WAVE) * hlm::WAVE * (size_t)width * (size_t)nq;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMalloc(p, bytes);
    };
    alloc((void**)&out->d_y0_all, b_state);
    alloc((void**)&out->d_y_final_all, b_state);
    alloc((void**)&out->d_query_times, b_q);
    alloc((void**)&out->d_dense_all, b_dense);
    alloc((void**)&out->d_stiff, sizeof(int32_t) * (size_t)ns);
    alloc((void**)&out->d_n_attempt, sizeof(uint32_t) * (size_t)ns);
    alloc((void**)&out->d_n_accept, sizeof(uint32_t) * (size_t)ns);
    if (e == hipSuccess && b_state) e = hipMemcpy(out->d_y0_all, h_y0, b_state, hipMemcpyHostToDevice);
    if (e == hipSuccess && b_q) e = hipMemcpy(out->d_query_times, h_q, b_q, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(out->d_y0_all); (void)hipFree(out->d_y_final_all); (void)hipFree(out->d_query_times);
        (void)hipFree(out->d_dense_all); (void)hipFree(out->d_stiff); (void)hipFree(out->d_n_attempt);
        (void)hipFree(out->d_n_accept);
        std::memset(out, 0, sizeof(*out));
        return fail(HLM_ERR_HIP, "setup_gpu_buffers: %s", hipGetErrorString(e));
    }
    return HLM_SUCCESS;
}

int hlm_launch_rk45_kernel(int uid, const hlm_gpu_buffers* b, double t0, double tf, const void* d_sp)
{
    if (!b) return fail(HLM_ERR_INVALID_ARGUMENT, "buffers are NULL");
    if (b->neq != 0 && b->neq != model_neq(uid))
        return fail(HLM_ERR_INVALID_ARGUMENT, "buffers were set up for N_EQ = %d, model %d has %d", b->neq, uid, model_neq(uid));
    ForcingSlot* slot = forc_slot();
    if (!slot) return HLM_ERR_HIP;
    hlm_device_job j;
    std::memset(&j, 0, sizeof(j));
    const float* host_tables[HLM_MAX_FORCINGS];
    double chunk_minutes = 0.0;
    int radau = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (slot->in_flight)
            return fail(HLM_ERR_STATE, "another hlm_launch_rk45_kernel / hlm_run_rk45 is running on this device "
                                       "with the same forcing registration (one host-API integration per device)");
        const int rc0 = fill_job_from_registry(&j, uid, slot->f);
        if (rc0 != HLM_SUCCESS) return rc0;
        chunk_minutes = slot->f.chunk_minutes;
        std::memcpy(host_tables, slot->f.h_tables, sizeof(host_tables));
        radau = g_radau;
        slot->in_flight = 1;
    }
    struct Release {  // every exit path frees the slot again
        ForcingSlot* s;
        ~Release() { std::lock_guard<std::mutex> lk(g_mu); s->in_flight = 0; }
    } release{slot};
    j.num_systems = b->num_systems;
    j.num_queries = b->num_queries;
    j.t0 = t0;
    j.tf = tf;
    j.d_y0 = b->d_y0_all;
    j.d_query = b->d_query_times;
    j.d_sp = d_sp;
    j.d_y_final = b->d_y_final_all;
    j.d_dense = b->d_dense_all;
    j.dense_tiled = b->dense_layout;
    j.dense_ncomp = b->dense_ncomp;
    std::memcpy(j.dense_comp, b->dense_comp, sizeof(j.dense_comp));
    j.d_stiff = b->d_stiff;
    j.d_n_attempt = b->d_n_attempt;
    j.d_n_accept = b->d_n_accept;
    const bool streamed = chunk_minutes != 0.0 && model_nf(uid) > 0;
    if (streamed && radau)
        return fail(HLM_ERR_INVALID_ARGUMENT, "the Radau fallback needs the forcing series resident in HBM "
                                              "(hlm_set_forcings), not streamed in time chunks");
    int rc;
    double* flag_state = nullptr;  // Radau "continue" mode: where the RK45 launch leaves the state of a flagged system
    if (radau == 2 && j.num_systems > 0) {
        HLM_HIP_TRY(hipMalloc((void**)&flag_state, sizeof(double) * (size_t)j.num_systems * (size_t)(model_neq(uid) + 1)));
        j.d_flag_state = flag_state;
    }
    if (streamed)
        rc = hlm_rk45_integrate_forcing_streamed(&j, host_tables, chunk_minutes, nullptr, nullptr);
    else
        rc = hlm_rk45_integrate_device(&j, nullptr);
    if (rc == HLM_SUCCESS && radau) rc = hlm_radau_fallback_device(&j, nullptr);  // rk45_api.hpp:189-247
    const hipError_t es = hipDeviceSynchronize();  // rk45_api.hpp:150
    if (flag_state) (void)hipFree(flag_state);
    if (rc != HLM_SUCCESS) return rc;
    HLM_HIP_TRY(es);
    return HLM_SUCCESS;
}

// dense rows of a tile-major device buffer -> h_dense in the host order (s*nq + q)*5 + c
// (rk45_api.hpp:255-267).  Pinned, device-mapped host memory is written by the transposing kernel
// directly; any other destination goes through a device staging buffer of whole tiles.
static hipError_t dense_to_host_order(const double* d_dense, int layout, int neq, size_t ns, size_t nq, double* h_dense)
{
    if (!h_dense || !d_dense || ns == 0 || nq == 0) return hipSuccess;
    const size_t row = sizeof(double) * (size_t)neq;
    if (layout == hlm::DENSE_LINEAR) return hipMemcpy(h_dense, d_dense, ns * nq * row, hipMemcpyDeviceToHost);
    const int nblk = (int)((ns + hlm::WAVE - 1) / hlm::WAVE);
    auto launch_tiles = [&](int blk0, int nb, double* dst_for_sys0) {  // dst_for_sys0: where system 0's rows would start
        const dim3 grid((unsigned)nb, (unsigned)((nq + hlm::TILE_TQ - 1) / hlm::TILE_TQ));
        const double* src = d_dense + (size_t)blk0 * nq * hlm::WAVE * (size_t)neq;
        double* dst = dst_for_sys0 + (size_t)blk0 * hlm::WAVE * nq * (size_t)neq;
        const int ns_left = (int)(ns - (size_t)blk0 * hlm::WAVE);
        launch_tile_to_host(layout, neq, grid, nullptr, src, dst, ns_left, (int)nq, 0, (int)nq);
        return hipGetLastError();
    };
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, h_dense) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
        hipError_t e = launch_tiles(0, nblk, static_cast<double*>(at.devicePointer));
        return e != hipSuccess ? e : hipDeviceSynchronize();
    }
    (void)hipGetLastError();  // pageable memory: not an error
    const size_t tile_bytes = (size_t)hlm::WAVE * nq * row;
    size_t tiles_per_pass = ((size_t)256 << 20) / tile_bytes;
    if (tiles_per_pass == 0) tiles_per_pass = 1;
    if (tiles_per_pass > (size_t)nblk) tiles_per_pass = (size_t)nblk;
    double* stage = nullptr;
    hipError_t e = hipMalloc((void**)&stage, tiles_per_pass * tile_bytes);
    for (int b0 = 0; e == hipSuccess && b0 < nblk; b0 += (int)tiles_per_pass) {
        const int nb = (int)((size_t)(nblk - b0) < tiles_per_pass ? (size_t)(nblk - b0) : tiles_per_pass);
        const size_t sys0 = (size_t)b0 * hlm::WAVE, nsys = (size_t)nb * hlm::WAVE < ns - sys0 ? (size_t)nb * hlm::WAVE : ns - sys0;
        // the kernel addresses dst by the system index within the launch: shift so that tile b0 lands at stage[0]
        e = launch_tiles(b0, nb, stage - (size_t)b0 * hlm::WAVE * nq * (size_t)neq);
        if (e == hipSuccess) e = hipMemcpy(h_dense + sys0 * nq * (size_t)neq, stage, nsys * nq * row, hipMemcpyDeviceToHost);
    }
    if (stage) (void)hipFree(stage);
    return e;
}

int hlm_retrieve_and_free(hlm_gpu_buffers* b, double* h_final, double* h_dense, int32_t* h_stiff,
                          uint32_t* h_n_attempt, uint32_t* h_n_accept)
{
    if (!b) return fail(HLM_ERR_INVALID_ARGUMENT, "buffers are NULL");
    const size_t ns = (size_t)b->num_systems, nq = (size_t)b->num_queries;
    hipError_t e = hipSuccess;
    auto d2h = [&](void* dst, const void* src, size_t bytes) {
        if (e == hipSuccess && dst && src && bytes) e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    };
    const int neq = b->neq ? b->neq : HLM_NEQ;
    d2h(h_final, b->d_y_final_all, sizeof(double) * ns * (size_t)neq);
    if (e == hipSuccess) e = dense_to_host_order(b->d_dense_all, b->dense_layout, b->dense_ncomp > 0 ? b->dense_ncomp : neq, ns, nq, h_dense);
    d2h(h_stiff, b->d_stiff, sizeof(int32_t) * ns);
    d2h(h_n_attempt, b->d_n_attempt, sizeof(uint32_t) * ns);
    d2h(h_n_accept, b->d_n_accept, sizeof(uint32_t) * ns);
    (void)hipFree(b->d_y0_all); (void)hipFree(b->d_y_final_all); (void)hipFree(b->d_query_times);
    (void)hipFree(b->d_dense_all); (void)hipFree(b->d_stiff); (void)hipFree(b->d_n_attempt);
    (void)hipFree(b->d_n_accept);
    std::memset(b, 0, sizeof(*b));
    if (e != hipSuccess) return fail(HLM_ERR_HIP, "retrieve_and_free: %s", hipGetErrorString(e));
    return HLM_SUCCESS;
}

int hlm_run_rk45_streamed(int uid, const double* h_y0, int ns, double t0, double tf, const double* h_q, int nq, const void* d_sp,
                          int window_queries, hlm_window_sink sink, void* user, double* h_final, int32_t* h_stiff,
                          uint32_t* h_n_attempt, uint32_t* h_n_accept)
{
    if (nq <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "run_rk45_streamed needs query times (use hlm_run_rk45 for a run without dense output)");
    hlm_gpu_buffers b;
    int rc = hlm_setup_gpu_buffers_model(uid, h_y0, ns, h_q, 0, &b);  // no dense buffer: the windows are the callee's
    if (rc != HLM_SUCCESS) return rc;
    double* d_query = nullptr;
    auto cleanup = [&](int code) {
        char keep[sizeof(g_err)];
        std::memcpy(keep, g_err, sizeof(keep));
        if (d_query) (void)hipFree(d_query);
        (void)hlm_retrieve_and_free(&b, nullptr, nullptr, nullptr, nullptr, nullptr);
        std::memcpy(g_err, keep, sizeof(keep));
        return code;
    };
    if (hipMalloc((void**)&d_query, sizeof(double) * (size_t)nq) != hipSuccess ||
        hipMemcpy(d_query, h_q, sizeof(double) * (size_t)nq, hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail(HLM_ERR_HIP, "run_rk45_streamed: query upload failed"));
    ForcingSlot* slot = forc_slot();
    if (!slot) return cleanup(HLM_ERR_HIP);
    hlm_device_job j;
    std::memset(&j, 0, sizeof(j));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (slot->in_flight)
            return cleanup(fail(HLM_ERR_STATE, "another host-API integration is running on this device with the same forcing registration"));
        if (g_radau) return cleanup(fail(HLM_ERR_INVALID_ARGUMENT, "run_rk45_streamed cannot be combined with the Radau fallback"));
        if (slot->f.chunk_minutes != 0.0 && model_nf(uid) > 0)
            return cleanup(fail(HLM_ERR_INVALID_ARGUMENT, "run_rk45_streamed needs the forcings resident in HBM (hlm_set_forcings)"));
        const int rc0 = fill_job_from_registry(&j, uid, slot->f);
        if (rc0 != HLM_SUCCESS) return cleanup(rc0);
        slot->in_flight = 1;
    }
    struct Release {
        ForcingSlot* s;
        ~Release() { std::lock_guard<std::mutex> lk(g_mu); s->in_flight = 0; }
    } release{slot};
    j.num_systems = ns;
    j.num_queries = nq;
    j.time_pieces = 0;  // (a launch shape of runs without dense output)
    j.t0 = t0;
    j.tf = tf;
    j.d_y0 = b.d_y0_all;
    j.d_query = d_query;
    j.d_sp = d_sp;
    j.d_y_final = b.d_y_final_all;
    j.d_stiff = b.d_stiff;
    j.d_n_attempt = b.d_n_attempt;
    j.d_n_accept = b.d_n_accept;
    j.dense_ncomp = b.dense_ncomp;
    std::memcpy(j.dense_comp, b.dense_comp, sizeof(j.dense_comp));
    rc = ns > 0 ? hlm_rk45_integrate_streamed_sink(&j, window_queries, sink, user, nullptr, nullptr) : HLM_SUCCESS;
    if (rc != HLM_SUCCESS) return cleanup(rc);
    (void)hipFree(d_query);
    return hlm_retrieve_and_free(&b, h_final, nullptr, h_stiff, h_n_attempt, h_n_accept);
}

int hlm_run_rk45(int uid, const double* h_y0, int ns, double t0, double tf, const double* h_q, int nq,
                 const void* d_sp, double* h_final, double* h_dense, int32_t* h_stiff,
                 uint32_t* h_n_attempt, uint32_t* h_n_accept)
{
    hlm_gpu_buffers b;
    int rc = hlm_setup_gpu_buffers_model(uid, h_y0, ns, h_q, nq, &b);
    if (rc != HLM_SUCCESS) return rc;
    rc = hlm_launch_rk45_kernel(uid, &b, t0, tf, d_sp);
    if (rc != HLM_SUCCESS) {
        char keep[sizeof(g_err)];
        std::memcpy(keep, g_err, sizeof(keep));
        (void)hlm_retrieve_and_free(&b, nullptr, nullptr, nullptr, nullptr, nullptr);
        std::memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    return hlm_retrieve_and_free(&b, h_final, h_dense, h_stiff, h_n_attempt, h_n_accept);
}


int hlm_shard_bounds(int ns, int G, int s, int* start, int* count)
{
    if (ns < 0 || G < 1 || s < 0 || s >= G) return fail(HLM_ERR_INVALID_ARGUMENT, "shard %d of %d over %d hillslopes", s, G, ns);
    const int base = ns / G, rem = ns % G;  // main.cpp:275-307
    if (start) *start = s * base + (s < rem ? s : rem);
    if (count) *count = base + (s < rem ? 1 : 0);
    return HLM_SUCCESS;
}

int hlm_gather_plan(int num_systems, int num_shards, int num_devices, int neq, hlm_gather_piece* pieces, int* num_pieces)
{
    if (num_systems < 0 || num_shards < 1 || num_devices < 1 || neq < 1 || neq > HLM_MAX_NEQ || !num_pieces)
        return fail(HLM_ERR_INVALID_ARGUMENT, "gather_plan: %d hillslopes, %d shards, %d devices, N_EQ %d", num_systems, num_shards, num_devices, neq);
    const uint64_t width[4] = {sizeof(double) * (uint64_t)neq, sizeof(int32_t), sizeof(uint32_t), sizeof(uint32_t)};
    const int capacity = pieces ? *num_pieces : 0;   // (in: entries `pieces` can hold; out: entries of the plan)
    int n = 0;
    for (int s = 0; s < num_shards; ++s) {
        int start = 0, count = 0;
        (void)hlm_shard_bounds(num_systems, num_shards, s, &start, &count);
        if (count == 0) continue;   // (more shards than hillslopes: the empty ones send nothing)
        for (int a = 0; a < 4; ++a, ++n) {
            if (!pieces || n >= capacity) continue;
            pieces[n].shard = s;
            pieces[n].from_rank = s % num_devices;
            pieces[n].array = a;
            pieces[n].dst_offset_bytes = (uint64_t)start * width[a];
            pieces[n].bytes = (uint64_t)count * width[a];
        }
    }
    *num_pieces = n;
    if (pieces && n > capacity) return fail(HLM_ERR_INVALID_ARGUMENT, "gather_plan: %d pieces, the array holds %d", n, capacity);
    return HLM_SUCCESS;
}

int hlm_set_sharded_upload(int mode)
{
    if (mode != HLM_UPLOAD_SINGLE_THREAD && mode != HLM_UPLOAD_THREADS && mode != HLM_UPLOAD_STAGED_THREADS) return fail(HLM_ERR_INVALID_ARGUMENT, "sharded upload mode %d", mode);
    std::lock_guard<std::mutex> lk(g_mu);
    g_sharded_upload = mode;
    return HLM_SUCCESS;
}

}  // extern "C"

namespace {

// ---- one process, the GPUs of a node: what hlm_run_rk45_sharded and hlm_run_rk45_sharded_streamed share -------------
// Host -> device.  A copy from the caller's pageable arrays occupies the calling thread until the bytes have left the
// host buffer, so one host thread feeding D devices uploads to them one after the other (round 4).  Every device
// therefore has its own host thread for the upload phase (HLM_UPLOAD_THREADS, the default): D uploads run side by side.
// The copies themselves are the runtime's: measured on an MI355X of this pool, hipMemcpyAsync / hipMemcpy2DAsync from
// pageable memory move the 3.18 GB of a 1 M-hillslope per-stream run at 55 GB/s -- the link's rate (the runtime pins the
// user pages in place) -- so there is nothing left for a staging scheme to win per device.  PinnedStager is that scheme
// (HLM_UPLOAD_STAGED_THREADS: the thread packs 16 MB chunks into two pinned buffers while the DMA engine moves the
// previous one; 27-34 GB/s, the rate of one memcpy thread): kept selectable for hosts whose runtime stages pageable
// copies through a bounce buffer, not the default.
struct PinnedStager {
    static constexpr size_t kChunk = (size_t)16 << 20;
    hipStream_t stream = nullptr;
    char* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int b = 0;
    hipError_t init(hipStream_t s)
    {
        stream = s;
        for (int k = 0; k < 2; ++k) {
            if (hipError_t e = hipHostMalloc((void**)&pin[k], kChunk, hipHostMallocDefault); e != hipSuccess) return e;
            if (hipError_t e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming); e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    hipError_t flush_chunk(void* dst, size_t bytes)
    {
        if (hipError_t e = hipMemcpyAsync(dst, pin[b], bytes, hipMemcpyHostToDevice, stream); e != hipSuccess) return e;
        if (hipError_t e = hipEventRecord(ev[b], stream); e != hipSuccess) return e;
        busy[b] = true;
        b ^= 1;
        return hipSuccess;
    }
    hipError_t acquire()
    {
        if (busy[b]) {
            if (hipError_t e = hipEventSynchronize(ev[b]); e != hipSuccess) return e;
            busy[b] = false;
        }
        return hipSuccess;
    }
    // dst[0 .. rows * width) (device, contiguous) = rows of `width` bytes taken every `pitch` bytes from src (host)
    hipError_t put2d(void* dst, const void* src, size_t pitch, size_t width, size_t rows)
    {
        char* d = static_cast<char*>(dst);
        const char* sp = static_cast<const char*>(src);
        if (width == pitch) { width *= rows; pitch = width; rows = rows ? 1 : 0; }   // contiguous: one long row
        size_t row = 0, col = 0;   // next byte to take
        while (row < rows) {
            if (hipError_t e = acquire(); e != hipSuccess) return e;
            size_t fill = 0;
            while (fill < kChunk && row < rows) {
                const size_t take = (width - col) < (kChunk - fill) ? (width - col) : (kChunk - fill);
                std::memcpy(pin[b] + fill, sp + row * pitch + col, take);
                fill += take;
                col += take;
                if (col == width) { col = 0; ++row; }
            }
            if (hipError_t e = flush_chunk(d, fill); e != hipSuccess) return e;
            d += fill;
        }
        return hipSuccess;
    }
    hipError_t put(void* dst, const void* src, size_t bytes) { return put2d(dst, src, bytes, bytes, bytes ? 1 : 0); }
    void destroy()
    {
        if (stream) (void)hipStreamSynchronize(stream);   // the copies out of the pinned buffers are done
        for (int k = 0; k < 2; ++k) {
            if (pin[k]) (void)hipHostFree(pin[k]);
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            pin[k] = nullptr;
            ev[k] = nullptr;
        }
    }
};

struct ShardedSetup {
    struct Shard {
        int dev_rank = 0, start = 0, count = 0;
        double* y0 = nullptr;
        void* sp = nullptr;
        float* forc = nullptr;      // per-stream layout: this shard's columns
        int32_t* col = nullptr;     // per-cell layout: this shard's slice of the index
        double* y_final = nullptr;
        int32_t* stiff = nullptr;
        uint32_t *n_attempt = nullptr, *n_accept = nullptr;
    };
    struct Device {
        hipStream_t stream = nullptr;
        float* tables = nullptr;    // per-cell layout: the tables, once per device
        hipEvent_t begin = nullptr, done = nullptr;   // around this device's integrations
        double upload_ms = 0.0;     // this device's uploads, enqueue to completion (host clock of its thread)
        std::string error;          // set by the device's worker thread
        int rc = HLM_SUCCESS;
    };
    const hlm_sharded_run* run = nullptr;
    int uid = 0, ns = 0, neq = 0, num_shards = 0, num_devices = 0, prev_dev = 0;
    bool reads_forcings = false;
    size_t table_rows = 0;
    std::vector<int> devs;
    std::vector<Shard> shards;
    std::vector<Device> dv;

    int validate(const hlm_sharded_run* r, int nshards, const int* devices, int ndev, const double* h_y_final)
    {
        run = r;
        if (!run) return fail(HLM_ERR_INVALID_ARGUMENT, "run is NULL");
        uid = run->uid; ns = run->num_systems; neq = model_neq(uid);
        num_shards = nshards; num_devices = ndev;
        if (neq == 0) return fail(HLM_ERR_UNKNOWN_MODEL, "unknown model UID %d", uid);
        if (ns < 0) return fail(HLM_ERR_INVALID_ARGUMENT, "negative num_systems");
        if (num_shards < 1 || num_devices < 1) return fail(HLM_ERR_INVALID_ARGUMENT, "num_shards and num_devices must be >= 1");
        if (num_devices > HLM_MAX_SHARD_DEVICES) return fail(HLM_ERR_INVALID_ARGUMENT, "at most %d devices", HLM_MAX_SHARD_DEVICES);
        if (ns > 0 && (!run->h_y0 || !h_y_final)) return fail(HLM_ERR_INVALID_ARGUMENT, "h_y0 / h_y_final must not be NULL");
        if (ns > 0 && model_needs_sp(uid) && !run->h_sp) return fail(HLM_ERR_INVALID_ARGUMENT, "Model 204 needs h_sp (SpatialParams)");
        if (run->nforc < 0 || run->nforc > HLM_MAX_FORCINGS) return fail(HLM_ERR_INVALID_ARGUMENT, "nforc %d outside [0,%d]", run->nforc, HLM_MAX_FORCINGS);
        reads_forcings = model_nf(uid) > 0 && run->nforc > 0;
        table_rows = 0;
        if (reads_forcings) {
            if (!run->h_forc_tables || run->forc_ld <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "nforc > 0 needs h_forc_tables and forc_ld > 0");
            if (!run->h_forc_col && run->forc_ld != ns)
                return fail(HLM_ERR_INVALID_ARGUMENT, "forc_ld (%lld) != num_systems (%d) without a column index", (long long)run->forc_ld, ns);
            for (int k = 0; k < run->nforc; ++k) {
                if (!run->h_forc_tables[k] || run->forc_nT[k] == 0 || !(run->forc_dt_hours[k] > 0.0))
                    return fail(HLM_ERR_INVALID_ARGUMENT, "forcing %d: table / nT / dt_hours", k);
                table_rows += (size_t)run->forc_nT[k];
            }
        }
        int visible = 0;
        HLM_HIP_TRY(hipGetDeviceCount(&visible));
        if (visible <= 0) return fail(HLM_ERR_NO_DEVICE, "no HIP device");
        devs.resize((size_t)num_devices);
        for (int i = 0; i < num_devices; ++i) {
            devs[(size_t)i] = devices ? devices[i] : i;
            if (devs[(size_t)i] < 0 || devs[(size_t)i] >= visible) return fail(HLM_ERR_INVALID_ARGUMENT, "device %d of %d visible", devs[(size_t)i], visible);
            for (int k = 0; k < i; ++k)
                if (devs[(size_t)k] == devs[(size_t)i]) return fail(HLM_ERR_INVALID_ARGUMENT, "device %d listed twice", devs[(size_t)i]);
        }
        return HLM_SUCCESS;
    }

    // everything of device d: stream, events, the per-cell tables, and the inputs of its shards (in shard order)
    void upload_device(int d, bool staged)
    {
        using clk = std::chrono::steady_clock;
        Device& D = dv[(size_t)d];
        auto bad = [&](const char* what, hipError_t e) { D.rc = HLM_ERR_HIP; D.error = std::string(what) + " failed: " + hipGetErrorString(e); };
#define HLM_UP_TRY(expr)                                         \
    do {                                                         \
        const hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) { bad(#expr, e_); stager.destroy(); return; } \
    } while (0)
        PinnedStager stager;
        HLM_UP_TRY(hipSetDevice(devs[(size_t)d]));
        HLM_UP_TRY(hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking));
        HLM_UP_TRY(hipEventCreate(&D.begin));
        HLM_UP_TRY(hipEventCreate(&D.done));
        if (staged) HLM_UP_TRY(stager.init(D.stream));
        const clk::time_point t0 = clk::now();
        auto put2d = [&](void* dst, const void* src, size_t pitch, size_t width, size_t rows) -> hipError_t {
            if (staged) return stager.put2d(dst, src, pitch, width, rows);
            if (pitch == width) return hipMemcpyAsync(dst, src, width * rows, hipMemcpyHostToDevice, D.stream);
            return hipMemcpy2DAsync(dst, width, src, pitch, width, rows, hipMemcpyHostToDevice, D.stream);
        };
        if (reads_forcings && run->h_forc_col && d < num_shards) {
            HLM_UP_TRY(hipMalloc((void**)&D.tables, sizeof(float) * table_rows * (size_t)run->forc_ld));
            size_t row = 0;
            for (int k = 0; k < run->nforc; ++k) {
                const size_t bytes = sizeof(float) * (size_t)run->forc_nT[k] * (size_t)run->forc_ld;
                HLM_UP_TRY(put2d(D.tables + row * (size_t)run->forc_ld, run->h_forc_tables[k], bytes, bytes, 1));
                row += (size_t)run->forc_nT[k];
            }
        }
        const size_t sp_bytes = sizeof(hlm_spatial_params);
        for (int s = d; s < num_shards; s += num_devices) {
            Shard& sh = shards[(size_t)s];
            if (sh.count == 0) continue;
            const size_t n = (size_t)sh.count;
            HLM_UP_TRY(hipMalloc((void**)&sh.y0, sizeof(double) * n * (size_t)neq));
            HLM_UP_TRY(hipMalloc((void**)&sh.y_final, sizeof(double) * n * (size_t)neq));
            HLM_UP_TRY(hipMalloc((void**)&sh.stiff, sizeof(int32_t) * n));
            HLM_UP_TRY(hipMalloc((void**)&sh.n_attempt, sizeof(uint32_t) * n));
            HLM_UP_TRY(hipMalloc((void**)&sh.n_accept, sizeof(uint32_t) * n));
            const size_t yb = sizeof(double) * n * (size_t)neq;
            HLM_UP_TRY(put2d(sh.y0, run->h_y0 + (size_t)sh.start * (size_t)neq, yb, yb, 1));
            if (run->h_sp) {
                HLM_UP_TRY(hipMalloc(&sh.sp, sp_bytes * n));
                HLM_UP_TRY(put2d(sh.sp, static_cast<const char*>(run->h_sp) + (size_t)sh.start * sp_bytes, sp_bytes * n, sp_bytes * n, 1));
            }
            if (reads_forcings && run->h_forc_col) {
                HLM_UP_TRY(hipMalloc((void**)&sh.col, sizeof(int32_t) * n));
                HLM_UP_TRY(put2d(sh.col, run->h_forc_col + sh.start, sizeof(int32_t) * n, sizeof(int32_t) * n, 1));
            } else if (reads_forcings) {
                // the shard's columns of the per-stream tables: [rows][count] out of [rows][ns]
                HLM_UP_TRY(hipMalloc((void**)&sh.forc, sizeof(float) * table_rows * n));
                size_t row = 0;
                for (int k = 0; k < run->nforc; ++k) {
                    HLM_UP_TRY(put2d(sh.forc + row * n, run->h_forc_tables[k] + sh.start, sizeof(float) * (size_t)run->forc_ld, sizeof(float) * n, (size_t)run->forc_nT[k]));
                    row += (size_t)run->forc_nT[k];
                }
            }
        }
        HLM_UP_TRY(hipStreamSynchronize(D.stream));
        D.upload_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        stager.destroy();
#undef HLM_UP_TRY
    }

    // split by the reference's rule and upload: one host thread per device (staged) or this thread for all of them
    int upload(int mode, double* upload_ms)
    {
        using clk = std::chrono::steady_clock;
        HLM_HIP_TRY(hipGetDevice(&prev_dev));
        shards.assign((size_t)num_shards, Shard{});
        dv.assign((size_t)num_devices, Device{});
        for (int s = 0; s < num_shards; ++s) {
            shards[(size_t)s].dev_rank = s % num_devices;
            (void)hlm_shard_bounds(ns, num_shards, s, &shards[(size_t)s].start, &shards[(size_t)s].count);
        }
        const clk::time_point tp = clk::now();
        if (mode != HLM_UPLOAD_SINGLE_THREAD) {
            const bool staged = mode == HLM_UPLOAD_STAGED_THREADS;
            std::vector<std::thread> th;
            for (int d = 0; d < num_devices; ++d) th.emplace_back([this, d, staged] { upload_device(d, staged); });
            for (std::thread& t : th) t.join();
        } else {
            for (int d = 0; d < num_devices; ++d) upload_device(d, false);
        }
        if (upload_ms) *upload_ms = std::chrono::duration<double, std::milli>(clk::now() - tp).count();
        for (int d = 0; d < num_devices; ++d)
            if (dv[(size_t)d].rc != HLM_SUCCESS) return fail(dv[(size_t)d].rc, "device %d: %s", devs[(size_t)d], dv[(size_t)d].error.c_str());
        return HLM_SUCCESS;
    }

    // the job of shard s on its device (registered solver parameters, math mode, launch shape)
    int job_for(int s, hlm_device_job* j) const
    {
        const Shard& sh = shards[(size_t)s];
        ForcingState F;
        if (reads_forcings) {
            F.d_forc = run->h_forc_col ? dv[(size_t)sh.dev_rank].tables : sh.forc;
            F.d_col = sh.col;
            F.ld = run->h_forc_col ? run->forc_ld : (int64_t)sh.count;
            F.nforc = run->nforc;
            std::memcpy(F.dt_h, run->forc_dt_hours, sizeof(F.dt_h));
            std::memcpy(F.nT, run->forc_nT, sizeof(F.nT));
        }
        std::memset(j, 0, sizeof(*j));
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if (const int rc0 = fill_job_from_registry(j, uid, F); rc0 != HLM_SUCCESS) return rc0;
        }
        j->num_systems = sh.count;
        j->t0 = run->t0;
        j->tf = run->tf;
        j->d_y0 = sh.y0;
        j->d_sp = sh.sp;
        j->d_y_final = sh.y_final;
        j->d_stiff = sh.stiff;
        j->d_n_attempt = sh.n_attempt;
        j->d_n_accept = sh.n_accept;
        return HLM_SUCCESS;
    }

    void release()
    {
        for (Shard& s : shards) {
            (void)hipSetDevice(devs[(size_t)s.dev_rank]);
            for (void* p : {(void*)s.y0, s.sp, (void*)s.forc, (void*)s.col, (void*)s.y_final, (void*)s.stiff, (void*)s.n_attempt, (void*)s.n_accept})
                if (p) (void)hipFree(p);
        }
        for (size_t d = 0; d < dv.size(); ++d) {
            (void)hipSetDevice(devs[d]);
            if (dv[d].stream) (void)hipStreamSynchronize(dv[d].stream);   // nothing of this device is in flight when its memory goes
            if (dv[d].tables) (void)hipFree(dv[d].tables);
            if (dv[d].done) (void)hipEventDestroy(dv[d].done);
            if (dv[d].begin) (void)hipEventDestroy(dv[d].begin);
            if (dv[d].stream) (void)hipStreamDestroy(dv[d].stream);
        }
        shards.clear();
        dv.clear();
        (void)hipSetDevice(prev_dev);
    }
};

}  // namespace

extern "C" {

int hlm_run_rk45_sharded(const hlm_sharded_run* run, int num_shards, const int* devices, int num_devices, int gather_mode,
                         double* h_y_final, int32_t* h_stiff, uint32_t* h_n_attempt, uint32_t* h_n_accept,
                         hlm_sharded_stats* stats)
{
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (gather_mode != HLM_GATHER_HOST && gather_mode != HLM_GATHER_RCCL) return fail(HLM_ERR_INVALID_ARGUMENT, "gather_mode %d", gather_mode);
    ShardedSetup S;
    if (const int rc = S.validate(run, num_shards, devices, num_devices, h_y_final); rc != HLM_SUCCESS) return rc;
    const int ns = S.ns, neq = S.neq;
    if (stats) {
        stats->num_shards = num_shards;
        stats->num_devices = num_devices;
        for (int d = 0; d < num_devices; ++d) stats->device_ids[d] = S.devs[(size_t)d];
    }
    if (ns == 0) return HLM_SUCCESS;
    int upload_mode, radau;
    { std::lock_guard<std::mutex> lk(g_mu); upload_mode = g_sharded_upload; radau = g_radau; }

    double* g_y = nullptr;          // the gathered arrays on devices[0] (HLM_GATHER_RCCL)
    int32_t* g_stiff = nullptr;
    uint32_t *g_att = nullptr, *g_acc = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hlm::RcclGather* comm = nullptr;   // the process's communicator over these devices (cached: hlm_gather_rccl.h)
    auto cleanup = [&](int rc) {
        char keep[sizeof(g_err)];
        std::memcpy(keep, g_err, sizeof(keep));
        S.release();   // (synchronises every device's stream before its buffers are freed)
        (void)hipSetDevice(S.devs[0]);
        for (void* p : {(void*)g_y, (void*)g_stiff, (void*)g_att, (void*)g_acc})
            if (p) (void)hipFree(p);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        (void)hipSetDevice(S.prev_dev);
        std::memcpy(g_err, keep, sizeof(keep));
        return rc;
    };
#define HLM_SH_TRY(expr)                                                                                  \
    do {                                                                                                  \
        const hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return cleanup(fail(HLM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
    } while (0)

    // ---- inputs: split by the reference's rule, one stream (and one uploading host thread) per device ---------
    double up_ms = 0.0;
    if (const int rcu = S.upload(upload_mode, &up_ms); rcu != HLM_SUCCESS) return cleanup(rcu);
    if (stats) {
        stats->upload_ms = up_ms;
        stats->upload_mode = upload_mode;
        for (int d = 0; d < num_devices; ++d) stats->upload_ms_device[d] = S.dv[(size_t)d].upload_ms;
    }
    // the communicator comes up BEFORE the timed integrations (ncclCommInitAll stalls the devices while it runs: the
    // first one of a process takes ~1 s, later ones are cached; round-4 advisor finding: it sat inside integrate_ms)
    if (gather_mode == HLM_GATHER_RCCL) {
        const clk::time_point tc = clk::now();
        std::string err;
        comm = hlm::rccl_gather_for(S.devs, &err);
        if (!comm) return cleanup(fail(HLM_ERR_HIP, "%s", err.c_str()));
        if (stats) { stats->comm_init_ms = ms_since(tc); stats->rccl_version = comm->version(); stats->rccl_nranks = comm->nranks(); }
    }

    // ---- integrate: every shard on its device's stream (asynchronous: the devices run side by side) ---------
    clk::time_point tp = clk::now();
    for (int d = 0; d < num_devices; ++d) {
        HLM_SH_TRY(hipSetDevice(S.devs[(size_t)d]));
        HLM_SH_TRY(hipEventRecord(S.dv[(size_t)d].begin, S.dv[(size_t)d].stream));
    }
    for (int s = 0; s < num_shards; ++s) {
        if (S.shards[(size_t)s].count == 0) continue;
        HLM_SH_TRY(hipSetDevice(S.devs[(size_t)S.shards[(size_t)s].dev_rank]));
        hlm_device_job j;
        if (const int rc0 = S.job_for(s, &j); rc0 != HLM_SUCCESS) return cleanup(rc0);
        hipStream_t st = S.dv[(size_t)S.shards[(size_t)s].dev_rank].stream;
        // hlm_set_radau_fallback: every shard's flagged hillslopes go through the implicit pass on the shard's own device, as each of
        // the reference's ranks relaunches for its own (rk45_api.hpp:189-247); mode 2: from the states its RK45 launch leaves
        double* flag_state = nullptr;
        if (radau == 2) {
            HLM_SH_TRY(hipMallocAsync(reinterpret_cast<void**>(&flag_state), sizeof(double) * (size_t)j.num_systems * (size_t)(neq + 1), st));
            j.d_flag_state = flag_state;
        }
        int rc1 = hlm_rk45_integrate_device(&j, st);
        if (rc1 == HLM_SUCCESS && radau) rc1 = hlm_radau_fallback_device(&j, st);
        if (flag_state) (void)hipFreeAsync(flag_state, st);   // (stream-ordered: after the pass)
        if (rc1 != HLM_SUCCESS) return cleanup(rc1);
    }
    for (int d = 0; d < num_devices; ++d) {
        HLM_SH_TRY(hipSetDevice(S.devs[(size_t)d]));
        HLM_SH_TRY(hipEventRecord(S.dv[(size_t)d].done, S.dv[(size_t)d].stream));
    }
    double slowest = 0.0;
    for (int d = 0; d < num_devices; ++d) {
        HLM_SH_TRY(hipSetDevice(S.devs[(size_t)d]));
        HLM_SH_TRY(hipStreamSynchronize(S.dv[(size_t)d].stream));
        float ms = 0.0f;
        HLM_SH_TRY(hipEventElapsedTime(&ms, S.dv[(size_t)d].begin, S.dv[(size_t)d].done));
        if (stats) stats->integrate_ms_device[d] = ms;
        if (ms > slowest) slowest = ms;
    }
    if (stats) { stats->integrate_ms = slowest; stats->wall_to_integrated_ms = ms_since(tp); }

    // ---- gather -------------------------------------------------------------------------------------------
    tp = clk::now();
    if (gather_mode == HLM_GATHER_RCCL) {
        HLM_SH_TRY(hipSetDevice(S.devs[0]));
        HLM_SH_TRY(hipMalloc((void**)&g_y, sizeof(double) * (size_t)ns * (size_t)neq));
        HLM_SH_TRY(hipMalloc((void**)&g_stiff, sizeof(int32_t) * (size_t)ns));
        HLM_SH_TRY(hipMalloc((void**)&g_att, sizeof(uint32_t) * (size_t)ns));
        HLM_SH_TRY(hipMalloc((void**)&g_acc, sizeof(uint32_t) * (size_t)ns));
        HLM_SH_TRY(hipEventCreate(&ev0));
        HLM_SH_TRY(hipEventCreate(&ev1));
        // the piece list: hlm_gather_plan (the same function a test reads without any device)
        int np = 0;
        (void)hlm_gather_plan(ns, num_shards, num_devices, neq, nullptr, &np);
        std::vector<hlm_gather_piece> plan((size_t)np);
        (void)hlm_gather_plan(ns, num_shards, num_devices, neq, plan.data(), &np);
        char* const dst_base[4] = {reinterpret_cast<char*>(g_y), reinterpret_cast<char*>(g_stiff), reinterpret_cast<char*>(g_att), reinterpret_cast<char*>(g_acc)};
        std::vector<hlm::GatherPiece> pieces;
        uint64_t bytes = 0;
        for (const hlm_gather_piece& q : plan) {
            const ShardedSetup::Shard& sh = S.shards[(size_t)q.shard];
            const void* const src[4] = {sh.y_final, sh.stiff, sh.n_attempt, sh.n_accept};
            pieces.push_back({q.from_rank, src[q.array], dst_base[q.array] + q.dst_offset_bytes, (size_t)q.bytes});
            bytes += q.bytes;
        }
        std::vector<hipStream_t> streams((size_t)num_devices);
        for (int d = 0; d < num_devices; ++d) streams[(size_t)d] = S.dv[(size_t)d].stream;
        HLM_SH_TRY(hipEventRecord(ev0, streams[0]));
        if (const std::string err = comm->gather(pieces, 0, streams); !err.empty()) return cleanup(fail(HLM_ERR_HIP, "%s", err.c_str()));
        HLM_SH_TRY(hipSetDevice(S.devs[0]));
        HLM_SH_TRY(hipEventRecord(ev1, streams[0]));
        for (int d = 0; d < num_devices; ++d) {
            HLM_SH_TRY(hipSetDevice(S.devs[(size_t)d]));
            HLM_SH_TRY(hipStreamSynchronize(streams[(size_t)d]));
        }
        float gms = 0.0f;
        HLM_SH_TRY(hipEventElapsedTime(&gms, ev0, ev1));
        if (stats) { stats->gather_ms = gms; stats->gather_bytes = bytes; }
        tp = clk::now();
        HLM_SH_TRY(hipSetDevice(S.devs[0]));
        HLM_SH_TRY(hipMemcpy(h_y_final, g_y, sizeof(double) * (size_t)ns * (size_t)neq, hipMemcpyDeviceToHost));
        if (h_stiff) HLM_SH_TRY(hipMemcpy(h_stiff, g_stiff, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost));
        if (h_n_attempt) HLM_SH_TRY(hipMemcpy(h_n_attempt, g_att, sizeof(uint32_t) * (size_t)ns, hipMemcpyDeviceToHost));
        if (h_n_accept) HLM_SH_TRY(hipMemcpy(h_n_accept, g_acc, sizeof(uint32_t) * (size_t)ns, hipMemcpyDeviceToHost));
    } else {
        for (const ShardedSetup::Shard& sh : S.shards) {
            if (sh.count == 0) continue;
            const size_t n = (size_t)sh.count, o = (size_t)sh.start;
            HLM_SH_TRY(hipSetDevice(S.devs[(size_t)sh.dev_rank]));
            HLM_SH_TRY(hipMemcpy(h_y_final + o * (size_t)neq, sh.y_final, sizeof(double) * n * (size_t)neq, hipMemcpyDeviceToHost));
            if (h_stiff) HLM_SH_TRY(hipMemcpy(h_stiff + o, sh.stiff, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            if (h_n_attempt) HLM_SH_TRY(hipMemcpy(h_n_attempt + o, sh.n_attempt, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
            if (h_n_accept) HLM_SH_TRY(hipMemcpy(h_n_accept + o, sh.n_accept, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
        }
    }
    if (stats) stats->download_ms = ms_since(tp);
#undef HLM_SH_TRY
    return cleanup(HLM_SUCCESS);
}

int hlm_run_rk45_sharded_streamed(const hlm_sharded_run* run, int num_shards, const int* devices, int num_devices,
                                  const double* h_query_times, int num_queries, int window_queries, hlm_shard_window_sink sink, void* user,
                                  double* h_y_final, int32_t* h_stiff, uint32_t* h_n_attempt, uint32_t* h_n_accept, hlm_sharded_stats* stats)
{
    using clk = std::chrono::steady_clock;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (num_queries <= 0 || !h_query_times) return fail(HLM_ERR_INVALID_ARGUMENT, "run_rk45_sharded_streamed needs query times (hlm_run_rk45_sharded runs without dense output)");
    if (window_queries <= 0) return fail(HLM_ERR_INVALID_ARGUMENT, "window_queries must be > 0");
    ShardedSetup S;
    if (const int rc = S.validate(run, num_shards, devices, num_devices, h_y_final); rc != HLM_SUCCESS) return rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_radau) return fail(HLM_ERR_INVALID_ARGUMENT, "run_rk45_sharded_streamed cannot be combined with the Radau fallback");
    }
    const int ns = S.ns, neq = S.neq;
    if (stats) {
        stats->num_shards = num_shards;
        stats->num_devices = num_devices;
        for (int d = 0; d < num_devices; ++d) stats->device_ids[d] = S.devs[(size_t)d];
    }
    if (ns == 0) return HLM_SUCCESS;
    int upload_mode, out_ncomp, out_comp[HLM_MAX_NEQ];
    {
        std::lock_guard<std::mutex> lk(g_mu);
        upload_mode = g_sharded_upload;
        out_ncomp = g_out_ncomp;
        std::memcpy(out_comp, g_out_comp, sizeof(out_comp));
    }
    if (const int rcs = validate_selection(S.uid, out_ncomp, out_comp); rcs != HLM_SUCCESS) return rcs;
    auto cleanup = [&](int rc) {
        char keep[sizeof(g_err)];
        std::memcpy(keep, g_err, sizeof(keep));
        S.release();
        std::memcpy(g_err, keep, sizeof(keep));
        return rc;
    };
    double up_ms = 0.0;
    if (const int rcu = S.upload(upload_mode, &up_ms); rcu != HLM_SUCCESS) return cleanup(rcu);
    if (stats) {
        stats->upload_ms = up_ms;
        stats->upload_mode = upload_mode;
        for (int d = 0; d < num_devices; ++d) stats->upload_ms_device[d] = S.dv[(size_t)d].upload_ms;
    }
    // Every device streams the rows of ITS shards through its own two device windows and two pinned staging buffers
    // (hlm_rk45_integrate_streamed_sink is synchronous: one host thread per device runs its shards one after the other);
    // nothing crosses between the GPUs -- the reference's ranks each write their own dense file (main.cpp:796-823).
    struct SinkCtx { hlm_shard_window_sink sink; void* user; int shard, start, count; };
    auto tramp = [](void* u, int qb, int qe, const double* rows) -> int {
        const SinkCtx* c = static_cast<const SinkCtx*>(u);
        return c->sink ? c->sink(c->user, c->shard, c->start, c->count, qb, qe, rows) : 0;
    };
    std::vector<double> wall((size_t)num_devices, 0.0), kern((size_t)num_devices, 0.0);
    std::vector<uint64_t> bytes((size_t)num_devices, 0);
    const clk::time_point tp = clk::now();
    auto device_work = [&](int d) {
        ShardedSetup::Device& D = S.dv[(size_t)d];
        auto hip_bad = [&](const char* what, hipError_t e) { D.rc = HLM_ERR_HIP; D.error = std::string(what) + " failed: " + hipGetErrorString(e); };
        if (hipError_t e = hipSetDevice(S.devs[(size_t)d]); e != hipSuccess) return hip_bad("hipSetDevice", e);
        double* d_query = nullptr;
        StreamScratch scratch;   // the device's windows and pinned staging buffers, shared by its shards
        if (hipError_t e = hipMalloc((void**)&d_query, sizeof(double) * (size_t)num_queries); e != hipSuccess) return hip_bad("hipMalloc(query)", e);
        hipError_t e = hipMemcpyAsync(d_query, h_query_times, sizeof(double) * (size_t)num_queries, hipMemcpyHostToDevice, D.stream);
        for (int s = d; s < num_shards && e == hipSuccess && D.rc == HLM_SUCCESS; s += num_devices) {
            const ShardedSetup::Shard& sh = S.shards[(size_t)s];
            if (sh.count == 0) continue;
            hlm_device_job j;
            if (const int rc0 = S.job_for(s, &j); rc0 != HLM_SUCCESS) { D.rc = rc0; D.error = g_err; break; }
            j.num_queries = num_queries;
            j.d_query = d_query;
            j.time_pieces = 0;
            j.dense_ncomp = out_ncomp;
            std::memcpy(j.dense_comp, out_comp, sizeof(out_comp));
            SinkCtx ctx{sink, user, s, sh.start, sh.count};
            hlm_stream_stats st;
            const int rc1 = streamed_sink_impl(&j, window_queries, tramp, &ctx, D.stream, &st, &scratch);
            if (rc1 != HLM_SUCCESS) { D.rc = rc1; D.error = g_err; break; }   // (g_err is thread-local: this thread's message)
            wall[(size_t)d] += st.wall_ms;
            kern[(size_t)d] += st.kernel_ms;
            bytes[(size_t)d] += st.dense_bytes;
            const size_t n = (size_t)sh.count, o = (size_t)sh.start;
            e = hipMemcpy(h_y_final + o * (size_t)neq, sh.y_final, sizeof(double) * n * (size_t)neq, hipMemcpyDeviceToHost);
            if (e == hipSuccess && h_stiff) e = hipMemcpy(h_stiff + o, sh.stiff, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
            if (e == hipSuccess && h_n_attempt) e = hipMemcpy(h_n_attempt + o, sh.n_attempt, sizeof(uint32_t) * n, hipMemcpyDeviceToHost);
            if (e == hipSuccess && h_n_accept) e = hipMemcpy(h_n_accept + o, sh.n_accept, sizeof(uint32_t) * n, hipMemcpyDeviceToHost);
        }
        (void)hipStreamSynchronize(D.stream);
        scratch.release();
        (void)hipFree(d_query);
        if (e != hipSuccess && D.rc == HLM_SUCCESS) hip_bad("sharded streamed run", e);
    };
    {
        std::vector<std::thread> th;
        for (int d = 0; d < num_devices; ++d) th.emplace_back(device_work, d);
        for (std::thread& t : th) t.join();
    }
    if (stats) {
        stats->wall_to_integrated_ms = std::chrono::duration<double, std::milli>(clk::now() - tp).count();
        for (int d = 0; d < num_devices; ++d) {
            stats->integrate_ms_device[d] = wall[(size_t)d];   // (the device's pipeline wall time: kernels and copies overlap inside it)
            if (wall[(size_t)d] > stats->integrate_ms) stats->integrate_ms = wall[(size_t)d];
            stats->dense_bytes += bytes[(size_t)d];
            stats->dense_kernel_ms += kern[(size_t)d];
        }
    }
    for (int d = 0; d < num_devices; ++d)
        if (S.dv[(size_t)d].rc != HLM_SUCCESS)
            return cleanup(fail(S.dv[(size_t)d].rc, "device %d: %s", S.devs[(size_t)d], S.dv[(size_t)d].error.c_str()));
    return cleanup(HLM_SUCCESS);
}

}  // extern "C"
