// hlm/rk45_api.hpp -- the reference's host API (namespace rk45_api) on top of libhlm_hip.so.
//
// Same names, argument order and result layout as reference src/solver/rk45_api.hpp:
//   setup_gpu_buffers<M>   :63-116   -> tuple<double*,double*,double*,double*,int*,int,int>
//   launch_rk45_kernel<M>  :120-154
//   retrieve_and_free<M>   :159-270  -> pair<FinalType, DenseType>
//   run_rk45<M>            :273-313  (declared but never instantiable in the reference --
//                                     its call sites pass the wrong number of arguments;
//                                     here it works)
// Every failure throws std::runtime_error (as :86-108,:150-153 do); numerical trouble is
// data (stiff flags), never an exception.
//
// Differences a caller can observe:
//  * the dense buffer on the device is tile-major ([tile][query][comp][lane]: whole cache lines per store,
//    csrc/hlm_rk45_dense_kernel.h) and is transposed to the host order [sys][query][comp] by a kernel on its
//    way out, so retrieve_and_free has no host-side re-order loop (:255-267);
//  * set_output_states({...}) (the reference's output.states, data/config.yaml:44-52) makes the dense rows
//    carry only the listed state components: second = dense[(s*nq + q)*K + k];
//  * never-written slots (queries at/before t0, rows of flagged systems) are zero, which
//    is what every committed reference artifact shows for its uninitialised cudaMalloc;
//  * forcings are registered once with set_forcings() instead of four cudaMemcpyToSymbol
//    calls (main.cpp:559-574);
//  * last_run_info() exposes the per-system flags and step counters of the last retrieve.
#pragma once

#include <cstdint>
#include <exception>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>
#include <utility>
#include <vector>

#include "hlm/model_registry.hpp"
#include "hlm/parameters_loader.hpp"
#include "hlm_hip.h"

namespace rk45_api {

using DenseType = std::vector<double>;
using FinalType = std::vector<double>;

struct RunInfo {
    std::vector<int32_t> stiff;        // 0 ok, 1 stiff-flagged (rk45_kernel.cu:160-170), 2 attempt cap
    std::vector<uint32_t> n_attempt;   // attempted RK45 steps per system
    std::vector<uint32_t> n_accept;
};

namespace detail {
struct Live {
    std::mutex mu;
    std::map<const void*, hlm_gpu_buffers> by_y0;  // buffers handed out by setup_gpu_buffers
};
inline RunInfo& last_info()
{
    static thread_local RunInfo info;  // per host thread: one thread may drive each GPU
    return info;
}
inline Live& live()
{
    static Live l;
    return l;
}
inline hlm_gpu_buffers adopt(double* y0, double* yf, double* q, double* dense, int* stiff, int ns, int nq,
                             bool erase)
{
    Live& L = live();
    std::lock_guard<std::mutex> lk(L.mu);
    auto it = L.by_y0.find(y0);
    if (it != L.by_y0.end() && it->second.d_y_final_all == yf) {
        hlm_gpu_buffers b = it->second;
        if (erase) L.by_y0.erase(it);
        return b;
    }
    hlm_gpu_buffers b{};  // buffers the caller allocated itself: no counter arrays
    b.d_y0_all = y0;
    b.d_y_final_all = yf;
    b.d_query_times = q;
    b.d_dense_all = dense;
    b.d_stiff = reinterpret_cast<int32_t*>(stiff);
    b.num_systems = ns;
    b.num_queries = nq;
    return b;
}
}  // namespace detail

// ---- caller-owned device memory (what main.cpp:392-404 / :552-557 do with cudaMalloc) ----
template <class T>
T* device_upload(const T* host, size_t count)
{
    void* d = nullptr;
    detail::check(hlm_device_malloc(&d, sizeof(T) * count), "device_upload: allocate");
    if (count) {
        const int rc = hlm_memcpy_h2d(d, host, sizeof(T) * count);
        if (rc != HLM_SUCCESS) { hlm_device_free(d); detail::check(rc, "device_upload: copy"); }
    }
    return static_cast<T*>(d);
}
inline void device_free(const void* d) { detail::check(hlm_device_free(const_cast<void*>(d)), "device_free"); }

// d_forc: device float[sum nT][ld]; dt in hours (main.cpp:513-514).  ld = num_systems and
// d_col = nullptr is the reference layout; a per-cell table passes ld = ncell and a device
// int32 column index per system.
inline void set_forcings(const float* d_forc, const std::vector<double>& dt_hours,
                         const std::vector<size_t>& nT, long long ld, const int32_t* d_col = nullptr)
{
    if (dt_hours.size() != nT.size()) throw std::runtime_error("set_forcings: dt/nT size mismatch");
    std::vector<uint64_t> n64(nT.begin(), nT.end());
    detail::check(hlm_set_forcings(d_forc, dt_hours.data(), n64.data(), (int)nT.size(), ld, d_col), "set_forcings");
}

// Forcing series kept in host memory and streamed through HBM in chunks of chunk_minutes by
// launch_rk45_kernel / run_rk45 (h_tables[j]: float [nT[j]][ld], valid until hlm_clear_forcings).
inline void set_forcings_host(const std::vector<const float*>& h_tables, const std::vector<double>& dt_hours,
                              const std::vector<size_t>& nT, long long ld, double chunk_minutes,
                              const int32_t* d_col = nullptr)
{
    if (dt_hours.size() != nT.size() || h_tables.size() != nT.size()) throw std::runtime_error("set_forcings_host: size mismatch");
    std::vector<uint64_t> n64(nT.begin(), nT.end());
    detail::check(hlm_set_forcings_host(h_tables.data(), dt_hours.data(), n64.data(), (int)nT.size(), ld, d_col, chunk_minutes),
                  "set_forcings_host");
}

// Output selection (data/config.yaml:44-52 "output.states"; the reference's writers take the variable list,
// I_O/output_series.cpp:18-72): the dense rows of the buffers set up AFTER this call hold only these state
// components (ascending, distinct) -- K/N_EQ of the HBM and PCIe bytes.  {} = all components (default).
inline void set_output_states(const std::vector<int>& comps)
{
    detail::check(hlm_set_output_states((int)comps.size(), comps.empty() ? nullptr : comps.data()), "set_output_states");
}

// Radau-IIA fallback for the systems RK45 flags (the pass rk45_api.hpp:189-247 of the reference
// attempts inside retrieve_and_free); off by default = the reference's observable behaviour.
inline void set_radau_fallback(bool on) { detail::check(hlm_set_radau_fallback(on ? 1 : 0), "set_radau_fallback"); }
// mode 2: the pass continues every flagged system from the state it had reached instead of restarting it at t0
// (radau_kernel.cu:44-52 restarts); 0 / 1 as set_radau_fallback(false / true)
inline void set_radau_mode(int mode) { detail::check(hlm_set_radau_fallback(mode), "set_radau_mode"); }
// hlm_set_radau_pieces: the Radau pass's work queue in time pieces (0 = automatic, 1 = whole systems, at most 16); same results
inline void set_radau_pieces(int pieces) { detail::check(hlm_set_radau_pieces(pieces), "set_radau_pieces"); }
// Launch shape of the runs without dense output (build feature, hlm_set_time_pieces): 0 = one wave per tile (the
// reference's map, rk45_kernel.cu:33), P > 1 = a persistent grid over P time pieces x tiles, negative = automatic
// (the default).  Same results bit for bit either way.
inline void set_time_pieces(int pieces) { detail::check(hlm_set_time_pieces(pieces), "set_time_pieces"); }
// What the queue of such a launch hands out and whether neighbouring tiles are packed (hlm_set_queue_shape): run_pieces
// 0 = automatic, n = runs of n pieces, -n = automatic with runs of at most n; group_tiles G > 1 = the hillslopes of G
// tiles that are still integrating share full waves -- for domains that lose hillslopes to the stiff flag.  Same results.
inline void set_queue_shape(int run_pieces, int group_tiles) { detail::check(hlm_set_queue_shape(run_pieces, group_tiles), "set_queue_shape"); }
// hlm_set_queue_hybrid: 1 = waves that hold saturated hillslopes run one block of general stages for all their lanes (domains
// whose static stores the storms fill: -14 ... -19 %; +18 % where they never overflow), -1 / 0 = a block per RHS form.  Same results.
inline void set_queue_hybrid(int mode) { detail::check(hlm_set_queue_hybrid(mode), "set_queue_hybrid"); }

// ---- 1) allocate device buffers & copy inputs ------------------------------------------------
template <class Model>
auto setup_gpu_buffers(const std::vector<double>& h_y0, const std::vector<double>& h_query_times)
{
    constexpr int N_EQ = Model::N_EQ;  // the kernels are templates over it (solver/rk45.h:51-66), one per registered model
    static_assert(N_EQ >= 1 && N_EQ <= HLM_MAX_NEQ, "Model::N_EQ outside what the device kernels support");
    hlm_gpu_buffers b{};
    detail::check(hlm_setup_gpu_buffers_model(Model::UID, h_y0.data(), int(h_y0.size() / N_EQ), h_query_times.data(),
                                              int(h_query_times.size()), &b),
                  "setup_gpu_buffers");
    {
        detail::Live& L = detail::live();
        std::lock_guard<std::mutex> lk(L.mu);
        if (b.d_y0_all) L.by_y0[b.d_y0_all] = b;
    }
    return std::make_tuple(b.d_y0_all, b.d_y_final_all, b.d_query_times, b.d_dense_all,
                           reinterpret_cast<int*>(b.d_stiff), b.num_systems, b.num_queries);
}

// ---- 2) integrate on the GPU (synchronous, like :150) ------------------------------------------
template <class Model>
void launch_rk45_kernel(double* d_y0_all, double* d_y_final_all, double* d_query_times, double* d_dense_all,
                        int* d_stiff, int num_systems, int num_queries, double t0, double tf,
                        const typename Model::SP_TYPE* d_sp)
{
    hlm_gpu_buffers b =
        detail::adopt(d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, num_systems, num_queries, false);
    if (!b.neq) b.neq = Model::N_EQ;  // buffers the caller allocated itself
    detail::check(hlm_launch_rk45_kernel(Model::UID, &b, t0, tf, d_sp), "launch_rk45_kernel");
}

// ---- 3) copy back & free ---------------------------------------------------------------------------
template <class Model>
std::pair<FinalType, DenseType> retrieve_and_free(double* d_y0_all, double* d_y_final_all, double* d_query_times,
                                                  double* d_dense_all, int* d_stiff, int num_systems,
                                                  int num_queries, double /*t0*/, double /*tf*/,
                                                  const typename Model::SP_TYPE* /*d_sp*/)
{
    constexpr int N_EQ = Model::N_EQ;
    hlm_gpu_buffers b =
        detail::adopt(d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, num_systems, num_queries, true);
    if (!b.neq) b.neq = N_EQ;
    FinalType h_final((size_t)num_systems * N_EQ);
    const int width = b.dense_ncomp > 0 ? b.dense_ncomp : N_EQ;  // doubles per dense row (set_output_states)
    DenseType h_dense((size_t)num_systems * (size_t)width * (size_t)num_queries);
    RunInfo info;
    info.stiff.resize(num_systems);
    if (b.d_n_attempt) info.n_attempt.resize(num_systems);
    if (b.d_n_accept) info.n_accept.resize(num_systems);
    detail::check(hlm_retrieve_and_free(&b, h_final.data(), h_dense.data(), info.stiff.data(),
                                        info.n_attempt.empty() ? nullptr : info.n_attempt.data(),
                                        info.n_accept.empty() ? nullptr : info.n_accept.data()),
                  "retrieve_and_free");
    detail::last_info() = std::move(info);
    return {std::move(h_final), std::move(h_dense)};
}

// ---- the three phases composed ------------------------------------------------------------------------
// first  = y_final[s*N_EQ + i];  second = dense[(s*nq + q)*N_EQ + c]
template <class Model>
std::pair<FinalType, DenseType> run_rk45(const std::vector<double>& h_y0, double t0, double tf,
                                         const std::vector<double>& h_query_times,
                                         typename Model::SP_TYPE* d_sp)
{
    auto [d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, ns, nq] =
        setup_gpu_buffers<Model>(h_y0, h_query_times);
    try {
        launch_rk45_kernel<Model>(d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, ns, nq, t0, tf, d_sp);
    } catch (...) {
        hlm_gpu_buffers b = detail::adopt(d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, ns, nq, true);
        hlm_retrieve_and_free(&b, nullptr, nullptr, nullptr, nullptr, nullptr);
        throw;
    }
    return retrieve_and_free<Model>(d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, ns, nq, t0, tf, d_sp);
}

// ---- run_rk45 for dense output that fits no host array (BASELINE config 5: 864 GB per GPU) ----------------------
// The rows reach `sink(q_begin, q_end, rows)` window by window, rows = [num_systems][q_end - q_begin][K] (K = N_EQ or the
// length of set_output_states), valid until the sink returns, while the next window integrates; a sink that throws
// ends the run and the exception is rethrown here.  Returns the final states; last_run_info() has flags / counters.
template <class Model>
FinalType run_rk45_streamed(const std::vector<double>& h_y0, double t0, double tf, const std::vector<double>& h_query_times,
                            typename Model::SP_TYPE* d_sp, int window_queries,
                            const std::function<void(int, int, const double*)>& sink)
{
    constexpr int N_EQ = Model::N_EQ;
    const int ns = int(h_y0.size() / N_EQ);
    struct Ctx {
        const std::function<void(int, int, const double*)>* fn;
        std::exception_ptr error;
    } ctx{&sink, nullptr};
    const hlm_window_sink tramp = [](void* user, int qb, int qe, const double* rows) -> int {
        Ctx* c = static_cast<Ctx*>(user);
        try {
            (*c->fn)(qb, qe, rows);
            return 0;
        } catch (...) {
            c->error = std::current_exception();
            return -1;
        }
    };
    FinalType h_final((size_t)ns * N_EQ);
    RunInfo info;
    info.stiff.resize(ns);
    info.n_attempt.resize(ns);
    info.n_accept.resize(ns);
    const int rc = hlm_run_rk45_streamed(Model::UID, h_y0.data(), ns, t0, tf, h_query_times.data(), int(h_query_times.size()), d_sp,
                                         window_queries, sink ? tramp : nullptr, &ctx, h_final.data(), info.stiff.data(),
                                         info.n_attempt.data(), info.n_accept.data());
    if (ctx.error) std::rethrow_exception(ctx.error);
    detail::check(rc, "run_rk45_streamed");
    detail::last_info() = std::move(info);
    return h_final;
}

// ---- one process, the GPUs of a node ----------------------------------------------------------------------------
// The reference splits the parameter rows over MPI worker ranks (main.cpp:258-309, 357-368) and each rank writes its
// own files; here one call takes the whole domain as host arrays, splits it by the same rule over `num_shards` shards
// on `devices`, and gathers final states, flags and step counters to devices[0] over RCCL (gather_rccl) or by one copy
// per device (hlm_run_rk45_sharded).  No dense output.  h_sp: the whole domain's SpatialParams (host); tables[j]:
// host float [nT[j]][ld]; col: host column index per hillslope (per-cell tables) or nullptr (per-stream, ld == ns).
struct ShardedStats : hlm_sharded_stats {};
template <class Model>
FinalType run_rk45_sharded(const std::vector<double>& h_y0, double t0, double tf, const typename Model::SP_TYPE* h_sp,
                           const std::vector<const float*>& tables, const std::vector<double>& dt_hours,
                           const std::vector<size_t>& nT, long long ld, const int32_t* col, int num_shards,
                           const std::vector<int>& devices, bool gather_rccl, ShardedStats* stats = nullptr)
{
    constexpr int N_EQ = Model::N_EQ;
    if (dt_hours.size() != nT.size() || tables.size() != nT.size() || nT.size() > HLM_MAX_FORCINGS)
        throw std::runtime_error("run_rk45_sharded: forcing tables / dt / nT size mismatch");
    hlm_sharded_run r{};
    r.uid = Model::UID;
    r.num_systems = int(h_y0.size() / N_EQ);
    r.t0 = t0;
    r.tf = tf;
    r.h_y0 = h_y0.data();
    r.h_sp = h_sp;
    r.h_forc_tables = tables.empty() ? nullptr : tables.data();
    for (size_t k = 0; k < nT.size(); ++k) { r.forc_dt_hours[k] = dt_hours[k]; r.forc_nT[k] = nT[k]; }
    r.nforc = int(nT.size());
    r.forc_ld = ld;
    r.h_forc_col = col;
    FinalType h_final((size_t)r.num_systems * N_EQ);
    RunInfo info;
    info.stiff.resize(r.num_systems);
    info.n_attempt.resize(r.num_systems);
    info.n_accept.resize(r.num_systems);
    detail::check(hlm_run_rk45_sharded(&r, num_shards, devices.empty() ? nullptr : devices.data(), devices.empty() ? 1 : int(devices.size()),
                                       gather_rccl ? HLM_GATHER_RCCL : HLM_GATHER_HOST, h_final.data(), info.stiff.data(),
                                       info.n_attempt.data(), info.n_accept.data(), stats),
                  "run_rk45_sharded");
    detail::last_info() = std::move(info);
    return h_final;
}

// The same with dense output (BASELINE config 5 on the GPUs of a node, hlm_run_rk45_sharded_streamed): every device streams
// the rows of its shards through its own pinned windows into sink(shard, sys_begin, sys_count, query_begin, query_end,
// rows[sys_count][W][K]) -- called from one host thread per device (calls of different devices may overlap); no dense
// rows cross between GPUs, as the reference's ranks each write their own dense file (main.cpp:796-823).
using ShardWindowSink = std::function<void(int shard, int sys_begin, int sys_count, int query_begin, int query_end, const double* rows)>;
template <class Model>
FinalType run_rk45_sharded_streamed(const std::vector<double>& h_y0, double t0, double tf, const std::vector<double>& h_query_times,
                                    int window_queries, const ShardWindowSink& sink, const typename Model::SP_TYPE* h_sp,
                                    const std::vector<const float*>& tables, const std::vector<double>& dt_hours,
                                    const std::vector<size_t>& nT, long long ld, const int32_t* col, int num_shards,
                                    const std::vector<int>& devices, ShardedStats* stats = nullptr)
{
    constexpr int N_EQ = Model::N_EQ;
    if (dt_hours.size() != nT.size() || tables.size() != nT.size() || nT.size() > HLM_MAX_FORCINGS)
        throw std::runtime_error("run_rk45_sharded_streamed: forcing tables / dt / nT size mismatch");
    hlm_sharded_run r{};
    r.uid = Model::UID;
    r.num_systems = int(h_y0.size() / N_EQ);
    r.t0 = t0;
    r.tf = tf;
    r.h_y0 = h_y0.data();
    r.h_sp = h_sp;
    r.h_forc_tables = tables.empty() ? nullptr : tables.data();
    for (size_t k = 0; k < nT.size(); ++k) { r.forc_dt_hours[k] = dt_hours[k]; r.forc_nT[k] = nT[k]; }
    r.nforc = int(nT.size());
    r.forc_ld = ld;
    r.h_forc_col = col;
    struct Ctx { const ShardWindowSink* sink; std::mutex mu; std::exception_ptr error; } ctx{&sink, {}, nullptr};
    auto tramp = [](void* u, int shard, int s0, int cnt, int qb, int qe, const double* rows) -> int {
        Ctx* c = static_cast<Ctx*>(u);
        try {
            (*c->sink)(shard, s0, cnt, qb, qe, rows);
            return 0;
        } catch (...) {   // an exception must not cross the C boundary: kept, rethrown by the caller's thread
            std::lock_guard<std::mutex> lk(c->mu);
            if (!c->error) c->error = std::current_exception();
            return 1;
        }
    };
    FinalType h_final((size_t)r.num_systems * N_EQ);
    RunInfo info;
    info.stiff.resize(r.num_systems);
    info.n_attempt.resize(r.num_systems);
    info.n_accept.resize(r.num_systems);
    const int rc = hlm_run_rk45_sharded_streamed(&r, num_shards, devices.empty() ? nullptr : devices.data(), devices.empty() ? 1 : int(devices.size()),
                                                 h_query_times.data(), int(h_query_times.size()), window_queries, sink ? +tramp : nullptr, &ctx,
                                                 h_final.data(), info.stiff.data(), info.n_attempt.data(), info.n_accept.data(), stats);
    if (ctx.error) std::rethrow_exception(ctx.error);
    detail::check(rc, "run_rk45_sharded_streamed");
    detail::last_info() = std::move(info);
    return h_final;
}

// flags / step counters of the most recent retrieve_and_free on the calling thread
inline RunInfo last_run_info() { return detail::last_info(); }

}  // namespace rk45_api
