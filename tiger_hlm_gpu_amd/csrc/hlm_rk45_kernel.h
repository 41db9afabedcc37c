// The hot path: per-hillslope adaptive Dormand-Prince RK45 with dense output and
// stiffness flagging, one wavefront lane per ODE system (CDNA4 / gfx950, wave64).
//
// Restates reference src/solver/rk45_kernel.cu:17-176 (loop), rk45_step_dense.cuh:33-145
// (step), :171-244 (dense interpolant), event_detector.cuh:46-53 (slope-jump norm).
// See DESIGN.md "Kernel" for the register/LDS/HBM plan.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hlm_hip.h"
#include "hlm_models.h"
#include "hlm_tableau.h"

namespace hlm {

// Waves per SIMD the register allocator is asked to make room for.  2 (<= 256 VGPRs) fits every
// variant without spills.  The FAST kernel without dense output needs 182 VGPRs; capping it at 168
// for a third wave spills 14 of them (60 B/lane of scratch, touched outside the stage arithmetic)
// and is 6 % faster (176 -> 165 ms on the C3 pass): the extra wave fills issue slots the other two
// leave while they wait on each other's VALU results.  The dense and STRICT variants need 230-256
// VGPRs and would spill into the loop (measured earlier: 276 ms vs 206 ms), so they stay at 2.
#ifndef HLM_MIN_WAVES_PER_SIMD
#define HLM_MIN_WAVES_PER_SIMD 2
#endif
#ifndef HLM_MIN_WAVES_PER_SIMD_LEAN
#define HLM_MIN_WAVES_PER_SIMD_LEAN 3
#endif

constexpr int WAVE = 64;
constexpr int SYS_PAUSED = 3;  // internal: lane stopped at the end of its query window

// resume record of a model with N state variables (SoA over systems): N + 2 doubles {y[N], t, h}
// then 5 int32 {reject_count, n_attempt, n_accept, next_q, status}
constexpr int resume_f64(int n_eq) { return n_eq + 2; }
constexpr int RESUME_I32 = 5;

// Kernel arguments (kernarg segment -> SGPRs; everything here is wave-uniform).
struct Rk45Args {
    const double* y0;       // [ns][N_EQ]
    const double* query;    // [nq]
    const void* sp;         // [ns] x 136 B
    const float* forc;      // [sum nT][ld]
    const int32_t* forc_col;
    long long forc_ld;
    double forc_dtmin[HLM_KERNEL_MAX_NF];          // dt_hours * 60 (rk45_kernel.cu:90), the forcings a model can read
    unsigned long long forc_nT[HLM_KERNEL_MAX_NF];
    unsigned long long forc_base[HLM_KERNEL_MAX_NF];  // sum_{k<j} rows_k * ld (rk45_kernel.cu:101-104)
    unsigned long long forc_first[HLM_KERNEL_MAX_NF]; // sample index of the first row present in this block
    unsigned long long forc_rows[HLM_KERNEL_MAX_NF];  // rows present (= nT unless the host streams time chunks)
    int nforc;
    int ns, nq;
    double t0, tf;
    double t_stop;                    // lanes pause when t >= t_stop (+inf: run to tf)
    double h0, rtol, atol, safety, min_scale, max_scale;  // Model::Parameters
    double fixed_step;
    uint32_t cap;
    double* y_final;     // [ns][N_EQ]
    double* dense;       // rows of this launch's query window, linear [ns][W][N_EQ] or tile-major
    int dense_tiled;     // 0: [ns][W][N_EQ]; 2: [ceil(ns/64)][W][N_EQ][64] (W = q_end - q_begin)
    int q_begin, q_end;  // global query indices [q_begin, q_end) written by this launch
    double* resume;      // nullable: per-system integrator state carried between launches
    int resume_in;       // 0 = start from y0, 1 = continue from `resume`
    int32_t* stiff;
    uint32_t* n_attempt;
    uint32_t* n_accept;
    // Output selection (SEL kernel variants only; hlm_device_job.dense_ncomp): a dense row holds dense_k
    // components; dense_pos[c] = position of state component c in the row, or -1 when it is not written.
    int dense_k;
    int dense_pos[HLM_MAX_NEQ];
    // nullable, [ns][N_EQ + 1]: a lane that is flagged stiff records the state it had reached, {y[N_EQ], t}, for a
    // Radau pass that continues from there instead of restarting at t0 (hlm_device_job.d_flag_state)
    double* flag_state;
    // nullable, [PATH_COUNTERS] (COUNT instances of the queue kernel only; hlm_device_job.d_path_counts): lane-attempts per
    // RHS form, summed over the launch -- see LanePathCount
    unsigned long long* path_counts;
};

// The launch arguments read again from the kernel-argument segment (constant address space: scalar loads), field by
// field -- for persistent kernels that want an item's view of them to live inside the item (see rk45_dense_queue_kernel).
// EVERY field of Rk45Args is copied here: a field added there is added here.
using Rk45ArgsConst = const Rk45Args __attribute__((address_space(4)));
__device__ __forceinline__ Rk45Args load_args(Rk45ArgsConst* p)
{
    Rk45Args a;
    a.y0 = p->y0; a.query = p->query; a.sp = p->sp; a.forc = p->forc; a.forc_col = p->forc_col; a.forc_ld = p->forc_ld;
#pragma unroll
    for (int k = 0; k < HLM_KERNEL_MAX_NF; ++k) {
        a.forc_dtmin[k] = p->forc_dtmin[k]; a.forc_nT[k] = p->forc_nT[k]; a.forc_base[k] = p->forc_base[k];
        a.forc_first[k] = p->forc_first[k]; a.forc_rows[k] = p->forc_rows[k];
    }
    a.nforc = p->nforc; a.ns = p->ns; a.nq = p->nq; a.t0 = p->t0; a.tf = p->tf; a.t_stop = p->t_stop;
    a.h0 = p->h0; a.rtol = p->rtol; a.atol = p->atol; a.safety = p->safety; a.min_scale = p->min_scale; a.max_scale = p->max_scale;
    a.fixed_step = p->fixed_step; a.cap = p->cap; a.y_final = p->y_final; a.dense = p->dense; a.dense_tiled = p->dense_tiled;
    a.q_begin = p->q_begin; a.q_end = p->q_end; a.resume = p->resume; a.resume_in = p->resume_in; a.stiff = p->stiff;
    a.n_attempt = p->n_attempt; a.n_accept = p->n_accept; a.dense_k = p->dense_k;
#pragma unroll
    for (int c = 0; c < HLM_MAX_NEQ; ++c) a.dense_pos[c] = p->dense_pos[c];
    a.flag_state = p->flag_state;
    a.path_counts = p->path_counts;
    return a;
}

// y + sum c_j * k_j accumulated left to right with one FMA per term, the order the
// reference's "acc += h*a*k" loop uses (rk45_step_dense.cuh:96-101).
__device__ __forceinline__ double lc(double acc) { return acc; }
template <class... Rest>
__device__ __forceinline__ double lc(double acc, double c, double k, Rest... rest)
{
    return lc(__builtin_fma(c, k, acc), rest...);
}

// 1/x for the FAST error ratio: v_rcp_f64 seed (~2^-26) + one Newton step.
__device__ __forceinline__ double fast_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

// (1/x)^(1/5) for the step-size controller, rk45_kernel.cu:151,156:
// pow(1.0/(err+1e-16), 0.2).  STRICT keeps the divide and takes the correctly rounded power for the double
// exponent 0.2 (hlm_strict_pow.h).
template <int MATH>
__device__ __forceinline__ double inv_fifth_root(double x)
{
    if constexpr (MATH == MATH_STRICT) {
        return strict_pow_fifth(1.0 / x);
    } else {
        // f32 exp2/log2 seed z ~ x^(-1/5) (x in [1e-16, huge]), residual e = 1 - x z^5,
        // exact root = z (1-e)^(-1/5) = z (1 + e/5 + 3/25 e^2 + 11/125 e^3 + O(e^4)).
        const float xf = (float)x;
        double z = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(xf));
        const double z2 = z * z;
        const double e = __builtin_fma(-x, z2 * z2 * z, 1.0);
        double p = __builtin_fma(e, 11.0 / 125.0, 3.0 / 25.0);
        p = __builtin_fma(e, p, 0.2);
        z = __builtin_fma(z * e, p, z);
        return z;
    }
}

// zero-order-hold forcing index, rk45_kernel.cu:90-98
__device__ __forceinline__ unsigned long long forcing_index(double t, double dtmin,
                                                            unsigned long long nT)
{
    const double r = t / dtmin;
    return (r < 0.0) ? 0ull : (r >= (double)nT ? nT - 1ull : (unsigned long long)r);
}

// row of the launch's forcing block that holds global sample `idx` (the block may be a time
// chunk [first, first+rows) of the series; clamped so that a bad chunk can never fault)
__device__ __forceinline__ size_t forcing_row(unsigned long long idx, unsigned long long first,
                                              unsigned long long rows)
{
    const unsigned long long r = idx >= first ? idx - first : 0ull;
    return (size_t)(r < rows ? r : rows - 1ull);
}

// device-coherent accesses to the lane states handed from wave to wave INSIDE a launch (the queue kernels: any wave of
// any XCD may continue a tile): agent-scope loads and stores, no cache write-back or invalidate anywhere
__device__ __forceinline__ double coherent_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int coherent_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- dense output of an accepted step ------------------------------------------------------------------------
// The reference evaluates y(t + theta h) = y + h sum_{m=0..3} Q_m theta^(m+1) with Q = P^T k, P = SciPy's 7 x 4
// matrix (rk45_step_dense.cuh:193-244).  That quartic is the continuous extension of the Dormand-Prince pair in
// monomial form; Hairer's DOPRI5 writes THE SAME polynomial (checked coefficient by coefficient against P: they
// agree to 2e-15, the rounding of the decimal constants) as
//     y + theta (d1 + (1 - theta) (d2 + theta (d3 + (1 - theta) d4))),
//     d1 = y_new - y,  d2 = h k0 - d1,  d3 = d1 - h k6 - d2,  d4 = h sum_j P[j][3] k_j
// -- one six-term combination of the slopes instead of three (10 instead of 18 operations per component and
// step), and 4 instead of 5 fused operations per component and row.  The rows differ from the monomial form's in
// the last bits only (errors of a few ulp of |y|; the parity bound is 1e-6 |y|); they never feed back into the
// integration, so the step sequence is untouched.  Both dense kernels use these two functions: their rows stay
// bit-identical to each other.
// d2 MAY BE k0 ITSELF (the ring kernel: a lane holding an accepted step has no use for k0 until the commit replaces
// it by k6, and a second set of registers for d2 would spill): k0[i] is read before d2[i] is written.
template <int NEQ>
__device__ __forceinline__ void dense_coefficients(const double (&y)[NEQ], const double (&yn)[NEQ], double h, const double (&k0)[NEQ],
                                                   const double (&k2)[NEQ], const double (&k3)[NEQ], const double (&k4)[NEQ],
                                                   const double (&k5)[NEQ], const double (&k6)[NEQ], double (&d1)[NEQ],
                                                   double (&d2)[NEQ], double (&d3)[NEQ], double (&d4)[NEQ])
{
    using namespace dp;
    const double h0 = h * P03, h2 = h * P23, h3 = h * P33, h4 = h * P43, h5 = h * P53, h6 = h * P63;
#pragma unroll
    for (int i = 0; i < NEQ; ++i) {
        const double k0i = k0[i];
        d4[i] = lc(h0 * k0i, h2, k2[i], h3, k3[i], h4, k4[i], h5, k5[i], h6, k6[i]);
        d1[i] = yn[i] - y[i];
        const double d2i = __builtin_fma(h, k0i, -d1[i]);
        d3[i] = __builtin_fma(-h, k6[i], d1[i]) - d2i;
        d2[i] = d2i;
    }
}
__device__ __forceinline__ double dense_value(double th, double th1, double y, double d1, double d2, double d3, double d4)
{
    return __builtin_fma(th, __builtin_fma(th1, __builtin_fma(th, __builtin_fma(th1, d4, d3), d2), d1), y);
}

// where a flagged lane leaves the state it had reached (Rk45Args.flag_state)
template <int NEQ>
__device__ __forceinline__ void record_flag_state(double* flag_state, int sys, const double (&y)[NEQ], double t)
{
    double* o = flag_state + (size_t)sys * (NEQ + 1);
#pragma unroll
    for (int i = 0; i < NEQ; ++i) o[i] = y[i];
    o[NEQ] = t;
}

// ---- the 7 stages of one attempt ---------------------------------------------------------
// k0 = rhs(y) unless still valid, k1..k6, and yt = the stage-6 state (= the 5th-order solution).
// PATH selects the RHS variant: PATH_GENERAL = Model::rhs; PATH_DRY = Model::rhs_fast, exact while
// Model::fastpath_ok holds -- and on that path component Model::FAST_ZERO_COMP has zero slope in
// every stage, so it is left out of the stage arithmetic (y + sum h*a*0 == y, bit for bit).  The
// return value tells whether any stage left the regime (always false for PATH_GENERAL).
// PATH_NORAIN = Model::rhs_norain: the dry path's regime with no rainfall in the held sample.
// PATH_WETU = Model::rhs_wetu: the unsaturated regime with a WET surface store -- the surface store moves (it only
// drains: -h w(h)), nothing is pinned, the cascade below the static store is idle as on PATH_NORAIN.  Only the
// per-lane selection (attempt_per_lane_path) takes it.
constexpr int PATH_GENERAL = 0, PATH_DRY = 1, PATH_NORAIN = 2, PATH_WETU = 3;
// PATH_GENERAL_WET: Model::rhs as PATH_GENERAL, and the return value tells whether any evaluated stage found the static
// store overflowing (Model::rhs_report_wet) -- what PATH_NORAIN / PATH_WETU report as "left the regime" (rk_attempt_hybrid)
constexpr int PATH_GENERAL_WET = 4;
constexpr bool path_pins(int path) { return path == PATH_DRY || path == PATH_NORAIN; }

// index of the c-th moving component when component Z is pinned (identity for Z >= N)
constexpr int moving_component(int c, int Z) { return c < Z ? c : c + 1; }

// CLOSED (no-rain path, FAST, launches that need no stage slopes afterwards): the components the model declares
// linear and homogeneous on that path (Model::NORAIN_LINEAR_COMP: pure decay y' = lambda y while nothing flows in)
// are left out of the stage arithmetic as well -- the method applied to y' = lambda y has a closed form, y R(lambda h)
// with R the stability polynomial, evaluated instead (hlm_tableau.h: LINEAR).
template <class Model, int PATH, bool CLOSED>
constexpr bool through_stages(int i)
{
    if ((PATH == 1 || PATH == 2) /* the paths that pin the surface store */ && i == Model::FAST_ZERO_COMP) return false;
    if (CLOSED)
        for (int k = 0; k < Model::NORAIN_LINEAR; ++k)
            if (i == Model::NORAIN_LINEAR_COMP[k]) return false;
    return true;
}

template <class Model, int MATH, int PATH, bool CLOSED = false>
__device__ __forceinline__ bool rk_stages(const typename Model::Local& P, const typename Model::Forc& FV,
                                          const double (&y)[Model::N_EQ], double h, bool have_k0, double (&k0)[Model::N_EQ],
                                          double (&k1)[Model::N_EQ], double (&k2)[Model::N_EQ], double (&k3)[Model::N_EQ],
                                          double (&k4)[Model::N_EQ], double (&k5)[Model::N_EQ], double (&k6)[Model::N_EQ],
                                          double (&yt)[Model::N_EQ])
{
    using namespace dp;
    constexpr int NEQ = Model::N_EQ;
    // components that move: all of them, or all but the one the fast path pins
    constexpr int Z = path_pins(PATH) ? Model::FAST_ZERO_COMP : NEQ;
    bool broke = false;
    auto f = [&](const double (&state)[NEQ], double (&slope)[NEQ]) {
        if constexpr (PATH == PATH_NORAIN) broke |= Model::template rhs_norain<MATH>(P, FV, state, slope);
        else if constexpr (PATH == PATH_WETU) broke |= Model::template rhs_wetu<MATH>(P, FV, state, slope);
        else if constexpr (PATH == PATH_DRY) broke |= Model::template rhs_fast<MATH>(P, FV, state, slope);
        else if constexpr (PATH == PATH_GENERAL_WET) broke |= Model::template rhs_report_wet<MATH>(P, FV, state, slope);
        else Model::template rhs<MATH>(P, FV, state, slope);
    };
    // components that stand still in the stage states: the pinned one, and (CLOSED) the linear ones, whose slopes the
    // RHS calls below then compute from an unchanged state and nobody reads
#pragma unroll
    for (int i = 0; i < NEQ; ++i)
        if (!through_stages<Model, PATH, CLOSED>(i)) yt[i] = y[i];
    if (!have_k0) f(y, k0);  // no FSAL in the reference: see the have_k0 comment in the kernel
    // a pinned component must really stand still: a carried-over k0 (k6 of a step the general path
    // took) can hold the first inflow into a still-dry store, which the stage states below would ignore
    if constexpr (path_pins(PATH)) broke |= k0[Z] != 0.0;
    {
        const double a10 = h * A10;
#pragma unroll
        for (int i = 0; i < NEQ; ++i) if (through_stages<Model, PATH, CLOSED>(i)) yt[i] = lc(y[i], a10, k0[i]);
        f(yt, k1);
    }
    {
        const double a20 = h * A20, a21 = h * A21;
#pragma unroll
        for (int i = 0; i < NEQ; ++i) if (through_stages<Model, PATH, CLOSED>(i)) yt[i] = lc(y[i], a20, k0[i], a21, k1[i]);
        f(yt, k2);
    }
    {
        const double a30 = h * A30, a31 = h * A31, a32 = h * A32;
#pragma unroll
        for (int i = 0; i < NEQ; ++i) if (through_stages<Model, PATH, CLOSED>(i)) yt[i] = lc(y[i], a30, k0[i], a31, k1[i], a32, k2[i]);
        f(yt, k3);
    }
    {
        const double a40 = h * A40, a41 = h * A41, a42 = h * A42, a43 = h * A43;
#pragma unroll
        for (int i = 0; i < NEQ; ++i) if (through_stages<Model, PATH, CLOSED>(i)) yt[i] = lc(y[i], a40, k0[i], a41, k1[i], a42, k2[i], a43, k3[i]);
        f(yt, k4);
    }
    {
        const double a50 = h * A50, a51 = h * A51, a52 = h * A52, a53 = h * A53, a54 = h * A54;
#pragma unroll
        for (int i = 0; i < NEQ; ++i)
            if (through_stages<Model, PATH, CLOSED>(i)) yt[i] = lc(y[i], a50, k0[i], a51, k1[i], a52, k2[i], a53, k3[i], a54, k4[i]);
        f(yt, k5);
    }
    {
        // stage 6 state: row 6 of a is b (rk45_step_dense.cuh:66-68)
        const double hb0 = h * B0, hb2 = h * B2, hb3 = h * B3, hb4 = h * B4, hb5 = h * B5;
#pragma unroll
        for (int i = 0; i < NEQ; ++i)
            if (through_stages<Model, PATH, CLOSED>(i)) yt[i] = lc(y[i], hb0, k0[i], hb2, k2[i], hb3, k3[i], hb4, k4[i], hb5, k5[i]);
        if constexpr (CLOSED) {
            // the linear components' 5th-order solution y R(z), z = lambda h, before the last RHS call (k6 = rhs(yt))
#pragma unroll
            for (int k = 0; k < Model::NORAIN_LINEAR; ++k) {
                const int i = Model::NORAIN_LINEAR_COMP[k];
                const double lam = Model::norain_rate(P, k), z = lam * h;
                // a slope carried over from another path's attempt may hold an inflow this path does not have
                broke |= k0[i] != lam * y[i];
                double q = __builtin_fma(LIN_R[6], z, LIN_R[5]);
                q = __builtin_fma(q, z, LIN_R[4]);
                q = __builtin_fma(q, z, LIN_R[3]);
                q = __builtin_fma(q, z, LIN_R[2]);
                q = __builtin_fma(q, z, LIN_R[1]);
                yt[i] = __builtin_fma(y[i], q * z, y[i]);
            }
        }
        f(yt, k6);
    }
    return broke;
}

// ---- one attempt: the stages plus the two norms the step control needs ------------------------
// err  = scaled infinity-norm of the embedded error estimate (rk45_step_dense.cuh:123-142; "ratio >
//        max" comparisons ignore NaN components exactly like the reference);
// jump = |k0 - k1|_inf for the slope-jump detector (rk45_kernel.cu:132-136, event_detector.cuh:46-53).
// They are computed inside the path-specific code so that a pinned component (zero slope in every
// stage: error term 0, tolerance atol, slope difference 0) drops out at compile time, and so that
// only y_new, k0 and k6 -- not all seven slope vectors -- have to meet in the same registers where
// the paths join.
template <class Model, int MATH, int PATH, bool FIXED, bool CLOSED = false>
__device__ __forceinline__ bool rk_attempt(const typename Model::Local& P, const typename Model::Forc& FV,
                                           const Rk45Args& A, const double (&y)[Model::N_EQ], double h, bool have_k0,
                                           double (&k0)[Model::N_EQ], double (&k1)[Model::N_EQ], double (&k2)[Model::N_EQ],
                                           double (&k3)[Model::N_EQ], double (&k4)[Model::N_EQ], double (&k5)[Model::N_EQ],
                                           double (&k6)[Model::N_EQ], double (&yt)[Model::N_EQ], double& err, double& jump)
{
    using namespace dp;
    constexpr int NEQ = Model::N_EQ;
    const bool broke = rk_stages<Model, MATH, PATH, CLOSED>(P, FV, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt);
    // closed-form error estimate y E(z) and first slope difference k0 - k1 = -k0 A10 z of the linear components
    [[maybe_unused]] double e_lin[NEQ], j_lin[NEQ];
    if constexpr (CLOSED && !FIXED) {
#pragma unroll
        for (int k = 0; k < Model::NORAIN_LINEAR; ++k) {
            const int i = Model::NORAIN_LINEAR_COMP[k];
            const double z = Model::norain_rate(P, k) * h, z2 = z * z, z5 = z2 * z2 * z;
            e_lin[i] = (y[i] * z5) * __builtin_fma(__builtin_fma(LIN_E7, z, LIN_E6), z, LIN_E5);
            j_lin[i] = k0[i] * (A10 * z);
        }
    }
    auto closed = [](int i) { return CLOSED && !(path_pins(PATH) && i == Model::FAST_ZERO_COMP) && !through_stages<Model, PATH, true>(i); };
    if constexpr (!FIXED) {
        constexpr int Z = path_pins(PATH) ? Model::FAST_ZERO_COMP : -1;  // pinned component, if any
        const double he0 = h * E0, he2 = h * E2, he3 = h * E3, he4 = h * E4, he5 = h * E5, he6 = h * E6;
        // yt is the 5th-order solution: b[6] = 0 makes it the same chain of operations as the
        // stage-6 state (rk45_step_dense.cuh:111-117)
        if constexpr (MATH == MATH_FAST && NEQ == 5) {
            // max_i |e_i| / tol_i with ONE reciprocal: max_i(|e_i| * prod_{j != i} tol_j) / prod_j tol_j
            // (tol_i in [atol, ~rtol*|y|]: the 5-fold products stay far from the double range); the product
            // tree is written out for the 5-component models
            double e[NEQ], tl[NEQ];
#pragma unroll
            for (int i = 0; i < NEQ; ++i) {
                if (i == Z) {
                    e[i] = 0.0;
                    tl[i] = A.atol;  // fma(rtol, max(|0|, |0|), atol)
                } else {
                    e[i] = closed(i) ? e_lin[i] : lc(he0 * k0[i], he2, k2[i], he3, k3[i], he4, k4[i], he5, k5[i], he6, k6[i]);
                    tl[i] = __builtin_fma(A.rtol, vmax_abs(y[i], yt[i]), A.atol);
                }
            }
            const double a = tl[0] * tl[1], b = tl[2] * tl[3];
            const double ab = a * b, all = ab * tl[4], bt4 = b * tl[4], at4 = a * tl[4];
            const double others[NEQ] = {tl[1] * bt4, tl[0] * bt4, tl[3] * at4, tl[2] * at4, ab};
            double num = 0.0;
#pragma unroll
            for (int i = 0; i < NEQ; ++i)
                if (i != Z) num = vmax_abs2(num, e[i] * others[i]);
            err = num * fast_rcp(all);
        } else if constexpr (MATH == MATH_FAST) {
            err = 0.0;  // any other N_EQ: one Newton-refined reciprocal per component
#pragma unroll
            for (int i = 0; i < NEQ; ++i) {
                if (i == Z) continue;
                const double y_err = closed(i) ? e_lin[i] : lc(he0 * k0[i], he2, k2[i], he3, k3[i], he4, k4[i], he5, k5[i], he6, k6[i]);
                err = vmax_abs2(err, y_err * fast_rcp(__builtin_fma(A.rtol, vmax_abs(y[i], yt[i]), A.atol)));
            }
        } else {
            err = 0.0;
#pragma unroll
            for (int i = 0; i < NEQ; ++i) {
                if (i == Z) continue;  // 0 / atol
                const double y_err = lc(he0 * k0[i], he2, k2[i], he3, k3[i], he4, k4[i], he5, k5[i], he6, k6[i]);
                const double tol = __builtin_fma(A.rtol, vmax_abs(y[i], yt[i]), A.atol);
                err = vmax_abs2(err, y_err / tol);  // "if (fabs(ratio) > max) max = fabs(ratio)"
            }
        }
        jump = 0.0;
#pragma unroll
        for (int i = 0; i < NEQ; ++i)
            if (i != Z) jump = vmax_abs2(jump, closed(i) ? j_lin[i] : k0[i] - k1[i]);
    }
    return broke;
}

// ---- held forcing samples -----------------------------------------------------------------------
// Zero-order hold sampled at the START time of a step and held for all its stages
// (rk45_kernel.cu:85-106).  Each lane keeps the held samples F0/F1 and the time at which its sample
// index next changes; the index (an f64 divide) and the 4-byte load are redone only when an accepted
// step carries t past that time -- about once per forcing interval instead of twice per attempt.
// (floor(t/dt) == idx for t < tb up to a final-ulp tie of the quotient, a measure-zero difference
// from the reference.)
template <class Model, bool RESUMABLE>
struct HeldForcing {
    static constexpr int NF = Model::NF, NFA = NF > 0 ? NF : 1;
    static_assert(NF <= HLM_KERNEL_MAX_NF, "a model may read at most HLM_KERNEL_MAX_NF forcings");
    size_t fcol = 0;
    double tb[NFA];                     // when sample j expires
    double tb_next = -__builtin_inf();  // min_j tb[j]: when the held samples expire
    double F[NFA];                      // the held samples, widened to double (model_204.hpp:82-83)

    __device__ __forceinline__ HeldForcing()
    {
#pragma unroll
        for (int j = 0; j < NFA; ++j) { tb[j] = -__builtin_inf(); F[j] = 0.0; }
    }
    __device__ __forceinline__ void init(const Rk45Args& A, int sys)
    {
        if constexpr (NF > 0) fcol = A.forc_col ? (size_t)A.forc_col[sys] : (size_t)sys;
    }
    // One rarely-taken branch per attempt: has t (which only moves on an accepted step) reached the
    // end of a held sample?  Returns true when a sample changed (FV was rebuilt; k0 is stale).
    // A forcing the host did not register (j >= nforc) stays 0, as model_204.hpp:82-83 has it.
    __device__ __forceinline__ bool refresh(const Rk45Args& A, const typename Model::Local& P, double t,
                                            typename Model::Forc& FV)
    {
        if constexpr (NF > 0) {
            if (!(t < tb_next)) {
                double next = __builtin_inf();
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if (j < A.nforc) {
                        if (!(t < tb[j])) {
                            const unsigned long long i = forcing_index(t, A.forc_dtmin[j], A.forc_nT[j]);
                            tb[j] = (i + 1ull < A.forc_nT[j]) ? (double)(i + 1ull) * A.forc_dtmin[j] : __builtin_inf();
                            F[j] = (double)A.forc[A.forc_base[j] + (RESUMABLE ? forcing_row(i, A.forc_first[j], A.forc_rows[j]) : (size_t)i) * (size_t)A.forc_ld + fcol];
                        }
                        next = fmin(next, tb[j]);
                    }
                }
                tb_next = next;
                FV = Model::prep_forcing(P, F);
                return true;
            }
        }
        return false;
    }
};

// ---- one attempt on the cheapest RHS path the whole wave qualifies for ----------------------------
// A model may offer a cheaper RHS that is exact while its per-lane precondition holds (Model 204:
// surface store == 0, the Manning term vanishes; and no rain on top of that).  The wave takes it only
// when ALL its active lanes qualify, and every stage reports whether the precondition broke during the
// step (inflow reached the surface store); if it did for any lane the attempt is simply recomputed
// with the general RHS -- same bits either way.  On return have_k0 is true (k0 stays valid through a
// reject / slope-jump retry) and yt is the 5th-order solution.
// The same for launches that do not read the stage slopes afterwards (no dense output), FAST arithmetic, models with
// linear components on the no-rain path: that path then takes those components in closed form, which rounds
// differently from the stage-by-stage arithmetic of the other two -- so WHICH path integrates a hillslope may no longer
// depend on its 63 neighbours, or the results would depend on the launch shape and on where a shard ends.  Here every
// lane decides for itself: the lanes that qualify run the unsaturated path -- its dry-surface form (PATH_NORAIN) or its
// wet-surface form (PATH_WETU), each under its mask when the wave is mixed -- and those that did not qualify or left
// the regime during the step (their own tests) are integrated by the dry / general paths -- which of the two is still
// decided per wave: they agree bit for bit.
template <class Model, int MATH, bool FIXED>
__device__ __forceinline__ void attempt_per_lane_path(const typename Model::Local& P, const typename Model::Forc& FV,
                                                      const Rk45Args& A, const double (&y)[Model::N_EQ], double h, bool& have_k0,
                                                      double (&k0)[Model::N_EQ], double (&k1)[Model::N_EQ], double (&k2)[Model::N_EQ],
                                                      double (&k3)[Model::N_EQ], double (&k4)[Model::N_EQ], double (&k5)[Model::N_EQ],
                                                      double (&k6)[Model::N_EQ], double (&yt)[Model::N_EQ], double& err, double& jump)
{
    bool stagewise = true;  // this lane still needs the dry / general path
    if (Model::norainpath_ok(P, FV, y)) {
        const bool had_k0 = have_k0;
        const bool broke = rk_attempt<Model, MATH, PATH_NORAIN, FIXED, true>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        stagewise = broke;
        have_k0 = broke ? had_k0 : true;  // (a k0 evaluated in an attempt that falls back is not kept: see below)
    }
    else if (Model::wetupath_ok(P, FV, y)) {
        const bool had_k0 = have_k0;
        const bool broke = rk_attempt<Model, MATH, PATH_WETU, FIXED, true>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        stagewise = broke;
        have_k0 = broke ? had_k0 : true;
    }
    if (stagewise) {
        bool general = true;
        if constexpr (Model::HAS_FASTPATH) {
            if (__builtin_amdgcn_ballot_w64(!Model::fastpath_ok(y)) == 0ull) {  // (among the lanes in here)
                const bool broke = rk_attempt<Model, MATH, PATH_DRY, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
                have_k0 = true;  // k0 of a qualifying state is exact
                general = __builtin_amdgcn_ballot_w64(broke) != 0ull;
            }
        }
        if (general) rk_attempt<Model, MATH, PATH_GENERAL, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
    }
    have_k0 = true;
}

// ---- one attempt on the GENERAL stages for the whole wave, every lane keeping the result of ITS form -------------------
// attempt_per_lane_path runs a block per RHS form present in the wave, each under its lanes' mask, and a lane that
// leaves its unsaturated form inside the step is computed twice: on capacities the storms fill (10 % of the
// lane-attempts on the general RHS, all of them computed twice; 40 % of the wave-attempts on several forms) a
// wave-attempt costs 726 VALU instructions against 476 of the wet unsaturated form alone (profiles/r05_pmc_issue.json).
// Here the wave runs ONE block: the general RHS through all seven stages for every lane -- for an unsaturated lane its
// snow / static / surface stores see exactly the operations of its own form (the idle cascade contributes literal
// zeros; the stage-wise forms agree with the general one bit for bit, tests) -- plus the closed form of the two
// draining stores, and each lane keeps what attempt_per_lane_path would have given it: the closed-form values if it
// qualifies for an unsaturated form and no stage found its static store overflowing (the regime test of rhs_norain /
// rhs_wetu, Model::saturated_at) and the carried k0 is that form's, the stage values otherwise.  Same bits per lane,
// whatever its neighbours do.  FAST arithmetic, models with linear stores (Model 204).
template <class Model, int MATH, bool FIXED>
__device__ __forceinline__ void rk_attempt_hybrid(const typename Model::Local& P, const typename Model::Forc& FV, const Rk45Args& A,
                                                  const double (&y)[Model::N_EQ], double h, bool have_k0, double (&k0)[Model::N_EQ],
                                                  double (&k1)[Model::N_EQ], double (&k2)[Model::N_EQ], double (&k3)[Model::N_EQ],
                                                  double (&k4)[Model::N_EQ], double (&k5)[Model::N_EQ], double (&k6)[Model::N_EQ],
                                                  double (&yt)[Model::N_EQ], double& err, double& jump)
{
    using namespace dp;
    constexpr int NEQ = Model::N_EQ, NL = Model::NORAIN_LINEAR;
    static_assert(MATH == MATH_FAST && NEQ == 5 && NL > 0, "the forms this replaces exist in FAST arithmetic for the 5-store model");
    const bool wet_any = rk_stages<Model, MATH, PATH_GENERAL_WET, false>(P, FV, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt);
    // would attempt_per_lane_path have kept this lane's closed-form attempt?  (its `broke`, term by term)
    bool closed_ok = Model::unsat_eligible(P, y) && !wet_any;
    if (y[Model::FAST_ZERO_COMP] == 0.0) closed_ok = closed_ok && k0[Model::FAST_ZERO_COMP] == 0.0;   // (the dry-surface form pins that store)
#pragma unroll
    for (int k = 0; k < NL; ++k)   // (a slope carried over from a general attempt may hold an inflow the closed form does not have)
        closed_ok = closed_ok && k0[Model::NORAIN_LINEAR_COMP[k]] == Model::norain_rate(P, k) * y[Model::NORAIN_LINEAR_COMP[k]];
    double e_lin[NL], j_lin[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = Model::NORAIN_LINEAR_COMP[k];
        const double lam = Model::norain_rate(P, k), z = lam * h, z2 = z * z, z5 = z2 * z2 * z;
        double q = __builtin_fma(LIN_R[6], z, LIN_R[5]);
        q = __builtin_fma(q, z, LIN_R[4]);
        q = __builtin_fma(q, z, LIN_R[3]);
        q = __builtin_fma(q, z, LIN_R[2]);
        q = __builtin_fma(q, z, LIN_R[1]);
        const double yc = __builtin_fma(y[i], q * z, y[i]);   // y R(z): rk_stages, CLOSED
        e_lin[k] = (y[i] * z5) * __builtin_fma(__builtin_fma(LIN_E7, z, LIN_E6), z, LIN_E5);
        j_lin[k] = k0[i] * (A10 * z);
        // (k6 of the closed form: the unsaturated RHS at the new state, 0 - y / alpha = lam y)
        if (closed_ok) { yt[i] = yc; k6[i] = lam * yc; }
    }
    if constexpr (!FIXED) {
        const double he0 = h * E0, he2 = h * E2, he3 = h * E3, he4 = h * E4, he5 = h * E5, he6 = h * E6;
        double e[NEQ], tl[NEQ];
#pragma unroll
        for (int i = 0; i < NEQ; ++i) {
            e[i] = lc(he0 * k0[i], he2, k2[i], he3, k3[i], he4, k4[i], he5, k5[i], he6, k6[i]);
            tl[i] = __builtin_fma(A.rtol, vmax_abs(y[i], yt[i]), A.atol);
        }
        jump = 0.0;
        double jd[NEQ];
#pragma unroll
        for (int i = 0; i < NEQ; ++i) jd[i] = k0[i] - k1[i];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = Model::NORAIN_LINEAR_COMP[k];
            e[i] = closed_ok ? e_lin[k] : e[i];
            jd[i] = closed_ok ? j_lin[k] : jd[i];
        }
        const double a = tl[0] * tl[1], b = tl[2] * tl[3];
        const double ab = a * b, all = ab * tl[4], bt4 = b * tl[4], at4 = a * tl[4];
        const double others[NEQ] = {tl[1] * bt4, tl[0] * bt4, tl[3] * at4, tl[2] * at4, ab};
        double num = 0.0;
#pragma unroll
        for (int i = 0; i < NEQ; ++i) num = vmax_abs2(num, e[i] * others[i]);
        err = num * fast_rcp(all);
#pragma unroll
        for (int i = 0; i < NEQ; ++i) jump = vmax_abs2(jump, jd[i]);
    }
}

// attempt_per_lane_path for waves that may hold saturated lanes (the HYBRID instances of the queue kernel).  wave_sat
// (wave-uniform, the caller's; false at the start of an item): some lane of this wave needed the general RHS in its
// previous attempt.  While it is clear the attempt is attempt_per_lane_path's, unchanged -- the headline workload never
// sets it.  Once set, the wave asks before every attempt whether a lane's static store overflows at its CURRENT state
// (or cannot take an unsaturated form at all): if so the whole wave runs rk_attempt_hybrid, one block instead of up to
// three; if not, the per-lane blocks again (a lane may still overflow inside the step: recomputed as before).
template <class Model, int MATH, bool FIXED>
__device__ __forceinline__ void attempt_per_lane_path_hybrid(const typename Model::Local& P, const typename Model::Forc& FV,
                                                             const Rk45Args& A, const double (&y)[Model::N_EQ], double h, bool& have_k0,
                                                             double (&k0)[Model::N_EQ], double (&k1)[Model::N_EQ], double (&k2)[Model::N_EQ],
                                                             double (&k3)[Model::N_EQ], double (&k4)[Model::N_EQ], double (&k5)[Model::N_EQ],
                                                             double (&k6)[Model::N_EQ], double (&yt)[Model::N_EQ], double& err, double& jump,
                                                             bool& wave_sat)
{
    if (wave_sat) {
        const bool needs_general = !Model::unsat_eligible(P, y) || Model::saturated_at(P, FV, y);
        if (__builtin_amdgcn_ballot_w64(needs_general) != 0ull) {
            rk_attempt_hybrid<Model, MATH, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
            have_k0 = true;
            return;   // (wave_sat stays set: the next attempt asks again)
        }
    }
    bool stagewise = true;
    if (Model::norainpath_ok(P, FV, y)) {
        const bool had_k0 = have_k0;
        const bool broke = rk_attempt<Model, MATH, PATH_NORAIN, FIXED, true>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        stagewise = broke;
        have_k0 = broke ? had_k0 : true;
    }
    else if (Model::wetupath_ok(P, FV, y)) {
        const bool had_k0 = have_k0;
        const bool broke = rk_attempt<Model, MATH, PATH_WETU, FIXED, true>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        stagewise = broke;
        have_k0 = broke ? had_k0 : true;
    }
    bool ran_general = false;
    if (stagewise) {
        bool general = true;
        if constexpr (Model::HAS_FASTPATH) {
            if (__builtin_amdgcn_ballot_w64(!Model::fastpath_ok(y)) == 0ull) {
                const bool broke = rk_attempt<Model, MATH, PATH_DRY, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
                have_k0 = true;
                general = __builtin_amdgcn_ballot_w64(broke) != 0ull;
            }
        }
        if (general) {
            rk_attempt<Model, MATH, PATH_GENERAL, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
            ran_general = true;
        }
    }
    wave_sat = __builtin_amdgcn_ballot_w64(ran_general) != 0ull;
    have_k0 = true;
}

// ---- which RHS form integrated an attempt (diagnostic instances only: hlm_device_job.d_path_counts) -----------------
// Lane-attempts by the form that produced the attempt's result -- [0] unsaturated with a dry surface (PATH_NORAIN, the
// linear stores in closed form), [1] unsaturated with a wet surface (PATH_WETU, closed form), [2] dry surface stage by
// stage (PATH_DRY), [3] the general RHS (PATH_GENERAL: the static store overflows, the whole cascade runs) -- and
// [4] lane-attempts computed TWICE: started on an unsaturated form, left its regime inside the step, recomputed stage by
// stage (they are counted under [2] / [3] as well).  [5] wave-attempts (one per wave and trip of the attempt loop),
// [6] wave-attempts that executed more than one form (lanes of the wave on different forms, or a recomputation).
constexpr int PATH_COUNTERS = 7;
struct LanePathCount {
    uint32_t n[PATH_COUNTERS] = {};
    __device__ __forceinline__ void lane(int k) { ++n[k]; }
    // called by every lane that took part in the attempt; the wave's first active lane keeps the wave-level counts
    __device__ __forceinline__ void wave(bool several)
    {
        const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
        if ((int)__builtin_ctzll(act) == (int)threadIdx.x) { ++n[5]; n[6] += several ? 1u : 0u; }
    }
    __device__ __forceinline__ void flush(unsigned long long* out)
    {
        if (!out) return;
#pragma unroll
        for (int k = 0; k < PATH_COUNTERS; ++k)
            if (n[k]) (void)__hip_atomic_fetch_add(&out[k], (unsigned long long)n[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// attempt_per_lane_path with the counts: THE SAME sequence of calls (the results are bit-identical, tested), kept as a
// copy because the plain function is part of the headline kernel, whose register allocation moves with anything that
// touches its text -- even template parameters that compile to nothing (measured: 36 -> 73 spilled VGPRs).
template <class Model, int MATH, bool FIXED>
__device__ __forceinline__ void attempt_per_lane_path_counted(const typename Model::Local& P, const typename Model::Forc& FV,
                                                              const Rk45Args& A, const double (&y)[Model::N_EQ], double h, bool& have_k0,
                                                              double (&k0)[Model::N_EQ], double (&k1)[Model::N_EQ], double (&k2)[Model::N_EQ],
                                                              double (&k3)[Model::N_EQ], double (&k4)[Model::N_EQ], double (&k5)[Model::N_EQ],
                                                              double (&k6)[Model::N_EQ], double (&yt)[Model::N_EQ], double& err, double& jump,
                                                              LanePathCount& pc)
{
    const bool nr = Model::norainpath_ok(P, FV, y), wu = !nr && Model::wetupath_ok(P, FV, y);
    // how many of the four blocks below this WAVE executes
    int forms = (__builtin_amdgcn_ballot_w64(nr) != 0ull) + (__builtin_amdgcn_ballot_w64(wu) != 0ull);
    bool tried_unsat = false;
    bool stagewise = true;
    if (nr) {
        const bool had_k0 = have_k0;
        const bool broke = rk_attempt<Model, MATH, PATH_NORAIN, FIXED, true>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        stagewise = broke;
        have_k0 = broke ? had_k0 : true;
        tried_unsat = true;
        if (!broke) pc.lane(0);
    }
    else if (wu) {
        const bool had_k0 = have_k0;
        const bool broke = rk_attempt<Model, MATH, PATH_WETU, FIXED, true>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        stagewise = broke;
        have_k0 = broke ? had_k0 : true;
        tried_unsat = true;
        if (!broke) pc.lane(1);
    }
    int stage_forms = 0;  // stage-wise blocks this lane's branch executed (uniform among the lanes that took it)
    if (stagewise) {
        bool general = true;
        if constexpr (Model::HAS_FASTPATH) {
            if (__builtin_amdgcn_ballot_w64(!Model::fastpath_ok(y)) == 0ull) {
                const bool broke = rk_attempt<Model, MATH, PATH_DRY, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
                have_k0 = true;
                general = __builtin_amdgcn_ballot_w64(broke) != 0ull;
                ++stage_forms;
                if (!general) pc.lane(2);
            }
        }
        if (general) {
            rk_attempt<Model, MATH, PATH_GENERAL, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
            ++stage_forms;
            pc.lane(3);
        }
        if (tried_unsat) pc.lane(4);
    }
    forms += __builtin_amdgcn_ballot_w64(stage_forms >= 2) != 0ull ? 2 : (__builtin_amdgcn_ballot_w64(stage_forms >= 1) != 0ull ? 1 : 0);
    pc.wave(forms > 1);
    have_k0 = true;
}

// NEED_K: the caller reads the stage slopes of the accepted step afterwards (dense output); else the no-rain path
// may take its linear components in closed form (FAST only: attempt_per_lane_path).
template <class Model, int MATH, bool FIXED, bool NEED_K = true>
__device__ __forceinline__ void attempt_on_cheapest_path(const typename Model::Local& P, const typename Model::Forc& FV,
                                                         const Rk45Args& A, const double (&y)[Model::N_EQ], double h, bool& have_k0,
                                                         double (&k0)[Model::N_EQ], double (&k1)[Model::N_EQ], double (&k2)[Model::N_EQ],
                                                         double (&k3)[Model::N_EQ], double (&k4)[Model::N_EQ], double (&k5)[Model::N_EQ],
                                                         double (&k6)[Model::N_EQ], double (&yt)[Model::N_EQ], double& err, double& jump)
{
    if constexpr (!NEED_K && MATH == MATH_FAST && Model::HAS_NORAINPATH && Model::NORAIN_LINEAR > 0) {
        attempt_per_lane_path<Model, MATH, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        return;
    }
    bool general = true;
    if constexpr (Model::HAS_NORAINPATH) {
        // between storms (no rain in the held sample, dry surface): the cascade below the static store is idle
        if (__builtin_amdgcn_ballot_w64(!Model::norainpath_ok(P, FV, y)) == 0ull) {
            const bool had_k0 = have_k0;
            const bool broke = rk_attempt<Model, MATH, PATH_NORAIN, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
            general = __builtin_amdgcn_ballot_w64(broke) != 0ull;
            // Unlike the dry path's, this path's precondition is not decided by y alone: "static
            // store below capacity" is only discovered by the RHS evaluations themselves, the one
            // for k0 included.  A k0 evaluated in an attempt that falls back is therefore not kept.
            have_k0 = general ? had_k0 : true;
        }
    }
    if constexpr (Model::HAS_FASTPATH) {
        if (general && __builtin_amdgcn_ballot_w64(!Model::fastpath_ok(y)) == 0ull) {
            const bool broke = rk_attempt<Model, MATH, PATH_DRY, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
            have_k0 = true;  // k0 of a qualifying state is exact
            general = __builtin_amdgcn_ballot_w64(broke) != 0ull;
        }
    }
    if (general) rk_attempt<Model, MATH, PATH_GENERAL, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
    have_k0 = true;
}

// ---- dense output ---------------------------------------------------------------------------
// The host-facing dense layout is [sys][query][comp]: a lane's rows are contiguous in memory
// (40 B per query) but the 64 lanes of a wave are nq*40 B apart, so a row store touches 64
// different cache lines per instruction.  Measured on MI355X (1 M hillslopes, 1-minute queries,
// 57.6 GB of rows; profiles/r01_dense_pmc.json): 26.6 ms, WRITE_SIZE 91.8 GB (partial-line
// evictions), 54 % of wave cycles waiting.  Parking rows in LDS and draining them cooperatively
// with coalesced <= 240-byte stores lowered WRITE_SIZE to 67.0 GB but cost +50 % VALU for the
// drain loop (27.7 ms) and was removed again; the tile-major layout below (rows of a wave
// contiguous) runs in 19.4 ms.  The streamed pipeline is PCIe-bound at 55 GB/s either way.
// Query times: the next query time gates the row loop; instead of a dependent global load
// per row the wave keeps a sliding window of QWIN query times in LDS (lanes of a wave
// advance through the queries together; a lane outside the window falls back to the global
// load).  Worth 2 % on the dense kernel (27.2 -> 26.6 ms): the stores, not this load, set
// its pace.
constexpr int DENSE_QWIN = 256;

// RESUMABLE: query windows / t_stop / forcing time chunks / checkpoints.  The plain variant
// (one uninterrupted launch, all forcings resident) carries none of that state: keeping it
// out of the default kernel is worth 3.5 % (199 -> 192 ms on the C3 pass).
// SEL (dense launches only): output selection, hlm_device_job.dense_ncomp -- a dense row holds the A.dense_k
// components c with A.dense_pos[c] >= 0, at that position (the reference's output.states, data/config.yaml:44-52).
template <class Model, int MATH, bool DENSE, bool FIXED, bool RESUMABLE, bool SEL = false>
__global__ void __launch_bounds__(WAVE, (MATH == MATH_FAST && !DENSE) ? HLM_MIN_WAVES_PER_SIMD_LEAN : HLM_MIN_WAVES_PER_SIMD)
    rk45_multi_kernel(const Rk45Args A)
{
    using namespace dp;
    static_assert(DENSE || !SEL, "output selection concerns the dense rows");
    constexpr int NEQ = Model::N_EQ, RESUME_F64 = resume_f64(NEQ);
    [[maybe_unused]] const int K = SEL ? A.dense_k : NEQ;  // components per dense row
    // XCD-aware block -> hillslope-chunk map: workgroups are dealt round-robin over the
    // 8 XCDs, so give each XCD one contiguous range of hillslopes (neighbouring
    // hillslopes share forcing rows -> they hit in that XCD's L2).  Bijective for any
    // grid size (cdna guide T1).
    const unsigned nblk = gridDim.x, orig = blockIdx.x;
    const unsigned xcd = orig & 7u, qd = nblk >> 3, rm = nblk & 7u;
    const unsigned blk = (xcd < rm ? xcd * (qd + 1u) : rm * (qd + 1u) + (xcd - rm) * qd) + (orig >> 3);
    const int sys = (int)(blk * WAVE + threadIdx.x);
    if (sys >= A.ns) return;

    const typename Model::Local P = Model::template load<MATH>(A.sp, sys);

    double y[NEQ];
#pragma unroll
    for (int i = 0; i < NEQ; ++i) y[i] = A.y0[(size_t)sys * NEQ + i];

    HeldForcing<Model, RESUMABLE> held;  // zero-order-hold samples of this lane (rk45_kernel.cu:85-106)
    held.init(A, sys);

    double t = A.t0;
    double h = FIXED ? A.fixed_step : A.h0;
    const double tf = A.tf;
    int next_q = 0;
    int reject_count = 0;
    int status = HLM_SYS_OK;
    uint32_t n_attempt = 0, n_accept = 0;
    double* dense_row = nullptr;
    [[maybe_unused]] size_t q_stride = NEQ;
    double tq_next = 0.0;
    __shared__ double lds_query[DENSE ? DENSE_QWIN : 1];
    [[maybe_unused]] int qwin_base = -(1 << 30);  // wave-uniform: first query index held in lds_query
    using lds_cdouble_ptr = const __attribute__((address_space(3))) double*;
    [[maybe_unused]] auto query_at = [&](int q) -> double {  // q < nq
        // two separate loads (ds_read / global_load): a select between an LDS and a global
        // POINTER would become a flat access
        const unsigned idx = (unsigned)(q - qwin_base);
        double v;
        if (idx < (unsigned)DENSE_QWIN) v = ((lds_cdouble_ptr)lds_query)[idx];
        else v = A.query[q];
        return v;
    };
    [[maybe_unused]] const int lane = (int)threadIdx.x;
    [[maybe_unused]] int qb = 0, qe = 0;
    if constexpr (DENSE) {
        // Query window [qb, qe) of this launch (the whole range unless the host streams the
        // dense output in chunks, hlm_rk45_integrate_streamed).  A lane that has to write a
        // query >= qe stops WITHOUT committing the step that contains it, saves
        // {y, t, h, counters} and re-executes that same attempt in the next launch: the
        // sequence of committed steps -- and every output bit -- equals the one-shot run.
        qb = A.q_begin;
        qe = A.q_end;
        // row(q) = dense_row + q * q_stride: the host-facing layout [sys][q][comp] (the tile-major layout is the ring
        // kernel's, hlm_rk45_dense_kernel.h)
        q_stride = SEL ? K : NEQ;
        dense_row = A.dense + ((size_t)sys * (size_t)(qe - qb) - (size_t)qb) * (SEL ? K : NEQ);
    }
    // Integrator state carried between launches (query windows, time chunks of streamed
    // forcings, checkpoints): a paused lane continues exactly where it stopped.
    int* const resume_i =
        RESUMABLE && A.resume ? reinterpret_cast<int*>(A.resume + (size_t)RESUME_F64 * A.ns) : nullptr;
    if (resume_i && A.resume_in) {
        const int st = resume_i[(size_t)4 * A.ns + sys];
        if (st != SYS_PAUSED) {
            // Finished or flagged in an earlier launch: its outputs were written then.  With a
            // per-window dense buffer (query windows) the slots of THIS window were never reached
            // and are 0.  With the whole query range in one buffer (time chunks / checkpoints)
            // the launch that finished the lane already zero-filled [next_q, nq) and the rows
            // before that hold its results: nothing may be touched.
            if constexpr (DENSE) {
                if (qb != 0 || qe != A.nq) {
                    for (int q = qb; q < qe; ++q) {
#pragma unroll
                        for (int c = 0; c < NEQ; ++c)
                            if (!SEL || c < K) dense_row[(size_t)q * q_stride + c] = 0.0;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NEQ; ++i) y[i] = A.resume[(size_t)i * A.ns + sys];
        t = A.resume[(size_t)NEQ * A.ns + sys];
        h = A.resume[(size_t)(NEQ + 1) * A.ns + sys];
        reject_count = resume_i[(size_t)0 * A.ns + sys];
        n_attempt = (uint32_t)resume_i[(size_t)1 * A.ns + sys];
        n_accept = (uint32_t)resume_i[(size_t)2 * A.ns + sys];
        next_q = resume_i[(size_t)3 * A.ns + sys];
    }
    if constexpr (DENSE) {
        // queries at or before the current time are never produced (rk45_kernel.cu:141):
        // their slot is 0.  (No-op on a resumed lane: earlier steps consumed them.)
        while (next_q < A.nq && A.query[next_q] <= t) {
            if (next_q >= qe) { status = SYS_PAUSED; break; }
#pragma unroll
            for (int c = 0; c < NEQ; ++c)
                if (!SEL || c < K) dense_row[(size_t)next_q * q_stride + c] = 0.0;
            ++next_q;
        }
        tq_next = next_q < A.nq ? A.query[next_q] : __builtin_inf();
    }

    // k0 = rhs(t, y, F) is recomputed by the reference on every attempt (no FSAL,
    // rk45_kernel.cu:114).  It is a pure function of (t, y, F): after a reject or a
    // slope-jump retry none of them changed, and after an accepted step k6 was evaluated at
    // exactly (t + 1.0*h, y_new, F), so whenever the held forcing samples did not change
    // the previous value IS the reference's k0, bit for bit.  have_k0 tracks that.
    bool have_k0 = false;
    double k0[NEQ];
    typename Model::Forc FV = Model::prep_forcing(P, held.F);

    // t_stop (time chunks / checkpoints): a lane whose t has reached it pauses BEFORE its next
    // attempt, so the continuation in the next launch repeats nothing and changes nothing.
    const double t_end = RESUMABLE ? fmin(tf, A.t_stop) : tf;
    // With dense rows stored inside the loop, the prologue's loads are waited for here rather than at
    // their first use in the loop body: gfx950 counts loads and stores in one counter (vmcnt), so such
    // a wait would hold every attempt until the previous step's row stores are acknowledged.
    if constexpr (DENSE) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    while (t < t_end && status == HLM_SYS_OK && n_attempt < A.cap) {
        if constexpr (DENSE) {
            // slide the query-time window when the wave's first active lane nears its end
            const int qf = __builtin_amdgcn_readfirstlane(next_q);
            if (qf < A.nq && (unsigned)(qf - qwin_base) >= (unsigned)(DENSE_QWIN - 16)) {
                const int nb = qf > 16 ? qf - 16 : 0;
                const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
                const int nact = __builtin_popcountll(act);
                const int rank = __builtin_popcountll(act & ((1ull << lane) - 1ull));
                for (int i = rank; i < DENSE_QWIN; i += nact)
                    lds_query[i] = (nb + i) < A.nq ? A.query[nb + i] : __builtin_inf();
                qwin_base = nb;
            }
        }
        if (t + h > tf) h = tf - t;  // rk45_kernel.cu:54

        if (held.refresh(A, P, t, FV)) have_k0 = false;

        ++n_attempt;
        // ---- 7 stages + the two norms, on the cheapest RHS path the wave qualifies for -------------
        double k1[NEQ], k2[NEQ], k3[NEQ], k4[NEQ], k5[NEQ], k6[NEQ], yt[NEQ], yn[NEQ];
        [[maybe_unused]] double err = 0.0, jump = 0.0;
        attempt_on_cheapest_path<Model, MATH, FIXED, DENSE>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        // 5th-order solution (rk45_step_dense.cuh:111-117).  b[6] = 0, so it is the same
        // chain of operations as the stage-6 state: identical bits, reuse it.
#pragma unroll
        for (int i = 0; i < NEQ; ++i) yn[i] = yt[i];

        [[maybe_unused]] double fac = 1.0;
        if constexpr (!FIXED) {
            // step-size factor shared by the accept and the reject branch
            // (rk45_kernel.cu:151,156)
            fac = A.safety * inv_fifth_root<MATH>(err + 1e-16);

            if (err <= 1.0) {
                reject_count = 0;
                if (jump > SLOPE_JUMP_THRESH) {  // slope-jump detector (rk45_kernel.cu:132-136)
                    h = fmax(h * 0.5, A.h0 * MIN_STEP_FRACTION);
                    continue;
                }
            } else {
                // rejected (rk45_kernel.cu:153-163)
                ++reject_count;
                double f = fmin(fac, 1.0);
                f = fmin(fmax(f, A.min_scale), A.max_scale);
                h *= f;
                if (reject_count > 5 || h < (tf - A.t0) * MIN_STEP_FRACTION) status = HLM_SYS_STIFF;
                continue;
            }
        }

        // ---- accepted step (always, in fixed-step mode) ----
        const double t1 = t + h;
        if constexpr (DENSE) {
            if (tq_next <= t1) {
                // the interpolant's coefficients once per step (the reference recomputes Q = P^T k for every
                // query, rk45_step_dense.cuh:222-231), in Hairer's form: see dense_coefficients
                double D1[NEQ], D2[NEQ], D3[NEQ], D4[NEQ];
                dense_coefficients<NEQ>(y, yn, h, k0, k2, k3, k4, k5, k6, D1, D2, D3, D4);
                bool pause = false;
                [[maybe_unused]] const double inv_h = MATH == MATH_FAST ? fast_rcp(h) : 0.0;
                do {
                    if (next_q >= qe) { pause = true; break; }
                    double th;  // rk45_kernel.cu:142
                    if constexpr (MATH == MATH_FAST) th = (tq_next - t) * inv_h; else th = (tq_next - t) / h;
                    const double th1 = 1.0 - th;
                    double* o = dense_row + (size_t)next_q * q_stride;
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) {
                        if (SEL && A.dense_pos[i] < 0) continue;  // (wave-uniform) component not selected
                        o[SEL ? A.dense_pos[i] : i] = dense_value(th, th1, y[i], D1[i], D2[i], D3[i], D4[i]);  // rk45_step_dense.cuh:235-243
                    }
                    ++next_q;
                    tq_next = next_q < A.nq ? query_at(next_q) : __builtin_inf();
                } while (tq_next <= t1);
                if (pause) {
                    // the rest of this step's queries belong to the next window: leave the
                    // step uncommitted (y, t, h unchanged) and redo it there
                    status = SYS_PAUSED;
                    --n_attempt;
                    break;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NEQ; ++i) {
            y[i] = yn[i];
            k0[i] = k6[i];  // = rhs(t1, y_new, F): next attempt's k0 unless the forcing moves
        }
        t = t1;
        ++n_accept;
        if constexpr (FIXED) h = A.fixed_step;
        else h *= vclamp_uniform(fac, A.min_scale, A.max_scale);  // rk45_kernel.cu:151-152
    }

    if (status == HLM_SYS_OK && t < t_end && n_attempt >= A.cap) status = HLM_SYS_CAPPED;  // attempt cap
    if (RESUMABLE && status == HLM_SYS_OK && t < tf) status = SYS_PAUSED;  // stopped by t_stop
    if (resume_i) {
        resume_i[(size_t)4 * A.ns + sys] = status;
        if (status == SYS_PAUSED) {
#pragma unroll
            for (int i = 0; i < NEQ; ++i) A.resume[(size_t)i * A.ns + sys] = y[i];
            A.resume[(size_t)NEQ * A.ns + sys] = t;
            A.resume[(size_t)(NEQ + 1) * A.ns + sys] = h;
            resume_i[(size_t)0 * A.ns + sys] = reject_count;
            resume_i[(size_t)1 * A.ns + sys] = (int)n_attempt;
            resume_i[(size_t)2 * A.ns + sys] = (int)n_accept;
            resume_i[(size_t)3 * A.ns + sys] = next_q;
        }
    }
    if (status == SYS_PAUSED) return;  // outputs are written by the launch that finishes the lane

    if (A.n_attempt) A.n_attempt[sys] = n_attempt;
    if (A.n_accept) A.n_accept[sys] = n_accept;
    const bool flagged = status != HLM_SYS_OK && t < tf;  // rk45_kernel.cu:167-170
    A.stiff[sys] = flagged ? status : HLM_SYS_OK;
#pragma unroll
    for (int i = 0; i < NEQ; ++i) A.y_final[(size_t)sys * NEQ + i] = flagged ? 0.0 : y[i];
    if (A.flag_state && flagged) record_flag_state<NEQ>(A.flag_state, sys, y, t);
    if constexpr (DENSE) {
        // queries this lane never reaches (flagged, or later than tf): zero rows
        const int q_stop = A.nq < qe ? A.nq : qe;
        for (; next_q < q_stop; ++next_q) {
#pragma unroll
            for (int c = 0; c < NEQ; ++c)
                if (!SEL || c < K) dense_row[(size_t)next_q * q_stride + c] = 0.0;
        }
    }
}

// ---- ordered compaction (the Radau pass's list of flagged systems: launch_radau) ------------------------------
// index[0 .. count) = the systems i with status[i] == want, in ascending order (neighbouring hillslopes
// share a forcing cell: keeping the order makes the lanes of a Radau wave take similar numbers of attempts).
// Three small launches: per-block counts, a scan of the block counts by one block, the scatter.
// status = stiff[] (want = HLM_SYS_STIFF) or the flag-time bins of radau_flag_bin_kernel.
constexpr int COMPACT_BLOCK = 1024;

__global__ void __launch_bounds__(COMPACT_BLOCK) compact_count_kernel(const int* __restrict__ status, int want, int ns, int* __restrict__ block_count)
{
    __shared__ int wave_count[COMPACT_BLOCK / WAVE];
    const int i = (int)(blockIdx.x * COMPACT_BLOCK + threadIdx.x);
    const bool alive = i < ns && status[i] == want;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
    if ((threadIdx.x & (WAVE - 1)) == 0) wave_count[threadIdx.x / WAVE] = __builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < COMPACT_BLOCK / WAVE; ++w) c += wave_count[w];
        block_count[blockIdx.x] = c;
    }
}

// exclusive scan of block_count[0 .. nb) in place (nb is ~1000 for a million systems: one block, serial
// chunks per thread), total -> *count
// base (nullable, in/out): the list already holds *base entries -- the offsets start there and *base becomes the new
// total, so that several passes with different `want` append to one list (the Radau pass orders the flagged systems
// by the time of their flag that way: radau_flag_bin_kernel)
// total_out (nullable): the new total once more, where the caller wants it kept (the Radau pass's entries per time piece)
__global__ void __launch_bounds__(COMPACT_BLOCK) compact_scan_kernel(int* __restrict__ block_count, int nb, int* __restrict__ count,
                                                                     int* __restrict__ base = nullptr, int* __restrict__ total_out = nullptr)
{
    __shared__ int part[COMPACT_BLOCK];
    const int per = (nb + COMPACT_BLOCK - 1) / COMPACT_BLOCK;
    const int lo = (int)threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += block_count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = base ? *base : 0;
        for (int i = 0; i < COMPACT_BLOCK; ++i) { const int v = part[i]; part[i] = run; run += v; }
        *count = run;
        if (base) *base = run;
        if (total_out) *total_out = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int v = block_count[i]; block_count[i] = run; run += v; }
}

__global__ void __launch_bounds__(COMPACT_BLOCK) compact_scatter_kernel(const int* __restrict__ status, int want, int ns,
                                                                        const int* __restrict__ block_offset, int32_t* __restrict__ index)
{
    __shared__ int wave_offset[COMPACT_BLOCK / WAVE];
    const int i = (int)(blockIdx.x * COMPACT_BLOCK + threadIdx.x);
    const bool alive = i < ns && status[i] == want;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
    const int lane = (int)(threadIdx.x & (WAVE - 1)), w = (int)(threadIdx.x / WAVE);
    if (lane == 0) wave_offset[w] = __builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = block_offset[blockIdx.x];
#pragma unroll
        for (int k = 0; k < COMPACT_BLOCK / WAVE; ++k) { const int v = wave_offset[k]; wave_offset[k] = run; run += v; }
    }
    __syncthreads();
    if (alive) index[wave_offset[w] + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
}

// bin[i] = which of the nbins equal parts of [t0, tf] system i was flagged in (earliest = 0), -1 when it was not
// flagged: the key of the Radau pass's work order in "continue" mode (a system flagged early has most left to do)
__global__ void __launch_bounds__(COMPACT_BLOCK) radau_flag_bin_kernel(const int* __restrict__ stiff, const double* __restrict__ flag_state,
                                                                       int neq, int ns, double t0, double tf, int nbins, int* __restrict__ bin)
{
    const int i = (int)(blockIdx.x * COMPACT_BLOCK + threadIdx.x);
    if (i >= ns) return;
    int b = -1;
    if (stiff[i] == HLM_SYS_STIFF) {
        const double x = (flag_state[(size_t)i * (neq + 1) + neq] - t0) / (tf - t0) * (double)nbins;
        b = x >= 0.0 ? (x < (double)nbins ? (int)x : nbins - 1) : 0;  // (NaN -> 0)
    }
    bin[i] = b;
}

// ---- self-test of the FAST-mode elementary functions (hlm_selftest_math) ---------------------
// which: 0 = x^(2/3) (pow_two_thirds<FAST>), 1 = x^(-1/5) (inv_fifth_root<FAST>), 2 = 1/x (fast_rcp),
//        3 = x^(2/3) STRICT (hlm_strict_pow.h), 4 = x^(-1/5) STRICT, 5 = div_by_invariant, 6 = IEEE n/d
//        (5 and 6 read the divisors from in[n .. 2n)), 7 / 8 = the seed's residual e inside 3 / 4's power (for x > 0:
//        what the f32 log2 / exp2 of this hardware leave for the series), 9 = (1/x)^(1/4) STRICT (the Radau pass's)
__global__ void selftest_math_kernel(int which, const double* __restrict__ in, double* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    double r;
    switch (which) {
        case 0: r = pow_two_thirds<MATH_FAST>(x); break;
        case 1: r = inv_fifth_root<MATH_FAST>(x); break;
        case 2: r = fast_rcp(x); break;
        case 3: r = pow_two_thirds<MATH_STRICT>(x); break;
        case 4: r = inv_fifth_root<MATH_STRICT>(x); break;
        case 5: {  // n / d with the divisor's correctly rounded reciprocal (in[i] = n, in[i + n] = d)
            const double d = in[i + n];
            r = div_by_invariant(x, d, 1.0 / d);
            break;
        }
        case 6: r = x / in[i + n]; break;  // IEEE division, the reference for case 5
        case 7: (void)strict_pow_two_thirds_core(x, &r); break;
        case 8: (void)strict_pow_fifth_core(x, &r); break;
        default: r = strict_pow_fourth(1.0 / x); break;  // 9
    }
    out[i] = r;
}

// ---- single RHS evaluations (hlm_selftest_rhs): the reference's debugRHS kernel, main.cpp:219-246 ----
// y: [n][5] states, sp: [n] SpatialParams, F: [n][2] forcing values (as the float samples the
// gather would deliver), out: [n][5] = Model::rhs.  fast_path = 1 evaluates rhs_fast instead
// (valid where Model::fastpath_ok(y) holds), fast_path = 2 rhs_norain (valid where
// Model::norainpath_ok holds); both store their "left the regime" flag in broke[n].
template <class Model, int MATH>
__global__ void selftest_rhs_kernel(const double* __restrict__ y, const void* sp, const float* __restrict__ F,
                                    double* __restrict__ out, int* __restrict__ broke, int fast_path, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NEQ = Model::N_EQ, NFA = Model::NF > 0 ? Model::NF : 1;
    const typename Model::Local P = Model::template load<MATH>(sp, i);
    double held[NFA];
#pragma unroll
    for (int j = 0; j < NFA; ++j) held[j] = j < 2 ? (double)F[2 * i + j] : 0.0;  // d_F holds two samples per evaluation
    const typename Model::Forc FV = Model::prep_forcing(P, held);
    double yy[NEQ], d[NEQ];
#pragma unroll
    for (int c = 0; c < NEQ; ++c) yy[c] = y[(size_t)i * NEQ + c];
    bool b = false;
    if (fast_path == 3) {
        if constexpr (Model::HAS_NORAINPATH && Model::NORAIN_LINEAR > 0 && MATH == MATH_FAST) b = Model::template rhs_wetu<MATH>(P, FV, yy, d);
        else Model::template rhs<MATH>(P, FV, yy, d);
    }
    else if (fast_path == 2) b = Model::template rhs_norain<MATH>(P, FV, yy, d);
    else if (fast_path) b = Model::template rhs_fast<MATH>(P, FV, yy, d);
    else Model::template rhs<MATH>(P, FV, yy, d);
#pragma unroll
    for (int c = 0; c < NEQ; ++c) out[(size_t)i * NEQ + c] = d[c];
    if (broke) broke[i] = b ? 1 : 0;
}

// ---- single attempts (hlm_selftest_attempt): one rk45_step (rk45_step_dense.cuh:33-145) per (state, parameters,
// forcing, h) on a chosen RHS path, from a cold start (k0 evaluated inside) -- the unit under the integrator's loop.
// path = HLM_ATTEMPT_SELECT_NODENSE (-1): what a launch WITHOUT dense output executes per attempt
//        (attempt_on_cheapest_path<.., NEED_K = false>: per-lane choice of the unsaturated path in closed form, dry /
//        general otherwise -- the headline kernel's attempt);
//        HLM_ATTEMPT_SELECT_DENSE (-2): what the dense kernels execute (NEED_K = true, wave-uniform choice, stage arithmetic);
//        0..3 = PATH_GENERAL / PATH_DRY / PATH_NORAIN / PATH_WETU forced for every lane, `closed` = the linear stores in
//        closed form (paths 2 and 3, FAST).  A forced path is valid where its precondition holds and `broke` is 0.
// out: ynew[n][NEQ] = the 5th-order solution, en[n][2] = {scaled error norm, slope-jump norm |k0 - k1|_inf}.
template <class Model, int MATH>
__global__ void __launch_bounds__(WAVE) selftest_attempt_kernel(const double* __restrict__ y, const void* sp, const float* __restrict__ F,
                                                                const double* __restrict__ hs, double rtol, double atol, int path, int closed,
                                                                double* __restrict__ ynew, double* __restrict__ en, int* __restrict__ broke, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NEQ = Model::N_EQ, NFA = Model::NF > 0 ? Model::NF : 1;
    const typename Model::Local P = Model::template load<MATH>(sp, i);
    double held[NFA];
#pragma unroll
    for (int j = 0; j < NFA; ++j) held[j] = j < 2 ? (double)F[2 * i + j] : 0.0;
    const typename Model::Forc FV = Model::prep_forcing(P, held);
    Rk45Args A{};
    A.rtol = rtol;
    A.atol = atol;
    double yy[NEQ], k0[NEQ], k1[NEQ], k2[NEQ], k3[NEQ], k4[NEQ], k5[NEQ], k6[NEQ], yt[NEQ];
#pragma unroll
    for (int c = 0; c < NEQ; ++c) yy[c] = y[(size_t)i * NEQ + c];
    const double h = hs[i];
    double err = 0.0, jump = 0.0;
    bool have_k0 = false, b = false;
    constexpr bool CHEAP = Model::HAS_NORAINPATH && Model::HAS_FASTPATH;
    constexpr bool CLOSABLE = CHEAP && Model::NORAIN_LINEAR > 0 && MATH == MATH_FAST;
    if (path == -1) attempt_on_cheapest_path<Model, MATH, false, false>(P, FV, A, yy, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
    else if (path == -2) attempt_on_cheapest_path<Model, MATH, false, true>(P, FV, A, yy, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
    else if (path == PATH_DRY && CHEAP) {
        if constexpr (CHEAP) b = rk_attempt<Model, MATH, PATH_DRY, false>(P, FV, A, yy, h, false, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
    } else if (path == PATH_NORAIN && CHEAP) {
        if constexpr (CLOSABLE) {
            if (closed) b = rk_attempt<Model, MATH, PATH_NORAIN, false, true>(P, FV, A, yy, h, false, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        }
        if constexpr (CHEAP) {
            if (!(closed && CLOSABLE)) b = rk_attempt<Model, MATH, PATH_NORAIN, false>(P, FV, A, yy, h, false, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        }
    } else if (path == PATH_WETU && CLOSABLE) {
        if constexpr (CLOSABLE) {
            if (closed) b = rk_attempt<Model, MATH, PATH_WETU, false, true>(P, FV, A, yy, h, false, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
            else b = rk_attempt<Model, MATH, PATH_WETU, false, false>(P, FV, A, yy, h, false, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
        }
    } else {
        rk_attempt<Model, MATH, PATH_GENERAL, false>(P, FV, A, yy, h, false, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
    }
#pragma unroll
    for (int c = 0; c < NEQ; ++c) ynew[(size_t)i * NEQ + c] = yt[c];
    en[(size_t)2 * i] = err;
    en[(size_t)2 * i + 1] = jump;
    if (broke) broke[i] = b ? 1 : 0;
}

// ---- tile-major window -> host order -------------------------------------------------------
// Transposes one query window from a tile-major device layout to the host-facing order
// dst[(sys*nq + q)*5 + c] (rk45_api.hpp:255-267).  dst is normally PINNED HOST memory mapped into the
// device: the kernel IS the device-to-host transfer (PCIe writes in 640-byte contiguous runs), running
// on its own stream beside the integration of the next window -- or a device staging buffer
// (hlm_retrieve_and_free).  One workgroup = 64 systems x TQ queries staged through LDS.
// src is [blk][W][NEQ][64] (the wave-synchronous kernel's layout).
constexpr int TILE_TQ = 16;

template <int NEQ>
__global__ void __launch_bounds__(256) dense_tile_to_host(const double* __restrict__ src, double* __restrict__ dst,
                                                          int ns, int nq, int q_begin, int W)
{
    // [ql][c][lane] with each 64-lane run padded to 65 so that the transposed reads below (consecutive threads =
    // consecutive components of one lane) do not all hit one LDS bank
    constexpr int RUN = WAVE + 1;
    __shared__ double tile[TILE_TQ * NEQ * RUN];  // ~41 KB
    const int blk = blockIdx.x, q0 = blockIdx.y * TILE_TQ;
    const int nql = (W - q0) < TILE_TQ ? (W - q0) : TILE_TQ;
    const int n_in = nql * WAVE * NEQ;
    const double* s = src + ((size_t)blk * (size_t)W + (size_t)q0) * (WAVE * NEQ);
    for (int i = threadIdx.x; i < n_in; i += 256) tile[(i / WAVE) * RUN + (i % WAVE)] = s[i];  // contiguous, coalesced
    __syncthreads();
    const int row_len = nql * NEQ;  // doubles each system contributes: contiguous in dst
    const int n_out = WAVE * row_len;
    for (int o = threadIdx.x; o < n_out; o += 256) {
        const int l = o / row_len, r = o - l * row_len;
        const int sys = blk * WAVE + l;
        if (sys < ns) {
            const int ql = r / NEQ, c = r - ql * NEQ;
            dst[((size_t)sys * (size_t)nq + (size_t)(q_begin + q0)) * NEQ + r] =
                tile[(ql * NEQ + c) * RUN + l];
        }
    }
}

}  // namespace hlm
