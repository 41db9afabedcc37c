// Radau-IIA fallback for the systems the RK45 loop flags as stiff.
//
// The reference launches a second kernel over the flagged systems (rk45_api.hpp:189-247 ->
// radau_kernel.cu:20-140 -> radau_step_dense.cuh) but never copies its results back, and the
// step it implements is not a consistent method (the dense interpolant, the error
// weights and the forcing indexing are wrong).  What is built here is the standard RADAU5
// formulation (Hairer & Wanner, Solving ODEs II, IV.8) on the reference's skeleton -- restart
// from (t0, y0), finite-difference Jacobian as approx_jacobian (radau_step_dense.cuh:13-31),
// <= 10 simplified-Newton iterations (:87), steps clipped to the forcing sample boundaries (an
// implicit step would otherwise hold one sample across many).  The CPU restatement the parity
// tests check it against lives under oracle/; both are validated against SciPy's Radau.
//
// Mapping to CDNA4: one lane per flagged system (the flag list is compacted first, in order -- the
// compact_* kernels of hlm_rk45_kernel.h -- so a wave holds 64 stiff systems that are neighbours).  The 15x15 Newton system is block-diagonalised with the RADAU5
// transformation (T, TI) into one real and one complex 5x5 system per attempt, small enough to
// factor and solve entirely in VGPRs: partial pivoting is done with per-lane selects (a row swap
// is 5 conditional exchanges), so no lane-indexed memory is needed at all.  A first version
// that kept the full 15x15 matrix in LDS ([element][lane], 140 KB per wave) took 100-160 us per
// attempt (LDS latency with one wave per CU, then instruction-cache misses once unrolled); this
// one is ~1/5 of the arithmetic and fits the instruction cache.  Cold path: a fraction of a
// percent of the systems in a production run.
#pragma once

#include "hlm_rk45_kernel.h"

namespace hlm {
namespace radau {

constexpr int NS = 3;
constexpr double SQRT6 = 2.449489742783178;
// Lambda = TI A^-1 T = [[g,0,0],[0,a,b],[0,-b,a]] (Hairer & Wanner, Solving ODEs II, IV.8)
constexpr double MU_REAL = 3.637834252744496;  // g
constexpr double MU_CRE = 2.6810828736277523;  // a
constexpr double MU_CIM = 3.050430199247411;   // b (the complex eigenvalue used is a - ib)
constexpr double T[NS][NS] = {{0.09443876248897524, -0.1412552950209542, 0.03002919410514742},
                              {0.2502131229653333, 0.20412935229379994, -0.3829421127572619},
                              {1.0, 1.0, 0.0}};
constexpr double TI[NS][NS] = {{4.178718591551904, 0.32768282076106237, 0.5233764454994495},
                               {-4.178718591551904, -0.32768282076106237, 0.47662355450055044},
                               {0.5028726349457868, -2.571926949855605, 0.5960392048282249}};
constexpr double E[NS] = {(-13.0 - 7.0 * SQRT6) / 3.0, (-13.0 + 7.0 * SQRT6) / 3.0, -1.0 / 3.0};
constexpr double PD[NS][3] = {
    {13.0 / 3.0 + 7.0 * SQRT6 / 3.0, -23.0 / 3.0 - 22.0 * SQRT6 / 3.0, 10.0 / 3.0 + 5.0 * SQRT6},
    {13.0 / 3.0 - 7.0 * SQRT6 / 3.0, -23.0 / 3.0 + 22.0 * SQRT6 / 3.0, 10.0 / 3.0 - 5.0 * SQRT6},
    {1.0 / 3.0, -8.0 / 3.0, 10.0 / 3.0}};
static_assert(T[0][0] != 0.0 && T[0][1] != 0.0 && T[0][2] != 0.0, "the FAST NaN / Inf guard of the Newton increments reads all three W through Z_1");
static_assert(T[2][0] == 1.0 && T[2][1] == 1.0 && T[2][2] == 0.0 && TI[1][0] == -TI[0][0] && TI[1][1] == -TI[0][1],
              "the shortcuts of stage_increments / transformed_residuals (FAST) read these entries off the matrices");
// Z = (T (x) I) W for one component.  FAST: the last row of T is (1, 1, 0) -- Z_3 = W_1 + W_2, no product with 0.
template <int MATH>
__device__ __forceinline__ void stage_increments(double w0, double w1, double w2, double& z0, double& z1, double& z2)
{
    z0 = T[0][0] * w0 + T[0][1] * w1 + T[0][2] * w2;
    z1 = T[1][0] * w0 + T[1][1] * w1 + T[1][2] * w2;
    if constexpr (MATH == MATH_FAST) z2 = w0 + w1;
    else z2 = T[2][0] * w0 + T[2][1] * w1 + T[2][2] * w2;
}
// (TI (x) I) F for one component.  FAST: the first two rows of TI share their first two entries up to the sign.
template <int MATH>
__device__ __forceinline__ void transformed_residuals(double f0, double f1, double f2, double& r0, double& r1, double& r2)
{
    if constexpr (MATH == MATH_FAST) {
        const double u = TI[0][0] * f0 + TI[0][1] * f1;
        r0 = u + TI[0][2] * f2;
        r1 = TI[1][2] * f2 - u;
    } else {
        r0 = TI[0][0] * f0 + TI[0][1] * f1 + TI[0][2] * f2;
        r1 = TI[1][0] * f0 + TI[1][1] * f1 + TI[1][2] * f2;
    }
    r2 = TI[2][0] * f0 + TI[2][1] * f1 + TI[2][2] * f2;
}
// 1/x in the job's arithmetic: IEEE division (STRICT) or the Newton-refined v_rcp_f64 the RK45 loop uses
// for its error ratio (FAST) -- an attempt of this pass divides ~30 times
template <int MATH>
__device__ __forceinline__ double recip(double x)
{
    if constexpr (MATH == MATH_FAST) return fast_rcp(x); else return 1.0 / x;
}

// N reciprocals at once.  FAST: ONE Newton-refined v_rcp_f64 of the product and 3 (N - 1) multiplications
// (x_0 ... x_{k-1} as prefix products forwards, the running reciprocal of the tail backwards) instead of N reciprocals
// of 6 issue slots each (v_rcp_f64 is quarter rate): 18 instead of 30 slots for five, 33 instead of 60 for ten.  The
// arguments are positive and moderate here (tolerances atol + rtol |y|; g/h - J_ii and |a/h - J_ii - i b/h|^2 with
// h in [1e-6, 1e4] minutes), so their product stays far inside the double range.  STRICT: IEEE divisions, one by one.
template <int MATH, int N>
__device__ __forceinline__ void recip_all(const double (&x)[N], double (&inv)[N])
{
    if constexpr (MATH == MATH_FAST && N > 1) {
        double pre[N];   // pre[k] = x_0 ... x_k
        pre[0] = x[0];
#pragma unroll
        for (int k = 1; k < N; ++k) pre[k] = pre[k - 1] * x[k];
        double r = fast_rcp(pre[N - 1]);   // 1 / (x_0 ... x_k) while k runs down
#pragma unroll
        for (int k = N - 1; k > 0; --k) {
            inv[k] = r * pre[k - 1];
            r *= x[k];
        }
        inv[0] = r;
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) inv[k] = recip<MATH>(x[k]);
    }
}

// (1 / x)^(1/4) for the step-size controller.  STRICT: the IEEE quotient and the correctly rounded fourth root
// (hlm_strict_pow.h; pow(1.0 / x, 0.25) as the oracle writes it).  FAST: as inv_fifth_root of the RK45
// loop -- an f32 exp2 / log2 seed z ~ x^(-1/4), residual e = 1 - x z^4, exact root z (1 - e)^(-1/4) =
// z (1 + e/4 + 5/32 e^2 + ...) -- a dozen instructions instead of two double-precision square roots and a reciprocal
// (~45).  x = err + 1e-16 >= 1e-16; beyond 1e30 the factor is clamped to minScale whatever its value, so x is capped
// there (keeps the f32 seed finite, an infinite estimate included).
template <int MATH>
__device__ __forceinline__ double inv_fourth_root(double x)
{
    if constexpr (MATH == MATH_STRICT) {
        return strict_pow_fourth(1.0 / x);
    } else {
        const double xc = fmin(x, 1e30);
        double z = (double)__builtin_amdgcn_exp2f(-0.25f * __builtin_amdgcn_logf((float)xc));
        const double z2 = z * z;
        const double e = __builtin_fma(-xc, z2 * z2, 1.0);
        const double p = __builtin_fma(e, 5.0 / 32.0, 0.25);
        z = __builtin_fma(z * e, p, z);
        return z;
    }
}

constexpr int NEWTON_MAXITER = 10;
constexpr double NEWTON_TOL = 1e-3;
constexpr double FD_EPS = 1e-8;

__device__ __forceinline__ void cswap(bool sw, double& a, double& b)
{
    const double u = a, v = b;
    a = sw ? v : u;
    b = sw ? u : v;
}

// real 5x5 LU with partial pivoting, all indices compile-time (registers); inv = reciprocal pivots
template <int NEQ>
__device__ __forceinline__ bool lu_real(double (&a)[NEQ][NEQ], int (&piv)[NEQ], double (&inv)[NEQ])
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NEQ; ++k) {
        int p = k;
        double best = fabs(a[k][k]);
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) {
            const double v = fabs(a[i][k]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        ok = ok && best > 1e-300;
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) {
            const bool sw = p == i;
#pragma unroll
            for (int j = 0; j < NEQ; ++j) cswap(sw, a[k][j], a[i][j]);
        }
        inv[k] = 1.0 / a[k][k];
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) {
            const double m = a[i][k] * inv[k];
            a[i][k] = m;
#pragma unroll
            for (int j = k + 1; j < NEQ; ++j) a[i][j] -= m * a[k][j];
        }
    }
    return ok;
}

template <int NEQ>
__device__ __forceinline__ void solve_real(const double (&a)[NEQ][NEQ], const int (&piv)[NEQ],
                                           const double (&inv)[NEQ], double (&b)[NEQ])
{
#pragma unroll
    for (int k = 0; k < NEQ; ++k) {
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) cswap(piv[k] == i, b[k], b[i]);
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) b[i] -= a[i][k] * b[k];
    }
#pragma unroll
    for (int i = NEQ - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int j = i + 1; j < NEQ; ++j) s -= a[i][j] * b[j];
        b[i] = s * inv[i];
    }
}

// the same for a complex matrix held as (re, im); pivot on |re| + |im|
template <int NEQ>
__device__ __forceinline__ bool lu_cplx(double (&ar)[NEQ][NEQ], double (&ai)[NEQ][NEQ], int (&piv)[NEQ],
                                        double (&ir)[NEQ], double (&ii)[NEQ])
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NEQ; ++k) {
        int p = k;
        double best = fabs(ar[k][k]) + fabs(ai[k][k]);
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) {
            const double v = fabs(ar[i][k]) + fabs(ai[i][k]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        ok = ok && best > 1e-300;
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) {
            const bool sw = p == i;
#pragma unroll
            for (int j = 0; j < NEQ; ++j) {
                cswap(sw, ar[k][j], ar[i][j]);
                cswap(sw, ai[k][j], ai[i][j]);
            }
        }
        const double d = 1.0 / (ar[k][k] * ar[k][k] + ai[k][k] * ai[k][k]);
        ir[k] = ar[k][k] * d;
        ii[k] = -ai[k][k] * d;
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) {
            const double mr = ar[i][k] * ir[k] - ai[i][k] * ii[k];
            const double mi = ar[i][k] * ii[k] + ai[i][k] * ir[k];
            ar[i][k] = mr;
            ai[i][k] = mi;
#pragma unroll
            for (int j = k + 1; j < NEQ; ++j) {
                ar[i][j] -= mr * ar[k][j] - mi * ai[k][j];
                ai[i][j] -= mr * ai[k][j] + mi * ar[k][j];
            }
        }
    }
    return ok;
}

template <int NEQ>
__device__ __forceinline__ void solve_cplx(const double (&ar)[NEQ][NEQ], const double (&ai)[NEQ][NEQ],
                                           const int (&piv)[NEQ], const double (&ir)[NEQ], const double (&ii)[NEQ],
                                           double (&br)[NEQ], double (&bi)[NEQ])
{
#pragma unroll
    for (int k = 0; k < NEQ; ++k) {
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) {
            const bool sw = piv[k] == i;
            cswap(sw, br[k], br[i]);
            cswap(sw, bi[k], bi[i]);
        }
#pragma unroll
        for (int i = k + 1; i < NEQ; ++i) {
            br[i] -= ar[i][k] * br[k] - ai[i][k] * bi[k];
            bi[i] -= ar[i][k] * bi[k] + ai[i][k] * br[k];
        }
    }
#pragma unroll
    for (int i = NEQ - 1; i >= 0; --i) {
        double sr = br[i], si = bi[i];
#pragma unroll
        for (int j = i + 1; j < NEQ; ++j) {
            sr -= ar[i][j] * br[j] - ai[i][j] * bi[j];
            si -= ar[i][j] * bi[j] + ai[i][j] * br[j];
        }
        br[i] = sr * ir[i] - si * ii[i];
        bi[i] = sr * ii[i] + si * ir[i];
    }
}

// ---- models whose Jacobian is LOWER TRIANGULAR (a cascade of stores: each one only feels the stores
// upstream of it -- Model 204 and the DummyModel are such).  Then mu/h I - J is triangular as well: no
// factorisation, no pivoting (LU with partial pivoting would not pivot either: |mu/h - J_ii| exceeds the
// entries below it), the solves are forward substitutions.  Entries a model's jacobian() leaves at the
// literal 0.0 drop out at compile time.
// SKIP(i, j): the model declares J_ij zero whatever the state (Model::jacobian_zero; FAST: the product 0 * b_j is
// left out -- it is +-0 unless b_j is Inf / NaN, which ends the Newton iteration through its own check anyway)
template <class Model, int MATH, int NEQ>
__device__ __forceinline__ void solve_real_tri(const double (&Jm)[NEQ][NEQ], const double (&inv)[NEQ], double (&b)[NEQ])
{
#pragma unroll
    for (int i = 0; i < NEQ; ++i) {
        double s = b[i];
#pragma unroll
        for (int j = 0; j < i; ++j)
            if (!(MATH == MATH_FAST && Model::jacobian_zero(i, j))) s += Jm[i][j] * b[j];  // the matrix entry is -J_ij
        b[i] = s * inv[i];
    }
}
template <class Model, int MATH, int NEQ>
__device__ __forceinline__ void solve_cplx_tri(const double (&Jm)[NEQ][NEQ], const double (&ir)[NEQ], const double (&ii)[NEQ],
                                               double (&br)[NEQ], double (&bi)[NEQ])
{
#pragma unroll
    for (int i = 0; i < NEQ; ++i) {
        double sr = br[i], si = bi[i];
#pragma unroll
        for (int j = 0; j < i; ++j) {
            if (MATH == MATH_FAST && Model::jacobian_zero(i, j)) continue;
            sr += Jm[i][j] * br[j];
            si += Jm[i][j] * bi[j];
        }
        br[i] = sr * ir[i] - si * ii[i];
        bi[i] = sr * ii[i] + si * ir[i];
    }
}

}  // namespace radau

// MATH: the arithmetic of the model's RHS (hlm_set_math_mode / hlm_device_job.math_mode), as in the RK45
// kernel.  It matters here more than there: an attempt evaluates the RHS 6 + 3 per Newton iteration
// = ~14 times (STRICT's x^(2/3): hlm_strict_pow.h, ~40 instructions each; OCML's pow took ~200 until round 4).
// Occupancy: the general path keeps two 5x5 LU factorisations (75 doubles) in registers: one wave per
// SIMD.  With a triangular Jacobian they are gone and two waves fit.
#ifdef HLM_RADAU_COUNTERS
// experiment builds: where the lanes of a wave wait for each other.  [2k] wave-level trips through section k,
// [2k + 1] lane-level trips (sum of the active lanes): k = 0 step attempts (set-ups), 1 Newton iterations, 2 converged-attempt
// tails, 3 trips of the state machine's loop (lanes not yet done)
__device__ unsigned long long g_radau_ctr[8];
#define HLM_RCOUNT(k)                                                                    \
    do {                                                                                 \
        const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true);                 \
        if (lane == (int)__builtin_ctzll(m_)) {                                          \
            atomicAdd(&g_radau_ctr[2 * (k)], 1ull);                                      \
            atomicAdd(&g_radau_ctr[2 * (k) + 1], (unsigned long long)__builtin_popcountll(m_)); \
        }                                                                                \
    } while (0)
#else
#define HLM_RCOUNT(k) do { } while (0)
#endif
#ifndef HLM_RADAU_TRI_WAVES
#define HLM_RADAU_TRI_WAVES 2
#endif
// the finish + set-up section of the state machine runs when this fraction of the wave's live lanes waits for it
#ifndef HLM_RADAU_SETUP_NUM
#define HLM_RADAU_SETUP_NUM 5
#endif
#ifndef HLM_RADAU_SETUP_DEN
#define HLM_RADAU_SETUP_DEN 8
#endif
constexpr int RADAU_SETUP_NUM = HLM_RADAU_SETUP_NUM, RADAU_SETUP_DEN = HLM_RADAU_SETUP_DEN;

// TIME PIECES (round 5).  The window [t0, tf] is cut into npieces equal parts and what a lane draws from the device
// counter is an ITEM (piece p, list slot): it integrates that system from where its previous item left it until the first
// attempt that would start at or after the piece's end -- the pause protocol of the RK45 queue kernel: no step is
// clipped to a piece's end, so every system's step sequence, and with it every result, is what it was -- leaves
// {y, t, h, the step memory, the held samples | the two counters, the next query} in a record by list slot (92 bytes
// for Model 204) and draws again.  The items' numbering: "entries(p)" below.
// Why: a lane that takes whole systems (round 4's refill) holds its last one for a third of the launch -- 419 003
// flagged systems on 131 072 lanes are 3.2 systems per lane, the lanes that finish their third first take a fourth, and
// the launch ends ~3.9 system-times after it began with 3.2 of work per lane: a fifth of the machine idles in the
// tail (measured: waves resident 90 % of the launch, and DONE lanes inside the resident ones).  With 16 pieces the
// tail is that of an item 1/16 as long.  "Continue" mode (the systems start at their flag times): the list is ordered
// by the piece of the flag time, cum[p] = the entries flagged in pieces <= p exist in piece p, and a system's first
// item starts it from its flag state.
constexpr int RADAU_MAX_PIECES = 32;
constexpr int RADAU_ITEM_PIECE_SHIFT = 26;   // a lane's item = list slot | piece << 26 | "first item of its system" << 31
// The state machine is left for the queue (records out, the next items in: an atomic and three dependent loads with all
// 64 lanes waiting, ~5 us) once this many lanes have ended their items, or one of them has waited this many trips
#ifndef HLM_RADAU_BREAK_LANES
#define HLM_RADAU_BREAK_LANES 16
#endif
#ifndef HLM_RADAU_BREAK_TRIPS
#define HLM_RADAU_BREAK_TRIPS 48
#endif
constexpr int RADAU_ENDED = 0x40000000;   // done[slot]: the system is through (the attempt cap, or a lost predecessor): later items skip it
struct RadauQueue {
    int npieces;       // >= 1
    int merged_to;     // E: a system's FIRST item runs through the pieces up to E in one go (0: every piece an item of its own)
    const int* cum;    // [npieces] list entries that exist in piece p; null: all of them in every piece (restart at t0)
    double* rec_f64;   // [N_EQ + 3 + max(NF, 1)][cap]: y, t, h, the step memory and the held samples of a system paused at a piece's end, by list slot
    int* rec_i32;      // [3][cap]: n_attempt, n_accept, next_q
    int* done;         // [cap], zeroed: p + 1 once piece p of the slot's system is through (| RADAU_ENDED)
    int cap;
};
#ifndef HLM_RADAU_WAIT_LIMIT_TICKS
#define HLM_RADAU_WAIT_LIMIT_TICKS 2000000000ull  // 20 s of wall_clock64() (100 MHz)
#endif

template <class Model, int MATH>
__global__ void __launch_bounds__(WAVE, (Model::HAS_JACOBIAN && Model::JACOBIAN_LOWER_TRIANGULAR) ? HLM_RADAU_TRI_WAVES : 1) radau_fallback_kernel(const Rk45Args A, const int* __restrict__ list,
                                                                 const int* __restrict__ count, int* __restrict__ next_item, const RadauQueue Q)
{
    using namespace radau;
    constexpr int NEQ = Model::N_EQ, NFA = Model::NF > 0 ? Model::NF : 1;
    [[maybe_unused]] const int lane = (int)threadIdx.x;   // (the counter builds)
    const int nflagged = *count;
    const double tf = A.tf;

    // dense layouts as in hlm_rk45_dense_kernel.h: 0 [sys][q][c], 2 [tile][q][c][lane]
    // output selection (hlm_device_job.dense_ncomp): rows of K components, component c at position pos(c) or absent
    const bool tiled = A.dense_tiled == 2;
    const int K = A.dense_k > 0 ? A.dense_k : NEQ;
    auto pos = [&](int c) -> int { return A.dense_k > 0 ? A.dense_pos[c] : c; };  // wave-uniform
    const size_t q_stride = tiled ? (size_t)WAVE * (size_t)K : (size_t)K;
    const size_t c_stride = tiled ? (size_t)WAVE : (size_t)1;

    // ---- the system a lane integrates.  The grid is persistent (as many one-wave workgroups as the device holds) and a
    // lane that has ended its item draws the next one from a device counter (round 4: whole systems; round 5: time
    // pieces of them, above) -- no wave idles behind its longest lane.  What made lane refill lose in the RK45 loop --
    // neighbours of different forcing cells in one wave defeat its wave-uniform cheap RHS paths -- does not exist here:
    // every lane evaluates the general RHS with its own forcing samples.
    // `item` = slot | piece << ITEM_PIECE_SHIFT is all the lane keeps of where its system came from (one register across
    // the state machine, where the system number was before the pieces: the list gives that back when the item ends)
    // (a launch of more than 2^26 systems runs in one piece -- launch_radau sees to it -- and the slot takes all the bits)
    constexpr int ITEM_PIECE_SHIFT = RADAU_ITEM_PIECE_SHIFT, ITEM_FIRST = (int)0x80000000u;   // (sign bit: the system's first item)
    const int ITEM_SLOT_MASK = Q.npieces > 1 ? (1 << ITEM_PIECE_SHIFT) - 1 : 0x7fffffff;
    auto piece_of = [&](int item) -> int { return Q.npieces > 1 ? (item >> ITEM_PIECE_SHIFT) & 31 : 0; };
    typename Model::Local P{};
    double y[NEQ];
    typename Model::Forc FV{};
    double t = A.t0, h = A.h0;
#pragma unroll
    for (int i = 0; i < NEQ; ++i) y[i] = 0.0;
    // LANE MEMORY IN LDS (round 5).  What a lane touches once per attempt or less lives in LDS, [word][lane] (a lane's own
    // 8-byte column of every word: no bank conflicts, no synchronisation -- nobody else reads it): the slope at the step's
    // start (set-up -> error estimate), the controller's proposal and the step's end (set-up -> commit), the held
    // samples and their ends, the counters, where the system came from and where its rows go.  ~16 doubles per lane =
    // 8 KB per wave of the CU's 160 KB, for 31 registers the state machine no longer holds across its trips: the two
    // hot sections (300+ instructions each with 30 doubles of temporaries) spilled to scratch inside themselves before.
    constexpr int LM_HCTL = NEQ, LM_TEND = NEQ + 1, LM_TB = NEQ + 2, LM_F = LM_TB + NFA, LM_FCOL = LM_F + NFA, LM_ROW = LM_FCOL + 1, LM_HCHG = LM_ROW + 1,
                  LM_WORDS = LM_HCHG + 1;
    __shared__ double lane_f64[LM_WORDS][WAVE];   // [0, NEQ) f0 | h_ctl (0 unless the step was clipped) | t_end | tb[NF] | F[NF] | fcol, dense_row (as bits) | h_change
    __shared__ int lane_i32[5][WAVE];             // n_attempt, n_accept, next_q, item, change_pending
#define f0_(i) lane_f64[(i)][lane]
#define h_ctl_ lane_f64[LM_HCTL][lane]
#define t_end_ lane_f64[LM_TEND][lane]
#define tb_(j) lane_f64[LM_TB + (j)][lane]
#define F_(j) lane_f64[LM_F + (j)][lane]
#define fcol_ reinterpret_cast<unsigned long long*>(&lane_f64[LM_FCOL][0])[lane]
#define dense_row_ reinterpret_cast<double**>(&lane_f64[LM_ROW][0])[lane]
#define n_attempt_ reinterpret_cast<uint32_t*>(&lane_i32[0][0])[lane]
#define n_accept_ reinterpret_cast<uint32_t*>(&lane_i32[1][0])[lane]
#define next_q_ lane_i32[2][lane]
#define item_ lane_i32[3][lane]
    // step memory across forcing changes (the rule and its reason: the CPU restatement under oracle/, "STEP MEMORY ..."): the first attempt after a change of
    // a held sample starts from the step accepted first after the previous change (h_change; 0 = none yet)
#define h_change_ lane_f64[LM_HCHG][lane]
#define change_pending_ lane_i32[4][lane]
    auto held_forcing = [&]() -> typename Model::Forc {
        double F[NFA];
#pragma unroll
        for (int j = 0; j < NFA; ++j) F[j] = F_(j);
        return Model::prep_forcing(P, F);
    };

    // entries(p): the list entries that exist in piece p.  The queue numbers its items ROUND-major: round 0 holds the first
    // item of every system -- from its own start through the end of piece max(first piece, E), E = Q.merged_to -- and
    // round k >= 1 piece max(first, E) + k of every system that has one, i.e. of the entries below entries(P - 1 - k).
    // Round 0 starts every flagged system at once, each where its own integration starts, and the item a record waits
    // for was drawn a whole round earlier.  (Restart at t0: E = 0, round = piece.  Numbered piece-major, "continue" mode
    // began with the few systems flagged in the first piece and left the machine waiting for their records.)
    // "Continue" mode takes E = P - 7: whole systems up to the last six pieces, which are handed out one by one.  Whole
    // systems to the end (round 4) left the launch's last tenth to its tail -- a quarter of the cold C3's flags fall in
    // the window's last eighth, and those systems, drawn last, are two pieces long; every piece an item (E = 0) balances
    // as well, but scatters the late rounds' few items over all waves (41 instead of 44 lanes per instruction).  What
    // removes a tail is the granularity of the LAST items only.
    // (the counts sit in LDS: a draw walks them, and from global memory that was up to 32 dependent loads -- 50 us with
    // the whole wave waiting -- per item in "continue" mode)
    __shared__ int q_entries[RADAU_MAX_PIECES];
    if (lane < Q.npieces) q_entries[lane] = Q.cum ? Q.cum[lane] : nflagged;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // read by the other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto entries = [&](int p) -> int { return q_entries[p]; };
    auto decode = [&](int i) -> bool {
        long long before = 0;
        for (int k = 0; k < Q.npieces - Q.merged_to; ++k) {
            const int c = entries(Q.npieces - 1 - k);
            if ((long long)i < before + (long long)c) {
                const int slot = (int)((long long)i - before);
                int first = 0;   // the piece the entry's system starts in
                while (first + 1 < Q.npieces && slot >= entries(first)) ++first;
                const int piece = (first > Q.merged_to ? first : Q.merged_to) + k;   // (the piece the item ends with)
                item_ = slot | piece << ITEM_PIECE_SHIFT | (k == 0 ? ITEM_FIRST : 0);
                return true;
            }
            before += (long long)c;
        }
        return false;
    };
    // the end of piece p (tf for the last one): a lane pauses before the first attempt that starts at or after it
    const double piece_len = (tf - A.t0) / (double)Q.npieces;
    auto piece_end = [&](int p) -> double { return p + 1 < Q.npieces ? vmin(A.t0 + (double)(p + 1) * piece_len, tf) : tf; };
    // what every item of a system needs, whatever its piece
    auto begin_item = [&](int sys) {
        P = Model::template load<MATH>(A.sp, sys);
        if constexpr (Model::NF > 0) fcol_ = A.forc_col ? (unsigned long long)A.forc_col[sys] : (unsigned long long)sys;
        dense_row_ = nullptr;
        if (A.nq > 0)
            dense_row_ = !tiled ? A.dense + (size_t)sys * (size_t)A.nq * (size_t)K
                                : A.dense + (size_t)(sys / WAVE) * (size_t)A.nq * q_stride +
                                      (size_t)(sys % WAVE);
    };
    // the first item of a system: its start
    auto start_system = [&](int sys) {
        // no sample held yet (tb = -inf: the first set-up samples at the lane's t and sees no change), no step remembered
#pragma unroll
        for (int j = 0; j < NFA; ++j) { tb_(j) = -__builtin_inf(); F_(j) = 0.0; }
        FV = held_forcing();
        h_change_ = 0.0;
        change_pending_ = 0;
        t = A.t0;
        h = A.h0;
        int next_q = 0;
        n_attempt_ = n_accept_ = 0u;
        double* const dense_row = dense_row_;
        if (A.flag_state) {
            // "continue" mode (hlm_device_job.d_flag_state): take over from the state at which RK45 gave up; the rows
            // it wrote up to there stay, the later ones (zero so far) are this pass's
            const double* fs = A.flag_state + (size_t)sys * (NEQ + 1);
#pragma unroll
            for (int i = 0; i < NEQ; ++i) y[i] = fs[i];
            t = fs[NEQ];
            while (next_q < A.nq && A.query[next_q] <= t) ++next_q;
        } else {
            // restart at t0 (radau_kernel.cu:44-52); queries at or before t0 are never produced: zero rows
#pragma unroll
            for (int i = 0; i < NEQ; ++i) y[i] = A.y0[(size_t)sys * NEQ + i];
            while (next_q < A.nq && A.query[next_q] <= t) {
#pragma unroll
                for (int c = 0; c < NEQ; ++c)
                    if (c < K) dense_row[(size_t)next_q * q_stride + (size_t)c * c_stride] = 0.0;
                ++next_q;
            }
        }
        next_q_ = next_q;
    };
    // a later item: the record piece - 1 left (device-coherent loads: another wave, maybe of another XCD, stored it)
    auto resume_system = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NEQ; ++i) y[i] = coherent_load(&Q.rec_f64[(size_t)i * Q.cap + slot]);
        t = coherent_load(&Q.rec_f64[(size_t)NEQ * Q.cap + slot]);
        h = coherent_load(&Q.rec_f64[(size_t)(NEQ + 1) * Q.cap + slot]);
        n_attempt_ = (uint32_t)coherent_load(&Q.rec_i32[(size_t)0 * Q.cap + slot]);
        n_accept_ = (uint32_t)coherent_load(&Q.rec_i32[(size_t)1 * Q.cap + slot]);
        next_q_ = coherent_load(&Q.rec_i32[(size_t)2 * Q.cap + slot]);
        // the samples the paused lane held (a pause is often ON a sample boundary: the first set-up of this piece must see
        // whether the sample changes there) and its step memory; tb = the most negative finite number: "held, ends before any t"
        h_change_ = coherent_load(&Q.rec_f64[(size_t)(NEQ + 2) * Q.cap + slot]);
#pragma unroll
        for (int j = 0; j < NFA; ++j) {
            F_(j) = coherent_load(&Q.rec_f64[(size_t)(NEQ + 3 + j) * Q.cap + slot]);
            tb_(j) = -1.7976931348623157e308;
        }
        FV = held_forcing();
        change_pending_ = 0;   // (never pending at a pause: a lane pauses before an attempt, a change is noticed in one)
    };
    auto pause_system = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NEQ; ++i) coherent_store(&Q.rec_f64[(size_t)i * Q.cap + slot], y[i]);
        coherent_store(&Q.rec_f64[(size_t)NEQ * Q.cap + slot], t);
        coherent_store(&Q.rec_f64[(size_t)(NEQ + 1) * Q.cap + slot], h);
        coherent_store(&Q.rec_i32[(size_t)0 * Q.cap + slot], (int)n_attempt_);
        coherent_store(&Q.rec_i32[(size_t)1 * Q.cap + slot], (int)n_accept_);
        coherent_store(&Q.rec_i32[(size_t)2 * Q.cap + slot], next_q_);
        coherent_store(&Q.rec_f64[(size_t)(NEQ + 2) * Q.cap + slot], h_change_);
#pragma unroll
        for (int j = 0; j < NFA; ++j) coherent_store(&Q.rec_f64[(size_t)(NEQ + 3 + j) * Q.cap + slot], F_(j));
    };
    // the outputs of a finished system (capped: the attempt cap ended it before tf)
    auto store_system = [&](int sys, bool capped) {
        // both counters ACCUMULATE: after the fallback they hold the RK45 work spent on the system up to
        // its flag plus the implicit attempts / accepted steps of this pass
        if (A.n_attempt) A.n_attempt[sys] += n_attempt_;
        if (A.n_accept) A.n_accept[sys] += n_accept_;
        double* const dense_row = dense_row_;
        const bool failed = capped && t < tf;
        if (failed) A.stiff[sys] = HLM_SYS_CAPPED;  // success leaves the RK45 flag in place (rk45_api.hpp:189-247 does)
#pragma unroll
        for (int i = 0; i < NEQ; ++i) A.y_final[(size_t)sys * NEQ + i] = failed ? 0.0 : y[i];
        for (int next_q = next_q_; next_q < A.nq; ++next_q) {  // rows never reached (later than tf, or after the cap): zero
#pragma unroll
            for (int c = 0; c < NEQ; ++c)
                if (c < K) dense_row[(size_t)next_q * q_stride + (size_t)c * c_stride] = 0.0;
        }
    };

    auto sample = [&](int j, double& tbj) -> double {
        // zero-order hold (rk45_kernel.cu:90-98); the index is nudged so that its interval
        // really contains t (a step clipped to a boundary lands exactly on it)
        unsigned long long idx = forcing_index(t, A.forc_dtmin[j], A.forc_nT[j]);
        while (idx + 1ull < A.forc_nT[j] && (double)(idx + 1ull) * A.forc_dtmin[j] <= t) ++idx;
        tbj = (idx + 1ull < A.forc_nT[j]) ? (double)(idx + 1ull) * A.forc_dtmin[j] : __builtin_inf();
        return (double)A.forc[A.forc_base[j] + (size_t)idx * (size_t)A.forc_ld + (size_t)fcol_];
    };

    // ---- the attempts, as a state machine whose trip is ONE simplified-Newton iteration (round 4) ----------------------
    // Written as "for each attempt: set-up, Newton loop, tail" (round 2/3) the wave runs every attempt's Newton loop
    // until its slowest lane has converged: a lane needs 2.49 iterations per attempt, its wave ran 4.82 -- 29 of 64
    // lanes active in the loop that is three quarters of the pass (DESIGN.md "Radau-IIA fallback").  Here every lane
    // is in one of two phases.  PH_NEWTON: it takes part in the trip's Newton iteration (3 RHS + two triangular
    // solves) and, when that iteration ends its attempt (converged / diverging / out of iterations), moves to
    // PH_SETUP and waits.  PH_SETUP: it waits for the wave's next finish + set-up section, which runs once enough
    // lanes wait (RADAU_SETUP_NUM / RADAU_SETUP_DEN of the live ones) or nobody is iterating: there the lane finishes
    // the attempt it came from (error estimate, accept / reject, dense rows, commit -- or h / 2 after a failed
    // iteration) and sets up its next one (forcing samples, clipping, f0, Jacobian, the 15 reciprocals), then iterates
    // again.  Each lane executes exactly the operation sequence of the nested loops, so every result is the same bit
    // for bit; only WHEN a lane does what differs, and the Newton iterations run with the lanes that need them.
    // the other phases are the queue's: PH_FINISHED the lane's item has ended (results or the record go out), PH_DRAW it
    // needs an item, PH_LOAD it has one and waits for the record of the piece before (never, normally: that piece was
    // drawn a whole list earlier), PH_DONE the queue is empty
    enum : int { PH_NEWTON = 0, PH_SETUP = 1, PH_DONE = 2, PH_FINISHED = 3, PH_DRAW = 4, PH_LOAD = 5 };
    enum : int { END_NONE = 0, END_CONVERGED = 1, END_FAILED = 2 };
    int phase = PH_DRAW, ended = END_NONE;
    unsigned long long waiting_since = 0ull;   // (wave-uniform) since when a lane of this wave has been waiting for a record, 0 = none is
    // state of the attempt a lane is in
    constexpr bool TRI = Model::HAS_JACOBIAN && Model::JACOBIAN_LOWER_TRIANGULAR;
    double Jm[NEQ][NEQ];
    [[maybe_unused]] double Mr[TRI ? 1 : NEQ][TRI ? 1 : NEQ], Cr[TRI ? 1 : NEQ][TRI ? 1 : NEQ], Ci[TRI ? 1 : NEQ][TRI ? 1 : NEQ];
    double inv_r[NEQ], inv_cr[NEQ], inv_ci[NEQ];
    [[maybe_unused]] int piv_r[NEQ], piv_c[NEQ];
    double Z[NS][NEQ], W[NS][NEQ], inv_sc[NEQ];
    double inv_h = 0.0, g_h = 0.0, a_h = 0.0, b_h = 0.0, prev = __builtin_inf();
    int it = 0;
#pragma unroll
    for (int i = 0; i < NEQ; ++i) {
        inv_r[i] = inv_cr[i] = inv_ci[i] = inv_sc[i] = 0.0;
#pragma unroll
        for (int j = 0; j < NEQ; ++j) Jm[i][j] = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) Z[s2][i] = W[s2][i] = 0.0;
    }
    auto solve_r = [&](double (&b)[NEQ]) {
        if constexpr (TRI) solve_real_tri<Model, MATH>(Jm, inv_r, b); else solve_real(Mr, piv_r, inv_r, b);
    };
    auto solve_c = [&](double (&br)[NEQ], double (&bi)[NEQ]) {
        if constexpr (TRI) solve_cplx_tri<Model, MATH>(Jm, inv_cr, inv_ci, br, bi); else solve_cplx(Cr, Ci, piv_c, inv_cr, inv_ci, br, bi);
    };

    // Two loops.  The inner one is the state machine; it is left when enough lanes have ended their items (HLM_RADAU_BREAK_*),
    // and the outer one's body -- results or records of the ended items out, the next items in -- runs between: kept out
    // of the hot loop so that its operands (the launch's pointers, the parameter record) are not live registers in there.
#pragma unroll 1
    while (true) {
    if (phase == PH_FINISHED) {
        const int item = item_, slot = item & ITEM_SLOT_MASK, piece = piece_of(item);
        if (!(t < tf) || n_attempt_ >= A.cap) {
            store_system(list[slot], t < tf);   // the system is through (tf, or the attempt cap before it)
            if (piece + 1 < Q.npieces) coherent_store(&Q.done[slot], RADAU_ENDED);   // (the cap, in an earlier piece: its later items have nothing to do)
        } else {
            // paused at the end of its piece: the record, its stores acknowledged at device scope, then the counter
            pause_system(slot);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            coherent_store(&Q.done[slot], piece + 1);
        }
        phase = PH_DRAW;
    }
    // ---- the queue: a small loop of its own (as `continue`s of the outer loop its paths cost the state machine 300 spilled registers)
#pragma unroll 1
    while (true) {
    if (phase == PH_DRAW) {
        const int drawn = atomicAdd(next_item, 1);
        phase = decode(drawn) ? PH_LOAD : PH_DONE;
    }
    if (phase == PH_LOAD) {
        const int item = item_, slot = item & ITEM_SLOT_MASK, piece = piece_of(item);
        const int sys = list[slot];
        const bool first = Q.npieces == 1 || item < 0;   // (ITEM_FIRST) no earlier item held this list entry
        if (first) {
            begin_item(sys);
            start_system(sys);
            phase = PH_SETUP;
        } else {
            const int v = coherent_load(&Q.done[slot]);
            if (v >= piece) {
                asm volatile("" ::: "memory");   // (the record is loaded after the counter, also as the compiler sees it)
                if (v & RADAU_ENDED) {
                    phase = PH_DRAW;
                } else {
                    begin_item(sys);
                    resume_system(slot);
                    phase = PH_SETUP;
                }
            } else if (waiting_since != 0ull && wall_clock64() - waiting_since > HLM_RADAU_WAIT_LIMIT_TICKS) {
                // a lost predecessor (never seen): the system comes back as stopped by a resource limit
                A.stiff[sys] = HLM_SYS_CAPPED;
#pragma unroll
                for (int i = 0; i < NEQ; ++i) A.y_final[(size_t)sys * NEQ + i] = 0.0;
                coherent_store(&Q.done[slot], RADAU_ENDED);
                phase = PH_DRAW;
            }
        }
    }
    // again while a lane must draw once more (a skipped item; every draw moves the counter) or all that is left to the
    // wave is waiting for records
    const bool some_wait = __builtin_amdgcn_ballot_w64(phase == PH_LOAD) != 0ull;
    waiting_since = !some_wait ? 0ull : (waiting_since != 0ull ? waiting_since : wall_clock64());
    if (__builtin_amdgcn_ballot_w64(phase == PH_DRAW) != 0ull) continue;
    if (__builtin_amdgcn_ballot_w64(phase == PH_SETUP || phase == PH_NEWTON) != 0ull || !some_wait) break;
    __builtin_amdgcn_s_sleep(16);
    }
    if (__builtin_amdgcn_ballot_w64(phase != PH_DONE) == 0ull) break;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no load of the block above is waited for inside the loop, behind the row stores
    int polls_in = 0, ended_trips = 0;
#pragma unroll 1
    while (true) {
        const unsigned long long m_setup = __builtin_amdgcn_ballot_w64(phase == PH_SETUP);
        const unsigned long long m_newton = __builtin_amdgcn_ballot_w64(phase == PH_NEWTON);
        if ((m_setup | m_newton) == 0ull) break;   // (only finished lanes left)
        HLM_RCOUNT(3);
        const int n_setup = __builtin_popcountll(m_setup), n_newton = __builtin_popcountll(m_newton);
        if (n_setup > 0 && (n_newton == 0 || n_setup * RADAU_SETUP_DEN >= (n_setup + n_newton) * RADAU_SETUP_NUM)) {
            if (phase == PH_SETUP) {
                // ---- finish the attempt this lane came from -------------------------------------------------------
                if (ended == END_FAILED) {
                    h *= 0.5;
                } else if (ended == END_CONVERGED) {
                    HLM_RCOUNT(2);
                    double y_new[NEQ], ev[NEQ];
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) {
                        y_new[i] = y[i] + Z[2][i];
                        ev[i] = f0_(i) + (E[0] * Z[0][i] + E[1] * Z[1][i] + E[2] * Z[2][i]) * inv_h;
                    }
                    solve_r(ev);
                    double err = 0.0;
                    double tol[NEQ], inv_tol[NEQ];
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) tol[i] = A.atol + A.rtol * vmax_abs(y[i], y_new[i]);   // (fmax(|.|, |.|) as one instruction: hlm_models.h)
                    recip_all<MATH>(tol, inv_tol);
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) {
                        const double r = fabs(ev[i]) * inv_tol[i];
                        err = vmax(err, r);   // (r >= 0 or NaN: the bare v_max_f64 keeps err for a NaN r, as "if (r > err)" did)
                    }
                    if (!(err == err)) {
                        h *= 0.5;  // NaN estimate: treat like a failed Newton
                    } else {
                        const double root4 = inv_fourth_root<MATH>(err + 1e-16);
                        const double fac = A.safety * root4;
                        if (err <= 1.0) {
                            const double t_end = t_end_;
                            if (A.nq > 0) {   // (wave-uniform)
                            int next_q = next_q_;
                            double* const dense_row = dense_row_;
                            while (next_q < A.nq) {
                                const double tq = A.query[next_q];
                                if (!(tq <= t_end)) break;
                                const double th = (tq - t) * inv_h;
                                double* o = dense_row + (size_t)next_q * q_stride;
#pragma unroll
                                for (int c = 0; c < NEQ; ++c) {
                                    const int pc = pos(c);
                                    if (pc < 0) continue;  // component not selected
                                    double acc = y[c], thp = th;
#pragma unroll
                                    for (int m = 0; m < 3; ++m) {
                                        acc += (PD[0][m] * Z[0][c] + PD[1][m] * Z[1][c] + PD[2][m] * Z[2][c]) * thp;
                                        thp *= th;
                                    }
                                    o[(size_t)pc * c_stride] = acc;
                                }
                                ++next_q;
                            }
                            next_q_ = next_q;
                            }
#pragma unroll
                            for (int i = 0; i < NEQ; ++i) y[i] = y_new[i];
                            t = t_end;
                            ++n_accept_;
                            if (change_pending_) { h_change_ = h; change_pending_ = 0; }   // the first step the new sample accepted
                            h *= vclamp_uniform(fac, A.min_scale, A.max_scale);
                            h = vmax(h, h_ctl_);   // (h_ctl_ = the controller's proposal if the step was clipped, else 0: "if (clipped && h < h_ctl) h = h_ctl")
                        } else {
                            h *= vclamp_uniform(vmin(fac, 1.0), A.min_scale, A.max_scale);
                        }
                    }
                }
                ended = END_NONE;
                // ---- the next attempt -- of this system, or (in the section after this one) of the next one ----------
                const uint32_t n_attempt = n_attempt_;
                if (!(t < piece_end(piece_of(item_))) || n_attempt >= A.cap) {
                    phase = PH_FINISHED;   // (the end of the piece -- tf in the last one -- or the attempt cap: the outer loop takes it from here)
                } else {
                    HLM_RCOUNT(0);
                    double t_b = __builtin_inf();  // the next sample boundary of any stream
                    if constexpr (Model::NF > 0) {
                        bool moved = false, changed = false;
#pragma unroll
                        for (int j = 0; j < Model::NF; ++j) {
                            if (j < A.nforc) {
                                double tbj = tb_(j);
                                if (!(t < tbj)) {
                                    const bool held = tbj > -__builtin_inf();   // (else: the system's first attempt)
                                    const double was = F_(j), now = sample(j, tbj);
                                    F_(j) = now;
                                    tb_(j) = tbj;
                                    moved = true;
                                    changed = changed || (held && __double_as_longlong(now) != __double_as_longlong(was));   // (by bits, as the CPU restatement: a NaN sample that stays is no change)
                                }
                                t_b = vmin(t_b, tbj);
                            }
                        }
                        if (moved) FV = held_forcing();
                        if (changed) {
                            const double hc = h_change_;
                            if (hc > 0.0 && h > hc) h = hc;
                            change_pending_ = 1;
                        }
                    }
                    const double h_ctl = h;
                    double t_end = t + h;
                    bool clipped = false;
                    if (t_end > tf) { t_end = tf; clipped = true; }
                    if (t_end > t_b) { t_end = t_b; clipped = true; }
                    if (clipped) h = t_end - t;
                    h_ctl_ = clipped ? h_ctl : 0.0;
                    t_end_ = t_end;
                    n_attempt_ = n_attempt + 1u;
                    double f0[NEQ];

                    // f0 and the Jacobian: the model's exact one-sided derivative where it supplies one, else the
                    // reference's finite differences (approx_jacobian, radau_step_dense.cuh:13-31: 5 more RHS calls)
                    if constexpr (Model::HAS_RHS_JACOBIAN) {
                        Model::template rhs_jacobian<MATH>(P, FV, y, f0, Jm);  // both in one pass: shared sub-expressions once
                    } else if constexpr (Model::HAS_JACOBIAN) {
                        Model::template rhs<MATH>(P, FV, y, f0);
                        Model::template jacobian<MATH>(P, FV, y, Jm);
                    } else {
                        Model::template rhs<MATH>(P, FV, y, f0);
#pragma unroll
                        for (int j = 0; j < NEQ; ++j) {
                            double yp[NEQ], f1[NEQ];
                            const double d = FD_EPS * fmax(1.0, fabs(y[j]));
#pragma unroll
                            for (int i = 0; i < NEQ; ++i) yp[i] = y[i];
                            yp[j] += d;
                            Model::template rhs<MATH>(P, FV, yp, f1);
#pragma unroll
                            for (int i = 0; i < NEQ; ++i) Jm[i][j] = (f1[i] - f0[i]) / d;
                        }
                    }
                    // (g/h) I - J (Newton's real block and the error estimator) and ((a - ib)/h) I - J: LU with partial
                    // pivoting in general; for a lower-triangular J just the reciprocals of the diagonals
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) f0_(i) = f0[i];   // (for the error estimate, in the section that finishes this attempt)
                    inv_h = recip<MATH>(h);
                    g_h = MU_REAL * inv_h;
                    a_h = MU_CRE * inv_h;
                    b_h = MU_CIM * inv_h;
                    bool ok = true;
                    if constexpr (TRI) {
                        double den[2 * NEQ], inv_den[2 * NEQ], cr[NEQ];
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) {
                            den[i] = g_h - Jm[i][i];
                            cr[i] = a_h - Jm[i][i];
                            den[NEQ + i] = cr[i] * cr[i] + b_h * b_h;   // |(a/h - J_ii) - i b/h|^2
                            ok = ok && fabs(den[i]) > 1e-300;
                        }
                        recip_all<MATH>(den, inv_den);
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) {
                            inv_r[i] = inv_den[i];
                            inv_cr[i] = cr[i] * inv_den[NEQ + i];
                            inv_ci[i] = b_h * inv_den[NEQ + i];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NEQ; ++i)
#pragma unroll
                            for (int j = 0; j < NEQ; ++j) {
                                Mr[i][j] = ((i == j) ? g_h : 0.0) - Jm[i][j];
                                Cr[i][j] = ((i == j) ? a_h : 0.0) - Jm[i][j];
                                Ci[i][j] = (i == j) ? -b_h : 0.0;
                            }
                        ok = lu_real(Mr, piv_r, inv_r);
                        ok = lu_cplx(Cr, Ci, piv_c, inv_cr, inv_ci) && ok;
                    }
                    {
                        double sc[NEQ];   // the Newton increment's units
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) sc[i] = A.atol + A.rtol * fabs(y[i]);
                        recip_all<MATH>(sc, inv_sc);
                    }
                    // Simplified Newton on W = (TI (x) I) Z,  Z_i = Y_i - y, from Z = W = 0 -- and its FIRST iteration
                    // right here: with Z = 0 all three stage states are y, so the three RHS evaluations of that iteration
                    // are f0 (the same function of the same bits): the iteration costs two triangular solves and no RHS
                    // call, 1 of a lane's 2.5 iterations per attempt, and W, Z need no zeroing.
                    if (ok) {
                        double d0[NEQ], d1[NEQ], d2[NEQ];
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) {   // (TI (x) I) F with F_1 = F_2 = F_3 = f0: TI's row sums
                            d0[i] = (TI[0][0] + TI[0][1] + TI[0][2]) * f0[i];
                            d1[i] = (TI[1][0] + TI[1][1] + TI[1][2]) * f0[i];
                            d2[i] = (TI[2][0] + TI[2][1] + TI[2][2]) * f0[i];
                        }
                        solve_r(d0);
                        solve_c(d1, d2);
                        double nrm = 0.0, chk = 0.0;
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) {
                            if constexpr (MATH != MATH_FAST) chk += d0[i] + d1[i] + d2[i];
                            W[0][i] = d0[i];
                            W[1][i] = d1[i];
                            W[2][i] = d2[i];
                            const double r = vmax_abs2(vmax_abs(d0[i], d1[i]), d2[i]) * inv_sc[i];
                            nrm = vmax(nrm, r);
                            stage_increments<MATH>(W[0][i], W[1][i], W[2][i], Z[0][i], Z[1][i], Z[2][i]);
                            if constexpr (MATH == MATH_FAST) chk += Z[0][i];   // (see the Newton section)
                        }
                        HLM_RCOUNT(1);
                        prev = nrm;
                        it = 1;
                        if (!(chk - chk == 0.0)) ended = END_FAILED;          // a NaN / Inf increment
                        else if (nrm <= NEWTON_TOL) ended = END_CONVERGED;    // (stays in PH_SETUP: the next section finishes it)
                        else phase = PH_NEWTON;
                    } else {
                        ended = END_FAILED;  // a singular block: as a Newton that did not converge (h / 2, next section)
                    }
                }
            }
        }
        if (phase == PH_NEWTON) {
            HLM_RCOUNT(1);
            double Fs[NS][NEQ], d0[NEQ], d1[NEQ], d2[NEQ];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                double Y[NEQ];
#pragma unroll
                for (int i = 0; i < NEQ; ++i) Y[i] = y[i] + Z[s2][i];
                Model::template rhs<MATH>(P, FV, Y, Fs[s2]);
            }
#pragma unroll
            for (int i = 0; i < NEQ; ++i) {
                double r0, r1, r2;
                transformed_residuals<MATH>(Fs[0][i], Fs[1][i], Fs[2][i], r0, r1, r2);
                d0[i] = r0 - g_h * W[0][i];
                d1[i] = r1 - (a_h * W[1][i] + b_h * W[2][i]);
                d2[i] = r2 - (a_h * W[2][i] - b_h * W[1][i]);
            }
            solve_r(d0);
            solve_c(d1, d2);
            double nrm = 0.0, chk = 0.0;
#pragma unroll
            for (int i = 0; i < NEQ; ++i) {
                // the NaN / Inf guard of the increments.  FAST: on the sum of the first stage's new increments instead --
                // Z_1 = T_11 W_1 + T_12 W_2 + T_13 W_3 with three nonzero weights is finite only if all three W are, and
                // W = W + d is not finite as soon as a d is not: the same decision from 5 additions instead of 15
                if constexpr (MATH != MATH_FAST) chk += d0[i] + d1[i] + d2[i];
                W[0][i] += d0[i];
                W[1][i] += d1[i];
                W[2][i] += d2[i];
                const double r = vmax_abs2(vmax_abs(d0[i], d1[i]), d2[i]) * inv_sc[i];
                nrm = vmax(nrm, r);
                stage_increments<MATH>(W[0][i], W[1][i], W[2][i], Z[0][i], Z[1][i], Z[2][i]);
                if constexpr (MATH == MATH_FAST) chk += Z[0][i];
            }
            // the exits of the nested Newton loop, in its order
            int end = END_NONE;
            if (!(chk - chk == 0.0)) end = END_FAILED;                 // a NaN / Inf increment
            else if (it > 0 && nrm > 2.0 * prev) end = END_FAILED;     // diverging
            else {
                prev = nrm;
                if (nrm <= NEWTON_TOL) end = END_CONVERGED;
                else if (it + 1 >= NEWTON_MAXITER) end = END_FAILED;   // out of iterations
            }
            ++it;
            if (end != END_NONE) {
                ended = end;
                phase = PH_SETUP;
            }
        }
        const unsigned long long m_ended = __builtin_amdgcn_ballot_w64(phase == PH_FINISHED);
        if (m_ended != 0ull && (__builtin_popcountll(m_ended) >= HLM_RADAU_BREAK_LANES || ++ended_trips >= HLM_RADAU_BREAK_TRIPS)) break;
        // (a lane that waits for a record asks again every 16 trips)
        if (__builtin_amdgcn_ballot_w64(phase == PH_LOAD) != 0ull && (++polls_in & 15) == 0) break;
    }
    }
#undef f0_
#undef h_ctl_
#undef t_end_
#undef tb_
#undef F_
#undef fcol_
#undef dense_row_
#undef n_attempt_
#undef n_accept_
#undef next_q_
#undef item_
#undef h_change_
#undef change_pending_

}

}  // namespace hlm
