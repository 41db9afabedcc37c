// Dense output with wave-synchronous stores: the RK45 loop of hlm_rk45_kernel.h with the query rows of
// a wave written to HBM by all its 64 lanes TOGETHER, so that every store instruction of the dense
// output writes whole cache lines.
//
// Restates reference src/solver/rk45_kernel.cu:139-148 (the query loop inside an accepted step) and
// rk45_step_dense.cuh:171-244 (the interpolant); the host-order copy of rk45_api.hpp:255-267 is
// dense_tile_to_host() at the end of hlm_rk45_kernel.h.
//
// Why.  The dense output is the one HBM-bound part of the path: 40 bytes per (hillslope, query),
// 57.6 GB for 1 M hillslopes x 1 day at 1-minute stride.  With each lane storing its rows whenever it
// has them (hlm_rk45_kernel.h) a row store of a wave touches up to 64 different places: measured on
// MI355X (profiles/r02_dense_pmc.json) 133 write requests per 2560-byte row block where 40 would do,
// the texture cache's pending-request stall dominating: 18.3 ms = 3.2 TB/s.  Making the lanes emit each
// row in lockstep (one query cursor per wave, a lane holding its accepted step until all 64 have
// reached the query) brought the requests down to exactly 40 -- and cost 31 % more VALU work: the
// lanes of a wave take the same NUMBER of attempts (they share a forcing cell) but reject at slightly
// different attempts, and in lockstep every reject of any lane stalls the other 63 (measured lane
// utilisation 72 %; 19.5 ms).  So the lanes run free, as in hlm_rk45_kernel.h, and the ROWS are what
// waits: a lane puts each row it produces into an LDS ring of R rows ([row % R][comp][lane]: its own
// 8-byte column, no transposition, no bank conflict) and the wave flushes row q to HBM -- in the tile
// layout [tile][q][comp][lane], five stores of 512 contiguous bytes -- once every lane has produced
// it.  A lane only waits when it is R rows ahead of the slowest lane of its wave.  The committed step
// sequence of every lane, and therefore every output bit, is that of the free-running kernel.
#pragma once

#include "hlm_rk45_kernel.h"

namespace hlm {

// device layouts of the dense buffer (hlm_device_job.dense_tiled)
constexpr int DENSE_LINEAR = 0;     // [sys][q][comp]: the host-facing order
constexpr int DENSE_TILE_COMP = 2;  // [sys/64][q][comp][sys%64]: one 512-byte run per store instruction (this file)

// rows of slack between the lanes of a wave (LDS: R x 2560 bytes per wave)
#ifndef HLM_DENSE_RING_ROWS
#define HLM_DENSE_RING_ROWS 4
#endif

// SEL: output selection (hlm_device_job.dense_ncomp; the reference's output.states, data/config.yaml:44-52): a row
// holds the A.dense_k selected components only, [tile][q][k][lane].  Every lane still evaluates the whole
// interpolant (the row arithmetic is a tenth of an iteration) and puts component c into ring column
// A.dense_pos[c] -- an unselected one into the spare last column, so that no register array is indexed at run
// time; the flush reads and stores columns 0 .. dense_k-1.  SEL = false is the kernel without any of this.
// Flush of the ring (phase C of the loop below).  1: the number of complete rows is found first, with R independent
// ballots, and the rows are then moved in a software pipeline -- the LDS reads of row r + 1 are issued before the
// stores of row r, into a second set of registers (the attempt's temporaries are dead here: they cost nothing).
// 0: one row per trip, each trip = ballots -> branch -> LDS reads -> wait -> stores: every row paid a full LDS round
// trip and two VALU -> SALU -> branch chains (measured with the phase counters below: 490 cycles per flushed row).
#ifndef HLM_FLUSH_PIPELINED
#define HLM_FLUSH_PIPELINED 1
#endif
// Wave priority by phase.  1: the row and flush phases (B, C: LDS, stores, ballots -- short instructions whose
// latencies the wave waits for) run at s_setprio 1, the attempt (A: long FP64 chains) at 0, so that a wave in its
// rows is not queued behind its SIMD partner's arithmetic: -2 % (same-box A/B, round 3).  2: the reverse (+0.3 %).
#ifndef HLM_DENSE_PRIO
#define HLM_DENSE_PRIO 1
#endif
// 1: a ring slot is padded to the next power of two doubles (N_EQ = 5: 512 instead of 320; 18 instead of 12 KB of LDS
// per wave, eight waves per CU still fit): the slot offset of a row is a shift.  -0.5 % (same-box A/B, round 3).
// 1: the row loop of phase B written out (a lane emits at most R rows per iteration): no back-edge, the compiler
// if-converts the R trips into one predicated block in which the rows of a lane interleave.  -1.6 % (same-box A/B).
// (A wave-uniform exit between the trips, so that a trip no lane needs is skipped instead of predicated: +0.8 %.)
#ifndef HLM_ROWS_UNROLLED
#define HLM_ROWS_UNROLLED 1
#endif
#ifndef HLM_RING_POW2
#define HLM_RING_POW2 1
#endif
#ifndef HLM_DENSE_PRIO_LEVEL
#define HLM_DENSE_PRIO_LEVEL 1
#endif
// Experiment knobs (round 5; tools/build_variants.sh + tools/dense_variant_probe.py; the defaults are the product):
// HLM_EXTRA_SALU / HLM_EXTRA_VALU = n: n more scalar / vector instructions per attempt (results unchanged) -- the slope
// of the launch time against the instruction count of each kind, i.e. what a diet of as many would buy.
// HLM_SEL_KMAX = k > 0: the output-selecting instances hold rows of at most k components (ring slot of k + 1 columns
// instead of N_EQ) and are built for HLM_SEL_WAVES waves per SIMD.
#ifndef HLM_EXTRA_SALU
#define HLM_EXTRA_SALU 0
#endif
#ifndef HLM_EXTRA_VALU
#define HLM_EXTRA_VALU 0
#endif
#ifndef HLM_SEL_KMAX
#define HLM_SEL_KMAX 0
#endif
#ifndef HLM_SEL_WAVES
#define HLM_SEL_WAVES HLM_MIN_WAVES_PER_SIMD
#endif
#define HLM_STR2(x) #x
#define HLM_STR(x) HLM_STR2(x)

// rows r .. n-1 of a flush: `cur` holds row r (already read from the ring)
template <int r, int R, int N, class Load, class Store>
__device__ __forceinline__ void flush_pipeline(int n, int q0, const double (&cur)[N], Load& load, Store& store)
{
    if constexpr (r + 1 < R) {
        if (r + 1 < n) {  // (wave-uniform)
            double nxt[N];
            load(q0 + r + 1, nxt);
            store(q0 + r, cur);
            flush_pipeline<r + 1, R, N>(n, q0, nxt, load, store);
            return;
        }
    }
    store(q0 + r, cur);
}

#ifdef HLM_PHASE_TIMING
// Experiment builds only (tools/dense_phase_probe.py): shader-clock cycles every wave spends in the phases of the
// loop below, summed over all waves: [0] attempt, [1] rows + commit, [2] forcing refresh, [3] flush, [4] iterations,
// [5] waves.  Read back with hlm_debug_phase_cycles().
__device__ unsigned long long g_phase_cycles[8];
#define HLM_STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define HLM_STAMP(v)
#endif

// One tile (64 hillslopes, one wave) through the query window [qb, qe) of a buffer that holds the rows [bqb, bqe):
// the body of both dense kernels below.  QUEUED: the lane states come from and go to the record with device-coherent
// accesses (another wave of the same launch wrote or will read them: rk45_dense_queue_kernel).
template <class Model, int MATH, bool FIXED, bool RESUMABLE, int R, bool SEL, bool QUEUED>
__device__ __forceinline__ void dense_ring_tile(const Rk45Args& A, const unsigned blk, const int qb, const int qe, const int bqb, const int bqe,
                                                const int resume_in)
{
    using namespace dp;
    static_assert(R > 0 && (R & (R - 1)) == 0, "the ring slot of row q is q & (R - 1)");
    static_assert(RESUMABLE || !QUEUED, "sub-windows hand the lanes over through the record");
    constexpr int NEQ = Model::N_EQ, RESUME_F64 = resume_f64(NEQ);
    const int lane = (int)threadIdx.x;
    const int sys = (int)(blk * WAVE) + lane;
    if (sys >= A.ns) return;
    auto rload_f = [](const double* p) -> double { if constexpr (QUEUED) return coherent_load(p); else return *p; };
    auto rload_i = [](const int* p) -> int { if constexpr (QUEUED) return coherent_load(p); else return *p; };
    auto rstore_f = [](double* p, double v) { if constexpr (QUEUED) coherent_store(p, v); else *p = v; };
    auto rstore_i = [](int* p, int v) { if constexpr (QUEUED) coherent_store(p, v); else *p = v; };

    using lds_double_ptr = __attribute__((address_space(3))) double*;
    // doubles per ring slot: N_EQ x 64, or (HLM_RING_POW2) the next power of two, so that a slot's offset is a shift
    constexpr int RING_COLS = (SEL && HLM_SEL_KMAX > 0 && HLM_SEL_KMAX < NEQ) ? HLM_SEL_KMAX + 1 : NEQ;   // columns of a ring slot
#if HLM_RING_POW2
    constexpr int SLOT = (RING_COLS * WAVE <= 64) ? 64 : (RING_COLS * WAVE <= 128) ? 128 : (RING_COLS * WAVE <= 256) ? 256 : (RING_COLS * WAVE <= 512) ? 512 : 1024;
#else
    constexpr int SLOT = RING_COLS * WAVE;
#endif
    static_assert(SLOT >= RING_COLS * WAVE, "a ring slot holds one row of the wave");
    if (SEL && RING_COLS < NEQ && A.dense_k >= RING_COLS) return;   // (experiment builds: a wider selection does not fit this instance)
    __shared__ double ring_mem[R * SLOT];  // [row % R][comp][lane]
    __shared__ double lds_query[DENSE_QWIN];     // sliding window of query times (see rk45_multi_kernel)
    const lds_double_ptr ring = (lds_double_ptr)ring_mem + lane;

    const typename Model::Local P = Model::template load<MATH>(A.sp, sys);
    double y[NEQ];
#pragma unroll
    for (int i = 0; i < NEQ; ++i) y[i] = A.y0[(size_t)sys * NEQ + i];
    HeldForcing<Model, RESUMABLE> held;
    held.init(A, sys);

    double t = A.t0;
    double h = FIXED ? A.fixed_step : A.h0;
    const double tf = A.tf;
    int reject_count = 0;
    int status = HLM_SYS_OK;
    uint32_t n_attempt = 0, n_accept = 0;

    // Query window [qb, qe) of this launch (the whole range unless the host streams the dense output
    // in windows, hlm_rk45_integrate_streamed); the buffer holds exactly these rows, tile-major:
    // element (q, c) of this lane = row0[(q - qb) * 64 N_EQ + c * 64].
    const int q_hi = A.nq < qe ? A.nq : qe;
    [[maybe_unused]] const int K = SEL ? A.dense_k : NEQ;  // components per row
    const size_t Q_STRIDE = SEL ? (size_t)WAVE * (size_t)A.dense_k : (size_t)WAVE * NEQ;
    double* const row0 = A.dense + (size_t)blk * (size_t)(bqe - bqb) * Q_STRIDE + (size_t)lane;
    // ring column of component c (SEL: its position in the row, or the spare last column -- dense_k < N_EQ)
    [[maybe_unused]] int col[NEQ];
#pragma unroll
    for (int c = 0; c < NEQ; ++c) col[c] = SEL ? (A.dense_pos[c] < 0 ? RING_COLS - 1 : A.dense_pos[c]) * WAVE : c * WAVE;

    // live: still integrating in this launch.  Not live = reached tf, flagged, capped, paused at
    // t_stop ("stuck": it continues in the next launch), or finished in an earlier launch.
    bool live = true, finished_before = false;
    int next_q = qb;  // this lane's query cursor: it has produced (or owes nothing for) the rows below it
    int* const resume_i =
        RESUMABLE && A.resume ? reinterpret_cast<int*>(A.resume + (size_t)RESUME_F64 * A.ns) : nullptr;
    if (resume_i && resume_in) {
        const int st = rload_i(&resume_i[(size_t)4 * A.ns + sys]);
        next_q = rload_i(&resume_i[(size_t)3 * A.ns + sys]);
        if (next_q < qb) next_q = qb;  // (windows are visited in order; never index before the buffer)
        if (st != SYS_PAUSED) {
            live = false;
            finished_before = true;
            status = st;
            t = tf;  // whatever stopped it, it takes no part in the integration any more
        } else {
#pragma unroll
            for (int i = 0; i < NEQ; ++i) y[i] = rload_f(&A.resume[(size_t)i * A.ns + sys]);
            t = rload_f(&A.resume[(size_t)NEQ * A.ns + sys]);
            h = rload_f(&A.resume[(size_t)(NEQ + 1) * A.ns + sys]);
            reject_count = rload_i(&resume_i[(size_t)0 * A.ns + sys]);
            n_attempt = (uint32_t)rload_i(&resume_i[(size_t)1 * A.ns + sys]);
            n_accept = (uint32_t)rload_i(&resume_i[(size_t)2 * A.ns + sys]);
        }
    }
    const int my_start_q = next_q;  // rows below it were written by earlier launches: never touched here
    // the wave's flush cursor: rows below it are in HBM for every lane that owes them in this launch
    int q_flush = 0x7fffffff;
    for (unsigned long long m = __builtin_amdgcn_ballot_w64(true); m != 0ull; m &= m - 1ull) {
        const int v = __builtin_amdgcn_readlane(next_q, (int)__builtin_ctzll(m));
        q_flush = v < q_flush ? v : q_flush;
    }

    const double inf = __builtin_inf();
    int qwin_base = -(1 << 30);  // wave-uniform: first query index held in lds_query
    using lds_cdouble_ptr = const __attribute__((address_space(3))) double*;
    // query time q from the LDS window: valid for q in [q_flush, q_flush + R + 1] after the slide in (B)
    auto query_at = [&](int q) -> double { return ((lds_cdouble_ptr)lds_query)[q - qwin_base]; };
    double tq_next = next_q < A.nq ? A.query[next_q] : inf;

    bool have_k0 = false;  // see rk45_multi_kernel: k6 of an accepted step is the next k0
    // D1, (k0 as D2), D3, D4: the interpolant of a pending step (dense_coefficients; k0 is dead until the commit)
    double k0[NEQ], k6[NEQ], yt[NEQ], D1[NEQ], D3[NEQ], D4[NEQ];
#pragma unroll
    for (int i = 0; i < NEQ; ++i) k0[i] = k6[i] = yt[i] = D1[i] = D3[i] = D4[i] = 0.0;
    typename Model::Forc FV = Model::prep_forcing(P, held.F);
    // A PENDING step: accepted, not yet committed because rows of it are still to be produced (the
    // ring is full, or the launch's query window ends inside it).  While a lane waits its registers
    // are not written (it is masked off in the attempt code), so the step lives where the attempt
    // left it: y (old state), yt (new state), k6 (the next k0), D1, k0 (= D2), D3, D4 (the interpolant).
    // The lane's state, one integer in a VGPR (tested where it is needed; four booleans carried through the
    // loop's divergent control flow cost three scalar instructions at every join):
    //   LIVE    integrating;
    //   PENDING holding an accepted step (see above);
    //   STUCK   paused at t_stop: it continues in the next launch, which also owes its remaining rows;
    //   ZEROS   stopped for good (tf reached, flagged, capped, or finished in an earlier launch): its rows
    //           from next_q on are 0.
    constexpr int M_LIVE = 0, M_PENDING = 1, M_STUCK = 2, M_ZEROS = 3;
    int mode = live ? M_LIVE : M_ZEROS;
    double t1 = t, h_next = h;
    const double t_end = RESUMABLE ? fmin(tf, A.t_stop) : tf;

    // One loop, one back-edge, one exit.  Every iteration: (A) the lanes that can integrate do one
    // attempt, (B) lanes with rows to produce put them into the ring while it has room, committing
    // their step after its last row, (C) the wave flushes the rows every lane is through with.
    bool wave_on = true;  // wave-uniform
    if (live && t < t_end) held.refresh(A, P, t, FV);  // the samples of the first attempt (have_k0 is false)
    // Every load issued so far (parameters, state, resume record) lands HERE, not at its first use
    // inside the loop: gfx950 has one counter (vmcnt) for loads AND stores, so a wait for a prologue
    // load that the compiler places in the loop body would, on every iteration, also wait for the row
    // stores of the previous flush to be acknowledged by memory -- the stores must drain behind the
    // next attempt's arithmetic instead.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
#ifdef HLM_PHASE_TIMING
    unsigned long long ph_a = 0, ph_b = 0, ph_f = 0, ph_c = 0, ph_n = 0;
#endif
    // (Tried and dropped, round 3: "drain" iterations -- when the flush made room while some lane still holds rows of
    // its accepted step, an iteration that only moves rows, so that no lane falls an attempt behind: 816 instead of
    // 621 iterations per wave, 15.9 instead of 15.1 ms.)
    while (wave_on) {
        HLM_STAMP(st0);
#if HLM_DENSE_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif HLM_DENSE_PRIO == 2
        __builtin_amdgcn_s_setprio(1);
#endif
        int progress = 0;
        // ---- (A) one attempt ------------------------------------------------------------------
        const bool can_step = mode == M_LIVE && t < t_end && n_attempt < A.cap;
        if (mode == M_LIVE && !can_step) {
            if (t < t_end) status = HLM_SYS_CAPPED;  // attempt cap (the build's guard, see hlm_hip.h)
            mode = (status == HLM_SYS_OK && t < tf) ? M_STUCK : M_ZEROS;  // (paused at t_stop : tf reached or capped)
        }
        if (can_step) {
            progress = 1;
            if (t + h > tf) h = tf - t;  // rk45_kernel.cu:54
            ++n_attempt;  // (the forcing samples held for this t were fetched before the loop or in (B'))
#if HLM_EXTRA_SALU > 0
            { unsigned int dummy_s = 0; asm volatile(".rept " HLM_STR(HLM_EXTRA_SALU) "\n s_add_u32 %0, %0, 1\n .endr" : "+s"(dummy_s) : : "scc"); }
#endif
#if HLM_EXTRA_VALU > 0
            { unsigned int dummy_v = (unsigned)lane; asm volatile(".rept " HLM_STR(HLM_EXTRA_VALU) "\n v_add_u32 %0, %0, 1\n .endr" : "+v"(dummy_v)); }
#endif
            double k1[NEQ], k2[NEQ], k3[NEQ], k4[NEQ], k5[NEQ];
            [[maybe_unused]] double err = 0.0, jump = 0.0;
            attempt_on_cheapest_path<Model, MATH, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
            bool accepted = true;
            [[maybe_unused]] double fac = 1.0;
            if constexpr (!FIXED) {
                fac = A.safety * inv_fifth_root<MATH>(err + 1e-16);  // rk45_kernel.cu:151,156
                if (err <= 1.0) {
                    reject_count = 0;
                    if (jump > SLOPE_JUMP_THRESH) {  // rk45_kernel.cu:132-136
                        h = fmax(h * 0.5, A.h0 * MIN_STEP_FRACTION);
                        accepted = false;
                    }
                } else {  // rk45_kernel.cu:153-163
                    ++reject_count;
                    double f = fmin(fac, 1.0);
                    f = fmin(fmax(f, A.min_scale), A.max_scale);
                    h *= f;
                    if (reject_count > 5 || h < (tf - A.t0) * MIN_STEP_FRACTION) {
                        status = HLM_SYS_STIFF;
                        mode = M_ZEROS;
                    }
                    accepted = false;
                }
            }
            if (accepted) {
                // committed in (B), after the step's last row
                t1 = t + h;
                h_next = FIXED ? A.fixed_step : h * vclamp_uniform(fac, A.min_scale, A.max_scale);  // rk45_kernel.cu:151-152
                mode = M_PENDING;
                if (tq_next <= t1) dense_coefficients<NEQ>(y, yt, h, k0, k2, k3, k4, k5, k6, D1, k0, D3, D4);  // once per step
            }
        }
        const bool owes_zeros = mode == M_ZEROS;
        HLM_STAMP(st1);
#if HLM_DENSE_PRIO == 1
        __builtin_amdgcn_s_setprio(HLM_DENSE_PRIO_LEVEL);
#elif HLM_DENSE_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif

        // ---- (B) rows -------------------------------------------------------------------------
        // slide the query-time window (the lanes' cursors are within R rows of the flush cursor)
        if ((unsigned)(q_flush - qwin_base) >= (unsigned)(DENSE_QWIN - R - 2)) {
            const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
            const int nact = __builtin_popcountll(act);
            const int rank = __builtin_popcountll(act & ((1ull << lane) - 1ull));
            for (int i = rank; i < DENSE_QWIN; i += nact) lds_query[i] = (q_flush + i) < A.nq ? A.query[q_flush + i] : inf;
            qwin_base = q_flush;
        }
        // A lane has a row to produce when its pending step contains the next query (rk45_kernel.cu:139-146),
        // or when that query is not later than its current time: such a row is never produced by the
        // reference (rk45_kernel.cu:141: "if (tq > t)") and stays 0 -- as does every row of a lane that
        // has stopped for good, which it puts into the ring like any other (the flush then never has to
        // tell a produced row from an owed zero).
        bool pending = mode == M_PENDING;
        bool want = owes_zeros || (pending && tq_next <= t1) || tq_next <= t;
        [[maybe_unused]] const double inv_h = MATH == MATH_FAST ? fast_rcp(h) : 0.0;
        // one row per trip; ZEROS = false is the variant for a wave none of whose lanes produces a zero row
        // in this iteration (the usual case: no select, no inner branch)
        const int q_lim = q_flush + R < q_hi ? q_flush + R : q_hi;  // room in the ring, end of the launch's window
        auto produce_rows = [&](auto zeros_tag) {
            constexpr bool ZEROS = decltype(zeros_tag)::value;
#if HLM_ROWS_UNROLLED
            // a lane is never more than R rows ahead of the flush cursor: at most R trips, written out (no back-edge,
            // the rows of one lane interleave in the schedule)
#pragma unroll
            for (int trip = 0; trip < R; ++trip) {
                if (!(want && next_q < q_lim)) break;
#else
            while (want && next_q < q_lim) {
#endif
                const double tq_after = query_at(next_q + 1);  // fetched while the row is computed (the window holds inf past the last query)
                double row[NEQ] = {};
                if (!ZEROS || (pending && tq_next > t)) {  // rk45_step_dense.cuh:235-243
                    double th;
                    if constexpr (MATH == MATH_FAST) th = (tq_next - t) * inv_h; else th = (tq_next - t) / h;
                    const double th1 = 1.0 - th;
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) row[i] = dense_value(th, th1, y[i], D1[i], k0[i], D3[i], D4[i]);
                }
                const lds_double_ptr slot = ring + (next_q & (R - 1)) * SLOT;
#pragma unroll
                for (int c = 0; c < NEQ; ++c) {
                    if constexpr (SEL) slot[col[c]] = row[c]; else slot[c * WAVE] = row[c];
                }
                ++next_q;
                tq_next = tq_after;
                want = (ZEROS && owes_zeros) || (pending && tq_next <= t1) || (ZEROS && tq_next <= t);
                progress = 1;
            }
        };
        if (__builtin_amdgcn_ballot_w64(owes_zeros || tq_next <= t) == 0ull) produce_rows(std::false_type{});
        else produce_rows(std::true_type{});
        if (pending && !(tq_next <= t1)) {
            // the step's last row is out: commit it (the only place a step is committed)
#pragma unroll
            for (int i = 0; i < NEQ; ++i) { y[i] = yt[i]; k0[i] = k6[i]; }
            t = t1;
            h = h_next;
            ++n_accept;
            mode = M_LIVE;
            progress = 1;
        }

        HLM_STAMP(st2);
        // ---- (B') the forcing samples of the next attempt ----------------------------------------
        // Fetched here, BEFORE this iteration's row stores are issued: the wait for the load then
        // only covers the stores of the previous flush, which have had a whole iteration to drain
        // (one vmcnt for loads and stores, see above).  t only changes in the commit above, so this is
        // the refresh the next attempt would do (rk45_kernel.cu:62-75), same samples, same k0 handling.
        if (mode == M_LIVE && t < t_end) {
            if (held.refresh(A, P, t, FV)) have_k0 = false;
        }

        HLM_STAMP(st3);
        // ---- (C) flush: rows go to HBM once every lane is through with them -----------------------------
        bool flushed = false;
        [[maybe_unused]] const unsigned long long paused = __builtin_amdgcn_ballot_w64(mode == M_STUCK);  // they produce nothing more in this launch
#if HLM_FLUSH_PIPELINED
        {
            // Row q_flush + r - 1 (r = 1 .. R; a lane is never more than R rows ahead of the cursor) is complete when no
            // lane that still produces rows in this launch lacks it -- and somebody has it: when only paused lanes are
            // left, their rows belong to the next launch.
            // Both conditions are monotone in r (who lacks row r lacks row r + 1), so the complete rows are counted
            // without any chain between the R ballots.  A launch that is not RESUMABLE has no paused lanes.
            const int ahead = next_q - q_flush;
            int nready = 0;
            if constexpr (RESUMABLE) {
                const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
                for (int r = 1; r <= R; ++r) {
                    const unsigned long long lacks = __builtin_amdgcn_ballot_w64(ahead < r);
                    nready += ((lacks & ~paused) == 0ull && lacks != active) ? 1 : 0;
                }
            } else {
#pragma unroll
                for (int r = 1; r <= R; ++r) nready += __builtin_amdgcn_ballot_w64(ahead < r) == 0ull ? 1 : 0;
            }
            const int room = q_hi - q_flush;
            nready = nready < room ? nready : (room > 0 ? room : 0);
            if (nready > 0) {
                // every lane reads its column of a slot (one wait per row) and stores it: per component 64 lanes x 8 B =
                // 512 contiguous bytes
                auto load = [&](int q, double (&v)[NEQ]) {
                    const lds_double_ptr slot = ring + (q & (R - 1)) * SLOT;
#pragma unroll
                    for (int c = 0; c < NEQ; ++c)
                        if (!SEL || c < K) v[c] = slot[c * WAVE];
                };
                auto store = [&](int q, const double (&v)[NEQ]) {
                    // (a first launch owns every row of its window; a paused lane only the rows it produced)
                    if (!RESUMABLE || (next_q > q && q >= my_start_q)) {
                        double* o = row0 + (size_t)(q - bqb) * Q_STRIDE;
#pragma unroll
                        for (int c = 0; c < NEQ; ++c)
                            if (!SEL || c < K) o[c * WAVE] = v[c];
                    }
                };
                double first[NEQ];
                load(q_flush, first);
                flush_pipeline<0, R, NEQ>(nready, q_flush, first, load, store);
                q_flush += nready;
                flushed = true;
            }
        }
#else
        while (q_flush < q_hi) {
            const bool has_row = next_q > q_flush;
            const unsigned long long have = __builtin_amdgcn_ballot_w64(has_row);
            // somebody is still on its way to this row, or only paused lanes are left (their rows belong to the next launch)
            if (__builtin_amdgcn_ballot_w64(!has_row) & ~paused) break;
            if (have == 0ull) break;
            // every lane reads its column of the slot (five reads in flight, one wait) and stores it:
            // per component 64 lanes x 8 B = 512 contiguous bytes
            const lds_double_ptr slot = ring + (q_flush & (R - 1)) * SLOT;
            double v[NEQ];
#pragma unroll
            for (int c = 0; c < NEQ; ++c)
                if (!SEL || c < K) v[c] = slot[c * WAVE];
            if (has_row && (!RESUMABLE || q_flush >= my_start_q)) {  // (a first launch owns every row of its window)
                double* o = row0 + (size_t)(q_flush - bqb) * Q_STRIDE;
#pragma unroll
                for (int c = 0; c < NEQ; ++c)
                    if (!SEL || c < K) o[c * WAVE] = v[c];
            }
            ++q_flush;
            flushed = true;
        }
#endif
        wave_on = flushed || __builtin_amdgcn_ballot_w64(progress != 0) != 0ull;
#ifdef HLM_PHASE_TIMING
        {
            HLM_STAMP(st4);
            ph_a += st1 - st0; ph_b += st2 - st1; ph_f += st3 - st2; ph_c += st4 - st3; ++ph_n;
        }
#endif
    }
#ifdef HLM_PHASE_TIMING
    if (lane == (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {
        atomicAdd(&g_phase_cycles[0], ph_a); atomicAdd(&g_phase_cycles[1], ph_b); atomicAdd(&g_phase_cycles[2], ph_f);
        atomicAdd(&g_phase_cycles[3], ph_c); atomicAdd(&g_phase_cycles[4], ph_n); atomicAdd(&g_phase_cycles[5], 1ull);
    }
#endif

    if (mode == M_PENDING) {
        // the launch's query window ends inside this step: it is given back -- (y, t, h) are still the
        // step's start and the attempt is redone in the next launch, which then produces the rest of its rows
        --n_attempt;
        status = SYS_PAUSED;
    }
    if (RESUMABLE && mode == M_STUCK) status = SYS_PAUSED;  // stopped at t_stop
    if (resume_i) {
        // a lane that has stopped for good has (zero-)written every row the wave flushed
        rstore_i(&resume_i[(size_t)3 * A.ns + sys], (status != SYS_PAUSED && q_flush > next_q) ? q_flush : next_q);
        if (!finished_before) {
            rstore_i(&resume_i[(size_t)4 * A.ns + sys], status);
            if (status == SYS_PAUSED) {
#pragma unroll
                for (int i = 0; i < NEQ; ++i) rstore_f(&A.resume[(size_t)i * A.ns + sys], y[i]);
                rstore_f(&A.resume[(size_t)NEQ * A.ns + sys], t);
                rstore_f(&A.resume[(size_t)(NEQ + 1) * A.ns + sys], h);
                rstore_i(&resume_i[(size_t)0 * A.ns + sys], reject_count);
                rstore_i(&resume_i[(size_t)1 * A.ns + sys], (int)n_attempt);
                rstore_i(&resume_i[(size_t)2 * A.ns + sys], (int)n_accept);
            }
        }
    }
    if (status != SYS_PAUSED && !finished_before) {
        if (A.n_attempt) A.n_attempt[sys] = n_attempt;
        if (A.n_accept) A.n_accept[sys] = n_accept;
        const bool flagged = status != HLM_SYS_OK && t < tf;  // rk45_kernel.cu:167-170
        A.stiff[sys] = flagged ? status : HLM_SYS_OK;
#pragma unroll
        for (int i = 0; i < NEQ; ++i) A.y_final[(size_t)sys * NEQ + i] = flagged ? 0.0 : y[i];
        if (A.flag_state && flagged) record_flag_state<NEQ>(A.flag_state, sys, y, t);
    }
}

// One wave per tile for the launch's whole query window (grid = ceil(ns / 64)).
template <class Model, int MATH, bool FIXED, bool RESUMABLE, int R, bool SEL = false>
__global__ void __launch_bounds__(WAVE, SEL ? HLM_SEL_WAVES : HLM_MIN_WAVES_PER_SIMD) rk45_dense_ring_kernel(const Rk45Args A)
{
    // XCD-aware block -> hillslope-chunk map (see rk45_multi_kernel)
    const unsigned nblk = gridDim.x, orig = blockIdx.x;
    const unsigned xcd = orig & 7u, qd = nblk >> 3, rm = nblk & 7u;
    const unsigned blk = (xcd < rm ? xcd * (qd + 1u) : rm * (qd + 1u) + (xcd - rm) * qd) + (orig >> 3);
    dense_ring_tile<Model, MATH, FIXED, RESUMABLE, R, SEL, false>(A, blk, A.q_begin, A.q_end, A.q_begin, A.q_end, A.resume_in);
}

// The same launch as a PERSISTENT grid over (sub-window, tile) items (round 4).  The launch's query window is cut into
// nsub sub-windows and every resident wave keeps drawing the next item, sub-window-major, from a device counter; a
// lane pauses at the end of its sub-window exactly as it does at the end of a query window (the RESUMABLE protocol:
// same committed steps, same rows) and whichever wave draws the tile's next sub-window continues from the state
// record, read and written with device-coherent accesses.  Why: with one wave per tile the launch ends on waves that
// all run the window's length (1 M hillslopes: 7.6 rounds of 2 048 waves, 500 000: 3.8) and its last round is the
// emptier the fewer rounds there are -- 0.51 of the HBM spec at 1 M hillslopes per launch, 0.48 at 500 000, whatever
// the window's length (profiles/r04_dense_subwindows_probe.txt); with items a quarter of a window long the end of the
// launch is a quarter as ragged.  Item (s, tile) waits for (s - 1, tile) through a per-tile counter, as in
// rk45_queue_kernel: drawn earlier, so finished or running on a resident wave.
template <class Model, int MATH, bool FIXED, int R, bool SEL = false>
__global__ void __launch_bounds__(WAVE, HLM_MIN_WAVES_PER_SIMD) rk45_dense_queue_kernel(const Rk45Args A, unsigned int* __restrict__ queue, int nsub)
{
    const int lane = (int)threadIdx.x;
    const unsigned ntiles = ((unsigned)A.ns + WAVE - 1u) / WAVE, nitems = ntiles * (unsigned)nsub;
    unsigned int* const done = queue + 1;
    const long long Q0 = A.q_begin, QN = (long long)A.q_end - A.q_begin;
    // (no branch on the lane number anywhere in the protocol, and the loop is bounded: see rk45_queue_kernel)
    for (unsigned drawn = 0; drawn <= nitems; ++drawn) {
        const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)__hip_atomic_fetch_add(queue, lane == 0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (item >= nitems) break;
        const unsigned s = item / ntiles, tile = item - s * ntiles;
        bool lost = false;
        if (s > 0) {
            auto finished = [&]() { return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&done[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
            if (finished() < s) {
                const unsigned long long since = wall_clock64();
                while (finished() < s) {
                    __builtin_amdgcn_s_sleep(16);
                    const unsigned aborted = (unsigned)__builtin_amdgcn_readfirstlane(
                        (int)__hip_atomic_load(&done[ntiles], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (aborted != 0u || wall_clock64() - since > 2000000000ull) {   // 20 s: a lost wave; never seen
                        __hip_atomic_store(&done[ntiles], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        lost = true;
                        break;
                    }
                }
            }
            asm volatile("" ::: "memory");  // (the record is loaded after the counter, also as the compiler sees it)
        }
        const int qs = (int)(Q0 + QN * (long long)s / nsub), qe = (int)(Q0 + QN * (long long)(s + 1u) / nsub);
        if (lost) {
            // (watchdog) nobody continues this tile: its unfinished hillslopes come back flagged, with zero finals
            const int sys = (int)(tile * WAVE) + lane;
            int* const resume_i = reinterpret_cast<int*>(A.resume + (size_t)resume_f64(Model::N_EQ) * A.ns);
            if (sys < A.ns && coherent_load(&resume_i[(size_t)4 * A.ns + sys]) == SYS_PAUSED) {
                A.stiff[sys] = HLM_SYS_CAPPED;
#pragma unroll
                for (int i = 0; i < Model::N_EQ; ++i) A.y_final[(size_t)sys * Model::N_EQ + i] = 0.0;
                coherent_store(&resume_i[(size_t)4 * A.ns + sys], HLM_SYS_CAPPED);
            }
        } else {
            // The item's view of the launch arguments is READ AGAIN from the kernel-argument segment through a pointer
            // the compiler cannot see through: what the body derives from them then lives inside the item, instead of
            // being hoisted out of the item loop and held -- and spilled: 93 SGPRs, each back through a v_readlane --
            // across it.
            Rk45ArgsConst* kp = (Rk45ArgsConst*)__builtin_amdgcn_kernarg_segment_ptr();   // (A is the kernel's first argument: offset 0)
            asm volatile("" : "+s"(kp));
            const Rk45Args a = load_args(kp);
            dense_ring_tile<Model, MATH, FIXED, true, R, SEL, true>(a, tile, qs, qe, a.q_begin, a.q_end, (s > 0u || a.resume_in != 0) ? 1 : 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the rows and the record are out: publish
        __hip_atomic_store(&done[tile], s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (all lanes, one value)
    }
}

}  // namespace hlm
