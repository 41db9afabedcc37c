// The RK45 loop of hlm_rk45_kernel.h as a PERSISTENT grid over a queue of (time chunk, tile) items: the window
// [t0, tf] is cut into nchunk pieces, a tile is 64 consecutive hillslopes, and every resident wave keeps drawing the
// next item -- chunk-major, so item (c, tile) is drawn a whole chunk of tiles after (c - 1, tile) -- until the queue
// is empty.  Same integration, bit for bit (a lane pauses before the first attempt whose start time has reached the
// chunk's end and continues from that state: the protocol of the RESUMABLE launches, tested against the uninterrupted
// run); what changes is the SHAPE of the launch.
//
// Why.  One launch of the fixed map (rk45_multi_kernel: one wave per tile, grid = ceil(ns / 64)) drains at its end:
// 15 625 waves on 3 072 resident slots are 5.09 rounds of 31 ms waves, the waves of a round do not finish together
// (a rainy forcing cell costs 7 % more attempts than the mean), and the last stragglers run on a nearly empty
// machine.  Measured on MI355X (headline workload): 160.3 ms for 1 M hillslopes, but 150.5 ms per million in a
// 4 M launch (20 rounds amortise the same drain) -- 6 % of the 1 M launch is its tail.  With the window cut into
// pieces the slots never run dry before the last piece, and the drain is that of a wave 1/nchunk as long.
//
// Restates reference src/solver/rk45_kernel.cu:17-176 like rk45_multi_kernel (no dense output here: the dense
// kernels keep the fixed map).  Ordering between the pieces of one tile: a per-tile counter of finished pieces,
// stored after the state record and awaited by the wave that draws the next piece.  The record and the counter are
// read and written with agent-scope (device-coherent, sc1) loads and stores, the writer waits for its record stores
// (s_waitcnt) before it stores the counter, the reader loads the record after it has seen the counter: no cache
// write-back or invalidate anywhere.  (An agent-scope release / acquire FENCE pair does the same with two
// buffer_wbl2 + buffer_inv per item -- and the write-back flushes every dirty line of the XCD's L2, the scratch of
// all its waves included: 22.8 GB of HBM writes per headline launch, measured, and 3 % of its time.)  The awaited piece was drawn earlier (the queue is chunk-major), so it is finished or running on a
// resident wave: the wait cannot deadlock, and with more tiles than resident waves it never waits at all.
//
// RUNS (round 4).  What the queue hands out is a RUN of consecutive pieces of a tile, not a single piece: the wave
// that draws (run r, tile) integrates the pieces run_begin(r) .. run_begin(r + 1) - 1 of that tile one after the
// other with the lane states in its registers -- the lanes still pause at every piece's end (that is what re-aligns
// them in simulated time), but only the END of a run passes through the record in HBM.  Handing every piece over (round
// 3) re-read the 136-byte parameter record and moved 76 bytes of lane state out and back in 31 times per hillslope:
// 16.9 GB of HBM traffic per headline launch against 3.2 GB algorithmic.  What removes the launch's tail is the
// granularity of its LAST items only, so the runs shrink towards the end of the window (QueueRuns: by default each
// run takes half of what is left, at most a quarter and at least a sixteenth of the window; launches with no more tiles than resident waves
// keep handing every piece over: make_queue_runs in hlm_abi.hip has the measurements).  Same bits whatever the runs
// (hlm_device_job.queue_run_pieces, tests).
//
// GROUPS (round 4; hlm_device_job.queue_group_tiles = G > 1).  For workloads in which hillslopes drop out early
// (SURVEY's generator flags 42 % of them stiff; a flagged lane idles in its wave for the rest of the launch: 42 of 64
// lanes active).  The tiles are taken in GROUPS of G neighbours: item (piece, tile j) is pass j % G of group j / G --
// the wave reads the group's status words, packs the hillslopes that are still integrating into an LDS list (in their
// order, by ballot and popcount) and integrates the 64 hillslopes of ITS pass, or nothing when the group has fewer passes
// left.  All G passes of a (run, group) must build THE SAME list although they run at different times and the status
// words change under them (a sibling pass finishes or flags a hillslope): the list is therefore a function of the state
// at the START of the run -- a hillslope that stops integrating in run r stores its status with the run number in the
// upper bits (status | (r + 1) << 8), and a pass of run r lists every hillslope that is PAUSED or carries the tag of
// run r itself ("alive when the run began"; round-4 advisor finding: with the live status words a pass that came late
// saw a shorter list, and a hillslope could be skipped or integrated twice).  Nothing else reads the tag: every other
// reader of the record only asks "== SYS_PAUSED?".  A piece of a group starts when all G passes of its previous piece
// are through (the group's counter counts passes), so no status word of run r + 1 exists while a pass of run r lists.  Every piece is handed over through the record (runs of
// one piece), which this mode needs anyway; there is no global re-tiling, no extra launch and no drain, and a group
// of 8 tiles is one forcing cell's worth of neighbours, so the passes rarely mix forcing cells.  The results cannot depend on the
// packing: lane states travel by hillslope number and no RHS form is chosen by a neighbour's values.
//
// Every wave reaches the end of the grid whatever happens: the wait is bounded by a wall-clock watchdog
// (QUEUE_WAIT_LIMIT_TICKS of the 100 MHz real-time counter).  A wait that outlasts it -- a lost wave; never seen --
// gives the tile up: its unfinished hillslopes come back flagged HLM_SYS_CAPPED with zero finals, the channel of
// "stopped before tf by a resource limit", and an abort word makes every later wait give up at once.
#pragma once

#include <type_traits>

#include "hlm_rk45_kernel.h"

namespace hlm {

#ifdef HLM_QUEUE_MARKS
// experiment builds: progress marks of every wave in pinned HOST memory (tools/queue_hang_probe.py reads them while
// the kernel is still there): [block * 8 + k], k = 0 items drawn, 1 last item, 2 stage, 3 last counter seen, 4 polls
__device__ unsigned int* g_queue_marks = nullptr;
#define HLM_QMARK(k, v)                                                                                            \
    do {                                                                                                           \
        if (g_queue_marks && lane == 0)                                                                            \
            __hip_atomic_store(&g_queue_marks[blockIdx.x * 8 + (k)], (unsigned)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
    } while (0)
#else
#define HLM_QMARK(k, v) do { } while (0)
#endif
#ifdef HLM_QUEUE_PROF
// experiment builds: per-wave accounting of a queue launch in 100 MHz wall-clock ticks, [block * 8 + k]:
// 0 first item drawn at, 1 last item finished at, 2 ticks inside item bodies, 3 ticks waiting for a predecessor,
// 4 items drawn, 5 items with at least one running lane, 6 sum over items of the lanes that ran (high bits: trips with a
// forcing refresh), 7 trips of the attempt loop << 40 | sum of the lanes in them
__device__ unsigned long long* g_queue_prof = nullptr;
#define HLM_QPROF(k, expr)                                                                \
    do {                                                                                  \
        if (g_queue_prof && lane == 0) { unsigned long long& w_ = g_queue_prof[blockIdx.x * 8 + (k)]; w_ = (expr); } \
    } while (0)
#define HLM_QPROF_ADD(k, v) HLM_QPROF(k, w_ + (unsigned long long)(v))
#else
#define HLM_QPROF(k, expr) do { } while (0)
#define HLM_QPROF_ADD(k, v) do { } while (0)
#endif
#ifndef HLM_QUEUE_WAIT_LIMIT_TICKS
#define HLM_QUEUE_WAIT_LIMIT_TICKS 2000000000ull  // 20 s of wall_clock64() (100 MHz)
#endif
constexpr unsigned long long QUEUE_WAIT_LIMIT_TICKS = HLM_QUEUE_WAIT_LIMIT_TICKS;
constexpr int STATUS_TAG_SHIFT = 8;   // GROUPS: status | (run + 1) << STATUS_TAG_SHIFT (see below)

// The runs of a launch (kernel argument): run r covers the pieces [begin(r), begin(r + 1)).
constexpr int QUEUE_MAX_RUNS = 15;
constexpr int QUEUE_MAX_GROUP = 16;   // tiles per item at most (LDS: 4 bytes x 64 x 16 per wave)
struct QueueRuns {
    int nruns;
    int uniform;                      // > 0: every run has this many pieces (the last one what is left); else begin_[]
    int begin_[QUEUE_MAX_RUNS + 1];   // begin_[nruns] = the number of pieces
    // the pieces: piece c ends at t_begin + (c + 1) piece_len, the last one at min(tf, A.t_stop).  A plain launch cuts
    // [t0, tf]; a launch that integrates a WINDOW of it (time chunks of streamed forcings: lanes come from and go back
    // to the caller's state record, Rk45Args.resume / resume_in / t_stop) cuts the window
    double t_begin, piece_len;
    // tiles per item (hlm_device_job.queue_group_tiles; 1: an item is a tile).  G > 1: an item is a GROUP of G
    // neighbouring tiles whose hillslopes still integrating are packed into full passes of 64, piece by piece; item
    // (piece, tile j) is pass j % G of group j / G (see "GROUPS" below)
    int group;
    // test hook (HLM_QUEUE_STAGGER_TICKS in the environment of the launching process; 0 = off): pass p of a group waits
    // p x this many ticks of the 100 MHz clock before it builds its list, so that its sibling passes have finished
    // hillslopes by then -- the schedule the list rule above has to survive (tests/test_headline_kernel_gpu.py)
    unsigned int stagger_ticks;
    __host__ __device__ int begin(int r, int npieces) const
    {
        if (uniform > 0) { const long long b = (long long)r * uniform; return b < npieces ? (int)b : npieces; }
        return begin_[r];
    }
};

// queue[0]: next item; queue[1 + tile]: RUNS of `tile` finished so far; queue[1 + ntiles]: abort word.  The resume record (A.resume, SoA as in
// the RESUMABLE kernels: y[N_EQ], t, h | reject_count, n_attempt, n_accept, next_q, status) carries the lanes.
// the queue of a launch, zeroed by a kernel of the same stream
__global__ void queue_init_kernel(unsigned int* __restrict__ queue, unsigned int words)
{
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) queue[i] = 0u;
}

// RETILE: the instances that may change which hillslopes share a wave -- groups of tiles packed per piece
// (hlm_device_job.queue_group_tiles).  The plain instances (RETILE = false) carry none of that code: the headline
// kernel's register allocation is sensitive to it (+1 ... 5 % with it compiled in).
// COUNT: the diagnostic instances that also count lane-attempts per RHS form (Rk45Args.path_counts; LanePathCount).
// HYBRID: the instances whose attempt is attempt_per_lane_path_hybrid -- a wave that holds saturated lanes runs ONE
// block of general stages for all its lanes instead of a block per RHS form (hlm_device_job.queue_hybrid; same bits).
template <class Model, int MATH, bool FIXED, bool RETILE = false, bool COUNT = false, bool HYBRID = false>
__global__ void __launch_bounds__(WAVE, (MATH == MATH_FAST && !HYBRID) ? HLM_MIN_WAVES_PER_SIMD_LEAN : HLM_MIN_WAVES_PER_SIMD)
    rk45_queue_kernel(const Rk45Args A, unsigned int* __restrict__ queue, int nchunk, const QueueRuns R)
{
    using namespace dp;
    constexpr int NEQ = Model::N_EQ, RESUME_F64 = resume_f64(NEQ);
    const int lane = (int)threadIdx.x;
    const unsigned nsys = (unsigned)A.ns;
    const unsigned G = RETILE && R.group > 1 ? (unsigned)R.group : 1u;   // tiles per item
    const unsigned ntiles = ((nsys + WAVE - 1u) / WAVE + G - 1u) / G * G;   // items per run: the tiles (padded to whole groups)
    const unsigned nitems = ntiles * (unsigned)R.nruns;
    __shared__ int group_list[RETILE ? WAVE * QUEUE_MAX_GROUP : 1];   // (G > 1) the hillslopes of the item's group still integrating
    unsigned int* const done = queue + 1;
    int* const resume_i = reinterpret_cast<int*>(A.resume + (size_t)RESUME_F64 * A.ns);
    const double tf = A.tf;
    const double w_end = fmin(tf, A.t_stop);   // (t_stop = +inf unless the launch integrates a window)

    // Every step of the queue protocol below is written WITHOUT a branch on the lane number: all 64 lanes issue the
    // draw (lane 0 adds 1, the others 0: one atomic per wave after the compiler's wave-level combining), all poll the
    // tile's counter and all store it.  With `if (lane == 0)` around them the compiler threaded the end of one
    // iteration into the start of the next for the lanes != 0 (seen in the Dummy FAST instance: lane 0 left the
    // loop alone, nobody drew any more and the other lanes repeated item 0 for ever).  And whatever a compiler does
    // to this loop, a wave leaves it after nitems iterations at the latest.
    for (unsigned drawn = 0; drawn <= nitems; ++drawn) {
        const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)__hip_atomic_fetch_add(queue, lane == 0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        HLM_QMARK(1, item);
        HLM_QMARK(2, 1);
        if (item >= nitems) break;
        [[maybe_unused]] const unsigned long long prof_t0 = wall_clock64();
        if (drawn == 0) HLM_QPROF(0, prof_t0);
        HLM_QPROF_ADD(4, 1);
        const unsigned r = item / ntiles, tile = item - r * ntiles;   // run r of this tile:
        const unsigned c_begin = (unsigned)R.begin((int)r, nchunk), c_end = (unsigned)R.begin((int)r + 1, nchunk);   // its pieces
        bool lost = false;  // the watchdog fired: this tile is given up
        // (G > 1) item `tile` is pass `tile % G` of group `tile / G`: the group's counter counts finished passes, and
        // a piece of the group starts when all G passes of its previous piece are through (the packing of the next
        // piece needs every status word of the group)
        const unsigned dep = G > 1u ? tile / G : tile, dep_target = r * G;
        if (r > 0) {
            // the previous run of this tile: finished long ago when there are more tiles than resident waves
            auto finished = [&]() { return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&done[dep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
            if (finished() < dep_target) {
                const unsigned long long since = wall_clock64();
                HLM_QMARK(2, 2);
                [[maybe_unused]] unsigned polls = 0;
                while (finished() < dep_target) {
                    __builtin_amdgcn_s_sleep(16);
                    HLM_QMARK(4, ++polls);
                    const unsigned aborted = (unsigned)__builtin_amdgcn_readfirstlane(
                        (int)__hip_atomic_load(&done[ntiles], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (aborted != 0u || wall_clock64() - since > QUEUE_WAIT_LIMIT_TICKS) {
                        __hip_atomic_store(&done[ntiles], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        lost = true;
                        break;
                    }
                }
            }
            HLM_QMARK(2, 3);
            asm volatile("" ::: "memory");  // (the record is loaded after the counter, also as the compiler sees it)
        }
        [[maybe_unused]] const unsigned long long prof_t1 = wall_clock64();
        HLM_QPROF_ADD(3, prof_t1 - prof_t0);
        const bool from_record = r > 0 || A.resume_in != 0;   // else: this launch starts the hillslopes at (t0, y0)
        // the hillslopes of this item, 64 per pass: the tile's -- or (G > 1) the group's survivors, packed
        unsigned n_listed = WAVE;   // (G == 1: one pass over the tile's slots; the lanes filter themselves below)
        if (RETILE && G > 1u) {
            if (R.stagger_ticks != 0u && !lost) {   // (test hook: let the sibling passes get ahead)
                const long long until = (long long)wall_clock64() + (long long)(tile % G) * (long long)R.stagger_ticks;
                while ((long long)wall_clock64() < until) __builtin_amdgcn_s_sleep(32);
            }
            n_listed = 0u;
            const int run_tag = (int)(r + 1u) << STATUS_TAG_SHIFT;
            for (unsigned k = 0; k < G; ++k) {
                const unsigned slot = (dep * G + k) * WAVE + (unsigned)lane;
                const bool in_range = slot < nsys;
                const int s_k = in_range ? (int)slot : 0;
                // alive when this run began: still PAUSED, or stopped by a sibling pass of this very run
                bool alive = in_range;
                if (in_range && from_record) {
                    const int st = coherent_load(&resume_i[(size_t)4 * A.ns + s_k]);
#ifdef HLM_QUEUE_OLD_LIST
                    alive = st == SYS_PAUSED;   // (round 4's rule, kept only to show the stress test catches it)
#else
                    alive = st == SYS_PAUSED || (st & ~((1 << STATUS_TAG_SHIFT) - 1)) == run_tag;
#endif
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
                if (alive) group_list[n_listed + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = s_k;
                n_listed += (unsigned)__builtin_popcountll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the list is read by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        {
            const unsigned slot = (G > 1u ? tile % G : tile) * WAVE + (unsigned)lane;   // (G > 1: this item's pass of the packed list)
            const bool valid = G > 1u ? slot < n_listed : slot < nsys;
            const int sys = !valid ? 0 : (G > 1u ? group_list[slot] : (int)slot);
            if (valid && lost) {
                // (watchdog) whatever this hillslope's record says, nobody continues it: flagged, zero finals
                if (coherent_load(&resume_i[(size_t)4 * A.ns + sys]) == SYS_PAUSED) {
                    A.stiff[sys] = HLM_SYS_CAPPED;
                    if (A.n_attempt) A.n_attempt[sys] = 0u;
                    if (A.n_accept) A.n_accept[sys] = 0u;
#pragma unroll
                    for (int i = 0; i < NEQ; ++i) A.y_final[(size_t)sys * NEQ + i] = 0.0;
                }
                coherent_store(&resume_i[(size_t)4 * A.ns + sys], HLM_SYS_CAPPED | (G > 1u ? (int)(r + 1u) << STATUS_TAG_SHIFT : 0));
            } else if (valid) {
                int status = HLM_SYS_OK;
                bool runs = true;
                if (from_record) {
                    status = coherent_load(&resume_i[(size_t)4 * A.ns + sys]);
                    runs = status == SYS_PAUSED;  // else: finished or flagged in an earlier piece, its outputs were written then
                    if (runs) status = HLM_SYS_OK;
                }
#ifdef HLM_QUEUE_PROF
                {
                    const int nrun = __builtin_popcountll(__builtin_amdgcn_ballot_w64(runs));
                    if (runs && nrun > 0 && g_queue_prof && (int)__builtin_amdgcn_readfirstlane(lane) == lane) {
                        g_queue_prof[blockIdx.x * 8 + 5] += 1ull;
                        g_queue_prof[blockIdx.x * 8 + 6] += (unsigned long long)nrun;
                    }
                }
#endif
                if (runs) {
                    const typename Model::Local P = Model::template load<MATH>(A.sp, sys);
                    double y[NEQ];
                    double t = A.t0;
                    double h = FIXED ? A.fixed_step : A.h0;
                    int reject_count = 0;
                    uint32_t n_attempt = 0, n_accept = 0;
                    if (!from_record) {
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) y[i] = A.y0[(size_t)sys * NEQ + i];
                    } else {
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) y[i] = coherent_load(&A.resume[(size_t)i * A.ns + sys]);
                        t = coherent_load(&A.resume[(size_t)NEQ * A.ns + sys]);
                        h = coherent_load(&A.resume[(size_t)(NEQ + 1) * A.ns + sys]);
                        reject_count = coherent_load(&resume_i[(size_t)0 * A.ns + sys]);
                        n_attempt = (uint32_t)coherent_load(&resume_i[(size_t)1 * A.ns + sys]);
                        n_accept = (uint32_t)coherent_load(&resume_i[(size_t)2 * A.ns + sys]);
                    }
                    HeldForcing<Model, false> held;  // the samples of the first attempt are fetched at its t (as a resumed lane does)
                    held.init(A, sys);
                    typename Model::Forc FV = Model::prep_forcing(P, held.F);
                    bool have_k0 = false;  // k0 = rhs(t, y, F) is a pure function: recomputed, it has the bits of the carried one
                    double k0[NEQ];
                    [[maybe_unused]] typename std::conditional<COUNT, LanePathCount, int>::type pc{};
                    [[maybe_unused]] bool wave_sat = false;   // (HYBRID; wave-uniform) a lane of this wave needed the general RHS last time

                    // ---- the loop of rk45_multi_kernel, piece by piece --------------------------------------------
                    // lanes pause the first time t reaches the end of their piece (the last piece runs to tf) and go on
                    // together with the next piece of the run: the protocol of a hand-over, without the hand-over
                    double t_end = w_end;
                    for (unsigned c = c_begin; c < c_end; ++c) {
                        t_end = c + 1u < (unsigned)nchunk ? fmin(w_end, R.t_begin + (double)(c + 1u) * R.piece_len) : w_end;
                        while (t < t_end && status == HLM_SYS_OK && n_attempt < A.cap) {
#ifdef HLM_QUEUE_PROF
                            {   // [7]: trips of the attempt loop << 40 | lanes that took part in them; [6] high bits: trips in which a lane's held sample expired
                                const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true), x_ = __builtin_amdgcn_ballot_w64(!(t < held.tb_next));
                                if (g_queue_prof && lane == (int)__builtin_ctzll(m_)) {
                                    g_queue_prof[blockIdx.x * 8 + 7] += (1ull << 40) + (unsigned long long)__builtin_popcountll(m_);
                                    if (x_ != 0ull) g_queue_prof[blockIdx.x * 8 + 6] += 1ull << 40;
                                }
                            }
#endif
                            // rk45_kernel.cu:54 "if (t + h > tf) h = tf - t": only the last steps of the window are clipped, so the
                            // subtraction and the two selects wait behind a wave-uniform test (add + compare per attempt otherwise)
                            if (__builtin_amdgcn_ballot_w64(t + h > tf) != 0ull) {
                                asm volatile("");   // (keeps this a branch: the compiler otherwise turns it back into the selects)
                                if (t + h > tf) h = tf - t;
                            }
                            if (held.refresh(A, P, t, FV)) have_k0 = false;
                            ++n_attempt;
                            double k1[NEQ], k2[NEQ], k3[NEQ], k4[NEQ], k5[NEQ], k6[NEQ], yt[NEQ];
                            [[maybe_unused]] double err = 0.0, jump = 0.0;
                            if constexpr (COUNT) attempt_per_lane_path_counted<Model, MATH, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump, pc);
                            else if constexpr (HYBRID) attempt_per_lane_path_hybrid<Model, MATH, FIXED>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump, wave_sat);
                            else attempt_on_cheapest_path<Model, MATH, FIXED, false>(P, FV, A, y, h, have_k0, k0, k1, k2, k3, k4, k5, k6, yt, err, jump);
                            [[maybe_unused]] double fac = 1.0;
                            if constexpr (!FIXED) {
                                fac = A.safety * inv_fifth_root<MATH>(err + 1e-16);  // rk45_kernel.cu:151,156
                                if (err <= 1.0) {
                                    reject_count = 0;
                                    if (jump > SLOPE_JUMP_THRESH) {  // rk45_kernel.cu:132-136
                                        h = fmax(h * 0.5, A.h0 * MIN_STEP_FRACTION);
                                        continue;
                                    }
                                } else {  // rk45_kernel.cu:153-163
                                    ++reject_count;
                                    double f = fmin(fac, 1.0);
                                    f = fmin(fmax(f, A.min_scale), A.max_scale);
                                    h *= f;
                                    if (reject_count > 5 || h < (tf - A.t0) * MIN_STEP_FRACTION) status = HLM_SYS_STIFF;
                                    continue;
                                }
                            }
#pragma unroll
                            for (int i = 0; i < NEQ; ++i) {
                                y[i] = yt[i];
                                k0[i] = k6[i];  // = rhs(t1, y_new, F): next attempt's k0 unless the forcing moves
                            }
                            t += h;
                            ++n_accept;
                            if constexpr (FIXED) h = A.fixed_step;
                            else h *= vclamp_uniform(fac, A.min_scale, A.max_scale);  // fmin(fmax(fac, minScale), maxScale), rk45_kernel.cu:151-152
                        }
                    }

                    HLM_QMARK(2, 4);
                    if constexpr (COUNT) pc.flush(A.path_counts);
                    if (status == HLM_SYS_OK && t < t_end && n_attempt >= A.cap) status = HLM_SYS_CAPPED;  // attempt cap
                    if (status == HLM_SYS_OK && t < tf) {
                        // paused at the end of its run's last piece: the next run continues exactly here
                        coherent_store(&resume_i[(size_t)4 * A.ns + sys], SYS_PAUSED);
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) coherent_store(&A.resume[(size_t)i * A.ns + sys], y[i]);
                        coherent_store(&A.resume[(size_t)NEQ * A.ns + sys], t);
                        coherent_store(&A.resume[(size_t)(NEQ + 1) * A.ns + sys], h);
                        coherent_store(&resume_i[(size_t)0 * A.ns + sys], reject_count);
                        coherent_store(&resume_i[(size_t)1 * A.ns + sys], (int)n_attempt);
                        coherent_store(&resume_i[(size_t)2 * A.ns + sys], (int)n_accept);
                    } else {
                        // (G > 1: with the run's tag -- the sibling passes of this run still list the hillslope)
                        coherent_store(&resume_i[(size_t)4 * A.ns + sys], RETILE ? (status | (G > 1u ? (int)(r + 1u) << STATUS_TAG_SHIFT : 0)) : status);
                        if (A.n_attempt) A.n_attempt[sys] = n_attempt;
                        if (A.n_accept) A.n_accept[sys] = n_accept;
                        const bool flagged = status != HLM_SYS_OK && t < tf;  // rk45_kernel.cu:167-170
                        A.stiff[sys] = flagged ? status : HLM_SYS_OK;
#pragma unroll
                        for (int i = 0; i < NEQ; ++i) A.y_final[(size_t)sys * NEQ + i] = flagged ? 0.0 : y[i];
                        if (A.flag_state && flagged) record_flag_state<NEQ>(A.flag_state, sys, y, t);
                    }
                }
            }
        }
        // publish the run: the state record's stores complete (acknowledged at device scope), then the tile's counter
        HLM_QMARK(2, 5);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        HLM_QMARK(2, 6);
        HLM_QPROF_ADD(2, wall_clock64() - prof_t1);
        HLM_QPROF(1, wall_clock64());
        if (G > 1u) (void)__hip_atomic_fetch_add(&done[dep], lane == 0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one more pass of the group
        else __hip_atomic_store(&done[tile], r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (all lanes, one value)
        HLM_QMARK(2, 7);
    }
}

}  // namespace hlm
