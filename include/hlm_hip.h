/*
 * hlm_hip.h -- C ABI of libhlm_hip.so, the MI355X (gfx950) implementation of the
 * per-hillslope adaptive Dormand-Prince RK45 path of PrincetonUniversity/Tiger_HLM_GPU.
 *
 * Plain C: pointers, sizes, POD structs; no C++/torch types cross this boundary.
 * Every entry point returns 0 on success and a non-zero hlm_status otherwise;
 * hlm_last_error() then holds a message (thread-local).  Nothing throws.
 * Numerical trouble is DATA (stiff[] flags), never an error -- as in the reference.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * reference's src/).  The header-only C++ mirror of the reference's template API
 * (rk45_api::setModelParameters<M>, run_rk45<M>, ...) lives in include/hlm/ and
 * forwards to these functions.
 *
 * Threading: like the reference (global devParams + forcing symbols) the model registry is
 * process-global state and the forcing registration is per HIP device (the current device of
 * the calling thread); one host-API integration may be in flight per device.
 * hlm_rk45_integrate_device() itself is re-entrant (all state is in the job struct).
 */
#ifndef HLM_HIP_H
#define HLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HLM_ABI_VERSION 5
#define HLM_NEQ 5           /* Model204::N_EQ, models/model_204.hpp:19 (and the DummyModel's) */
#define HLM_MAX_NEQ 8       /* largest N_EQ a registered model may have (kernels are templates over it) */
#define HLM_MAX_FORCINGS 16 /* I_O/forcing_data.h:5 */

#define HLM_UID_DUMMY 0 /* DummyModel (README.md:41-42, rhs: model_dummy_python.ipynb:65-89) */
#define HLM_UID_204 204 /* Model204::UID, models/model_204.hpp:18 */
#define HLM_UID_TOY3 3  /* ToyModel3: N_EQ = 3, one forcing -- the build's "Add a New Model" example
                         * (README.md plug-in section; include/hlm/models/model_toy3.hpp) */

typedef enum {
    HLM_SUCCESS = 0,
    HLM_ERR_INVALID_ARGUMENT = 1,
    HLM_ERR_UNKNOWN_MODEL = 2,
    HLM_ERR_HIP = 3,        /* a HIP runtime call failed (rk45_api.hpp:86-108,150-153) */
    HLM_ERR_NO_DEVICE = 4,
    HLM_ERR_STATE = 5       /* e.g. launch before setup / forcings not registered */
} hlm_status;

/* per-system status in stiff[]: 0/1 as rk45_kernel.cu:160-170; 2 = attempt cap reached
 * (build's guard against the slope-jump livelock of rk45_kernel.cu:133-136). */
#define HLM_SYS_OK 0
#define HLM_SYS_STIFF 1
#define HLM_SYS_CAPPED 2

/* Math mode.  1 = FAST, the DEFAULT of the host API: divisions by per-system constants become
 * multiplications by their reciprocals, x^(2/3) and x^(-1/5) are evaluated by division-free series
 * from integer / f32 seeds (<= 8e-16 relative error over the whole double range), the error norm
 * uses one reciprocal.
 * 0 = STRICT: IEEE divides and the correctly rounded values of pow(x, 2.0/3.0) and pow(u, 0.2) -- the
 * double exponents the reference spells; what glibc's pow returns in 999 of 1 000 cases
 * (csrc/hlm_strict_pow.h) -- in the reference's own operation structure (FMA-contracted like nvcc's
 * default); 2.2x slower.  STRICT exists for two behaviours FAST does not have:
 *  (a) a NaN / Inf state component is carried along silently, as in the reference (its error norm
 *      skips NaN ratios); in FAST it poisons the error norm, every attempt is rejected and the
 *      system is flagged (stiff = 1, zero rows) after six rejects;
 *  (b) any atol >= 0 is accepted; FAST needs 1e-60 <= atol <= 1e50 (its error norm multiplies the
 *      five tolerances).  Until hlm_set_math_mode() is called the host API picks FAST where (b)
 *      holds and STRICT otherwise; an explicit hlm_device_job.math_mode is never overridden.
 * Neither mode reproduces libdevice's pow bit for bit; both meet the stated parity tolerance. */
#define HLM_MATH_STRICT 0
#define HLM_MATH_FAST 1

/* I_O/parameters_loader.hpp:19-37: 2 x long + 15 x double = 136 B, field order is ABI
 * (the reference memcpy's it to the device, main.cpp:400, and MPI_Sends it raw, :303). */
typedef struct hlm_spatial_params {
    long stream;
    long next_stream;
    double c1, infil, perco, Hu, lat, sw, ss, n_mann, slope, L, A_h, alpha3, alpha4,
        melt_f, temp_thr;
} hlm_spatial_params;

/* ---- library / device ------------------------------------------------------------ */
int hlm_abi_version(void);
const char* hlm_last_error(void);
int hlm_device_count(int* count);
int hlm_set_device(int device); /* main.cpp:314-319 cudaSetDevice */
int hlm_device_synchronize(void);

/* Device memory the CALLER owns, as main.cpp:392-404 (d_sp) and :552-557 (d_forc_ptr)
 * do with cudaMalloc/cudaMemcpy. */
int hlm_device_malloc(void** dptr, size_t bytes);
int hlm_device_free(void* dptr);
int hlm_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int hlm_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);

/* ---- model registry (model_registry.hpp:12-13, model_registry.cpp:18-60) ---------- */
/* p = {initialStep, rtol, atol, safety, minScale, maxScale} (model_204.hpp:22-30).
 * Replaces cudaMemcpyToSymbol(devParams). */
int hlm_set_model_parameters(int uid, const double p[6]);
int hlm_get_model_parameters(int uid, double p[6]);
/* Build features (not in the reference): fixed-step mode (h > 0 turns the controller
 * off; 0 restores adaptive), per-system attempt cap (0 = default 2^24), math mode. */
int hlm_set_fixed_step(int uid, double h);
int hlm_set_attempt_cap(uint32_t cap);
int hlm_set_math_mode(int mode);
/* hlm_device_job.time_pieces of the host API's launches without dense output: 0 / 1 = one wave per tile; negative
 * (the default) = automatic: from 4 096 hillslopes on, one piece per twelve simulated hours, at most 32, none
 * when that makes fewer than 8 (a 4-day window is the shortest that is cut). */
int hlm_set_time_pieces(int pieces);
/* (ABI 4) hlm_device_job.queue_run_pieces and queue_group_tiles of the host API's launches in time pieces (see the job's
 * fields): what the queue hands out -- 0 = automatic, n = runs of n pieces, -n = automatic with runs of at most n -- and
 * G > 1 = the hillslopes of G neighbouring tiles that are still integrating packed into full waves (domains that lose
 * hillslopes to the stiff flag: SURVEY's cold C3 runs its RK45 pass in 99 ms with (0, 8) against 118 ms with (0, 0)).
 * Results do not depend on either. */
int hlm_set_queue_shape(int run_pieces, int group_tiles);
/* (ABI 5) hlm_device_job.queue_hybrid of the host API's launches in time pieces: 1 = waves that hold saturated hillslopes
 * run one block of general stages for all their lanes (domains whose static stores the storms fill: -14 ... -19 %;
 * +18 % on domains where they never overflow: two waves per SIMD instead of three), -1 / 0 = a block per RHS form. */
int hlm_set_queue_hybrid(int mode);
/* 1: hlm_launch_rk45_kernel / hlm_run_rk45 follow the RK45 pass with the Radau-IIA pass below
 * (what rk45_api.hpp:189-247 attempts inside retrieve_and_free), every flagged system restarting at t0 as
 * radau_kernel.cu:44-52 does; 2: the same pass CONTINUING from the state each flagged system had reached
 * (hlm_device_job.d_flag_state); 0 (default): flagged systems keep stiff = 1 and zero rows, the reference's
 * observable behaviour. */
int hlm_set_radau_fallback(int enabled);
/* (ABI 5) The Radau pass's work queue hands out (time piece, flagged system) items: pieces = 1 ... 16 equal parts of
 * [t0, tf] (1: whole systems), 0 (default) = automatic -- 16 when the flagged systems restart at t0; when they continue
 * from their flag states, whole systems up to the last six of 16 pieces and those one by one ("continue" mode in
 * pieces always takes 16, the bins its list is ordered by).  A system pauses before the first
 * implicit attempt that starts at or after its piece's end and goes on from that state in the next one: no step is clipped to
 * a piece, results do not depend on the count.  (Why: hlm_radau_kernel.h "TIME PIECES".) */
int hlm_set_radau_pieces(int pieces);
/* Output selection of the host API: the reference's run configuration lists the state variables to write
 * (data/config.yaml:44-52 "output.states", parsed by I_O/config_loader.cpp:19-84; the writers take the variable
 * list, I_O/output_series.cpp:18-72).  n > 0: dense rows carry only the n components comps[0..n) (ascending,
 * distinct, each < the model's N_EQ -- checked when buffers are set up for a model): hlm_setup_gpu_buffers*
 * sizes d_dense_all for n doubles per row, the kernels store only those components (n/N_EQ of the HBM and PCIe
 * bytes), hlm_retrieve_and_free / hlm_run_rk45 deliver h_dense [ns][nq][n].  n = 0 (default): all components.
 * Final states are always complete (they are the next run's initial state).  "output.print_interval" needs no
 * switch: it is the spacing of the query times the caller passes. */
int hlm_set_output_states(int n, const int* comps);

/* ---- forcing symbols (I_O/forcing_data.cu:5-10; uploads at main.cpp:559-574) -------
 * d_forc: DEVICE pointer, float [sum_j nT[j]][ld] forcing-major then time then column
 * (main.cpp:527-549 with ld = num_systems and column = system).  dt_hours/nT: HOST
 * arrays of length nforc.  The caller keeps ownership of d_forc.
 * d_col (nullable DEVICE int32 [num_systems]) selects a column per system so that a
 * compact per-grid-cell table can be used instead of the per-stream expansion; NULL
 * means column = system (the reference layout). */
int hlm_set_forcings(const float* d_forc, const double* dt_hours, const uint64_t* nT,
                     int nforc, int64_t ld, const int32_t* d_col);
/* Alternative registration for the host API: the forcing series stay in HOST memory
 * (h_tables[j]: float [nT[j]][ld]; the pointers must stay valid until hlm_clear_forcings) and
 * hlm_launch_rk45_kernel / hlm_run_rk45 stream them through HBM in chunks of chunk_minutes
 * (hlm_rk45_integrate_forcing_streamed; chunk_minutes < 0: its growing chunks). */
int hlm_set_forcings_host(const float* const* h_tables, const double* dt_hours, const uint64_t* nT,
                          int nforc, int64_t ld, const int32_t* d_col, double chunk_minutes);
int hlm_clear_forcings(void);

/* ---- three-phase host API (rk45_api.hpp:63-116, :120-154, :159-270) --------------- */
typedef struct hlm_gpu_buffers {
    double* d_y0_all;      /* [ns][neq] */
    double* d_y_final_all; /* [ns][neq] */
    double* d_query_times; /* [nq] */
    double* d_dense_all;   /* dense rows in the device layout named by dense_layout (below) */
    int32_t* d_stiff;      /* [ns] */
    uint32_t* d_n_attempt; /* [ns] attempted RK45 steps (build's step counter) */
    uint32_t* d_n_accept;  /* [ns] */
    int num_systems;
    int num_queries;
    int neq;               /* N_EQ of the model the buffers were sized for (0 in a caller-filled struct = 5) */
    int dense_layout;      /* an hlm_device_job.dense_tiled value; hlm_setup_gpu_buffers chooses 2: tile-major
                            * [ceil(ns/64)][nq][neq][64] (the reference's device layout is [ns][neq][nq],
                            * rk45_kernel.cu:145, and is likewise re-ordered on the way to the host) */
    int dense_ncomp;       /* output selection the buffers were sized for (hlm_set_output_states): 0 = all neq
                            * components, K > 0 = dense rows hold the K components dense_comp[0..K) only */
    int dense_comp[HLM_MAX_NEQ];
} hlm_gpu_buffers;

/* setup_gpu_buffers<M>: allocate the device buffers, copy h_y0 ([ns][M::N_EQ]) and h_query_times.
 * The template parameter of the reference is the uid here; hlm_setup_gpu_buffers is the N_EQ = 5 form. */
int hlm_setup_gpu_buffers_model(int uid, const double* h_y0, int num_systems, const double* h_query_times,
                                int num_queries, hlm_gpu_buffers* out);
int hlm_setup_gpu_buffers(const double* h_y0, int num_systems, const double* h_query_times,
                          int num_queries, hlm_gpu_buffers* out);
/* launch_rk45_kernel<M>: integrate all systems on the current device and synchronise.
 * d_sp: DEVICE pointer to num_systems hlm_spatial_params (may be NULL for the
 * DummyModel).  Uses the registered parameters and forcings. */
int hlm_launch_rk45_kernel(int uid, const hlm_gpu_buffers* b, double t0, double tf,
                           const void* d_sp);
/* retrieve_and_free<M>: D2H of final states (h_final [ns][neq]) and dense output
 * (h_dense [ns][nq][neq] = index (s*nq+q)*neq+c, rk45_api.hpp:255-267: the device layout is
 * transposed on the way, by a kernel writing pinned host memory directly or through a device
 * staging buffer), then free.
 * h_stiff / h_n_attempt / h_n_accept are optional (NULL to skip).  The Radau relaunch the
 * reference attempts in here (rk45_api.hpp:189-247, after its copies) runs in
 * hlm_launch_rk45_kernel instead, and only when hlm_set_radau_fallback(1) was called; by default
 * flagged systems keep zero rows, exactly what the reference's host receives. */
int hlm_retrieve_and_free(hlm_gpu_buffers* b, double* h_final, double* h_dense,
                          int32_t* h_stiff, uint32_t* h_n_attempt, uint32_t* h_n_accept);
/* run_rk45<M> (rk45_api.hpp:273-313): the three phases composed. */
int hlm_run_rk45(int uid, const double* h_y0, int num_systems, double t0, double tf,
                 const double* h_query_times, int num_queries, const void* d_sp,
                 double* h_final, double* h_dense, int32_t* h_stiff,
                 uint32_t* h_n_attempt, uint32_t* h_n_accept);

/* ---- device-resident core (everything above is built on this) --------------------- */
typedef struct hlm_device_job {
    int uid;
    int num_systems;
    int num_queries;
    double t0, tf;
    /* inputs, all DEVICE pointers */
    const double* d_y0;      /* [ns][5] AoS (rk45_kernel.cu:45) */
    const double* d_query;   /* [nq] ascending */
    const void* d_sp;        /* [ns] x 136 B */
    const float* d_forc;     /* see hlm_set_forcings */
    const int32_t* d_forc_col;
    int64_t forc_ld;
    double forc_dt_hours[HLM_MAX_FORCINGS];
    uint64_t forc_nT[HLM_MAX_FORCINGS];
    int nforc;
    /* solver knobs */
    double params[6];
    double fixed_step;
    uint32_t attempt_cap;
    int math_mode;
    /* outputs, DEVICE pointers; every slot is written (never-reached slots get 0,
     * matching the zeros in the reference's artifacts, SURVEY A.1 note 11) */
    double* d_y_final;       /* [ns][5] */
    double* d_dense;         /* [ns][nq][5]; may be NULL when nq == 0 */
    int32_t* d_stiff;        /* [ns] */
    uint32_t* d_n_attempt;   /* [ns] nullable */
    uint32_t* d_n_accept;    /* [ns] nullable */
    /* Query window (build feature for dense output larger than HBM; the reference's single
     * allocation, rk45_api.hpp:76,92, cannot hold BASELINE config 5).  query_end == 0 means
     * the whole range.  With a window, d_dense holds only rows [query_begin, query_end):
     * [ns][query_end-query_begin][5], and d_resume (hlm_resume_bytes(ns) bytes) carries the
     * integrator state between consecutive launches: resume_in = 0 on the first window,
     * 1 afterwards.  Windows must be visited in order and tile [0, nq).  The committed step
     * sequence, and therefore every output bit, equals the one-shot run. */
    int query_begin, query_end;
    void* d_resume;
    int resume_in;
    /* Device layout of d_dense (W = window length, nq when no window):
     * 0: [ns][W][5], the host-facing order (free-running kernel, hlm_rk45_kernel.h: every lane stores its
     *    rows when it has them; 1.6x write amplification at 1-minute stride);
     * 2: tile-major [ceil(ns/64)][W][5][64], written by the kernel with wave-synchronous stores
     *    (hlm_rk45_dense_kernel.h): every store instruction writes 512 contiguous bytes, HBM write traffic
     *    equals the algorithmic 40 B per row.  What the host API and hlm_rk45_integrate_streamed's
     *    transfer-kernel path use; a caller using it transposes itself.
     * (1, tile-major [ceil(ns/64)][W][64][5], lost on every workload of rounds 1-4 and is refused since ABI 5.) */
    int dense_tiled;
    /* Time chunks / checkpoints (build feature).  use_t_stop != 0: every hillslope pauses the
     * first time its t reaches t_stop (before its next attempt; needs d_resume) and continues in
     * the next launch with resume_in = 1 -- the step sequence, hence every output bit, is that of
     * the uninterrupted run.  The bytes of d_resume ARE a checkpoint (copy them out and back).
     * forc_first_index[j] / forc_rows[j] (rows 0 = all nT[j]) describe which samples of forcing
     * j the block d_forc holds in this launch, [first, first + rows): the host can stream the
     * forcing series through HBM in time chunks ("NetCDF forcings streamed", BASELINE config 3)
     * as long as each chunk covers every sample index floor(t / dt_j) for t in [t_launch, t_stop). */
    int use_t_stop;
    double t_stop;
    uint64_t forc_first_index[HLM_MAX_FORCINGS];
    uint64_t forc_rows[HLM_MAX_FORCINGS];
    /* Output selection (ABI 3; see hlm_set_output_states): dense_ncomp = K > 0 makes every dense row K doubles
     * wide, holding components dense_comp[0..K) (ascending, distinct, < N_EQ) -- d_dense is [ns][W][K],
     * [tiles][W][64][K] or [tiles][W][K][64] by dense_tiled; hlm_rk45_integrate_streamed then fills
     * h_dense [ns][nq][K].  0 = all N_EQ components. */
    int dense_ncomp;
    int dense_comp[HLM_MAX_NEQ];
    /* Radau pass in "continue" mode (build option; the reference's kernel restarts at t0, radau_kernel.cu:44-52):
     * nullable DEVICE buffer of num_systems x (N_EQ + 1) doubles.  When set, a hillslope the RK45 launch flags
     * stiff leaves the state it had reached there ({y[N_EQ], t}), and hlm_radau_fallback_device(job) continues
     * every flagged hillslope from that state -- the rows RK45 wrote up to there stay, nothing before the flag is
     * integrated twice -- taking the hillslopes flagged earliest (most left to do) first.  NULL: restart at t0. */
    double* d_flag_state;
    /* Launch shape (build feature; csrc/hlm_rk45_queue_kernel.h).  time_pieces > 1, for a plain job without dense
     * output (no d_resume / t_stop / query window / forcing chunk): the window [t0, tf] is cut into that many pieces
     * and a persistent grid of waves draws (piece, 64-hillslope tile) items from a device queue until it is empty,
     * instead of one wave per tile for the whole window -- the same integration bit for bit (a lane pauses before
     * the first attempt that starts at or after its piece's end and continues from that state), but the launch no
     * longer drains at its end behind the longest waves: 6 % of a 1 M-hillslope x 30-day launch.  0 / 1: one wave
     * per tile (the reference's map, rk45_kernel.cu:33).  Negative: automatic -- from 4 096 hillslopes on, one piece
     * per twelve simulated hours, at most 32, none when that makes fewer than 8 (few pieces cost more than they
     * save); one wave per tile otherwise.  At most 4096. */
    int time_pieces;
    /* (ABI 4) What the queue hands out is a RUN of consecutive pieces of a tile; only the end of a run passes the
     * lane states through HBM to whichever wave draws the tile's next run, inside a run they stay in registers (the
     * lanes still pause at every piece's end: that re-aligns them in simulated time).  0 = automatic: with more tiles
     * than the device holds waves each run takes half of what is left of the window, at most a quarter and at least a
     * sixteenth of it (32 pieces: 8, 8, 8, 4, 2, 2) -- the granularity the end of the launch needs, a third of the HBM
     * traffic; smaller launches hand every piece over (the queue then evens out how the waves sit on the SIMDs).
     * n >= 1: runs of n pieces each (1 = every piece handed over, round 3's protocol: the better choice when a few
     * tiles cost several times the mean -- hillslopes dropping out with stiff flags --, because long runs start the
     * tiles late in the queue order late).  -n: the automatic schedule with runs of at most n pieces.  Same results
     * bit for bit. */
    int queue_run_pieces;
    /* (ABI 4) G > 1, for a job that launches in time pieces: an item of the queue is a GROUP of G neighbouring tiles
     * and one piece; the wave packs the group's hillslopes that are still integrating into full passes of 64 lanes
     * (on the device, piece by piece, from the status words of the state record) -- flagged hillslopes stop costing
     * lanes without any global re-tiling, extra launch or drain.  With queue_run_pieces = 0 every piece is handed over
     * through the record (what workloads that lose hillslopes want); a run length or a cap named there is honoured:
     * the groups are then packed at the start of every run and stay as packed inside it.  Same results bit for bit.
     * 0 / 1: an item is a tile.  At most 16. */
    int queue_group_tiles;
    /* (ABI 4) Launch shape of a dense launch in the tile layout (dense_tiled = 2): n > 1 cuts the launch's query window
     * into n sub-windows and runs the launch as a persistent grid over (sub-window, tile) items -- a lane pauses at the
     * end of its sub-window as it does at the end of a query window and whichever wave draws the tile's next
     * sub-window continues from the state record -- so that the launch does not end on waves that each run the whole
     * window.  Same rows bit for bit.  0 = automatic, 1 = one wave per tile for the whole window. */
    int dense_sub_windows;
    /* (ABI 5) Diagnostic: nullable DEVICE array of HLM_PATH_COUNTERS uint64 the launch ADDS to (the caller zeroes it):
     * lane-attempts by the RHS form that produced the attempt's result -- [0] unsaturated static store with a dry
     * surface store, [1] unsaturated with a wet surface (both: the two draining stores in closed form), [2] dry surface
     * stage by stage, [3] the general RHS (models/model_204.hpp:94-113 in full: the static store overflows, x2 > 0, the
     * cascade and the Manning pow run) -- [4] lane-attempts computed twice (begun on an unsaturated form, left its
     * regime inside the step, recomputed stage by stage; counted under [2] / [3] too), [5] wave-attempts, [6]
     * wave-attempts that executed more than one form.  Served by a counting instance of the queue kernel: the job must
     * launch in time pieces, FAST arithmetic, adaptive steps, queue_group_tiles <= 1 (else HLM_ERR_INVALID_ARGUMENT).
     * Results are those of the plain instance bit for bit; bench.py's saturating variants read it. */
    uint64_t* d_path_counts;
    /* (ABI 5) How a wave of the queue kernel integrates an attempt when its lanes need DIFFERENT forms of the RHS (FAST
     * arithmetic; Model 204: unsaturated with a dry / a wet surface store, each with the two draining stores in closed
     * form, and the general RHS once the static store overflows).  -1: a block per form present in the wave, each under
     * its lanes' mask, and a lane that overflows inside the step computed twice (rounds 3-4).  1: a wave that held a
     * saturated lane in its previous attempt and holds one now runs ONE block -- the general RHS through the seven
     * stages for every lane -- and every lane keeps the result of its own form (the closed-form values where it
     * qualifies for them): capacities the storms fill, where 40 % of the wave-attempts ran several blocks.  Same bits,
     * lane by lane.  0: automatic = -1 (the one-block instances need two waves per SIMD instead of three: measured
     * -14 ... -19 % where the storms fill the stores, +18 % where they never overflow; csrc/hlm_abi.hip: launch_queue). */
    int queue_hybrid;
} hlm_device_job;
#define HLM_PATH_COUNTERS 7

/* The number of time pieces hlm_rk45_integrate_device(job) launches with (the automatic rule resolved; 0 = one wave
 * per tile: dense output, a resumable job, lane refill, or too small a launch).  No device is touched. */
int hlm_time_pieces_for(const hlm_device_job* job);

size_t hlm_resume_bytes_model(int uid, int num_systems); /* (N_EQ + 2) doubles + 5 int32 per system */
size_t hlm_resume_bytes(int num_systems);                /* the N_EQ = 5 models: 76 bytes per system */
/* Model::N_EQ and the number of forcings Model::rhs reads, for a registered uid (solver/rk45.h:51-66:
 * the reference's kernels are templates over Model::N_EQ; so are these).  State arrays of the C ABI
 * (y0, y_final, dense rows) have N_EQ doubles per system / row. */
int hlm_model_info(int uid, int* n_eq, int* n_forcings);

/* Enqueue the integration on `hip_stream` (a hipStream_t, NULL = default stream) of the
 * current device.  Asynchronous: returns after the launch. */
int hlm_rk45_integrate_device(const hlm_device_job* job, void* hip_stream);

/* Radau-IIA fallback (replaces radau_kernel.cu:20-140 / radau_step_dense.cuh, re-derived as
 * the standard RADAU5 scheme -- see csrc/hlm_radau_kernel.h): every system of `job` whose
 * d_stiff entry is 1 is re-integrated from (t0, y0) with the implicit 3-stage method; its
 * y_final and dense rows are overwritten, d_n_attempt and d_n_accept are incremented by the implicit
 * attempts / accepted steps (same convention for both: RK45 work up to the flag + this pass).  d_stiff stays 1 on success and becomes 2 if the attempt cap ends the run
 * (final row zero; job.attempt_cap, default 2^20 implicit attempts per system for this pass).  Enqueue on the same stream after hlm_rk45_integrate_device(job).  The job
 * must be a plain one-launch job (no d_resume / t_stop / query window / forcing chunk). */
int hlm_radau_fallback_device(const hlm_device_job* job, void* hip_stream);

/* Kernel timing helper for bench.py: runs the job `iters` times on `hip_stream`,
 * bracketing EACH launch with HIP events on that stream; ms[i] receives the elapsed
 * milliseconds of launch i. */
int hlm_rk45_time_device(const hlm_device_job* job, void* hip_stream, int iters, float* ms);

/* Dense output streamed to the host in query windows (BASELINE config 5: "dense-output at
 * 1-min query stride ... overlapped hipMemcpyAsync writeback").  Runs `job` (its d_dense /
 * window fields are ignored) window by window of `window_queries` queries into two
 * alternating device buffers; while window k+1 integrates on `hip_stream`, window k is
 * copied on a second stream into h_dense[(s*nq + q)*5 + c] (the host order of
 * rk45_api.hpp:255-267; pinned memory from hlm_host_alloc_pinned makes the copy truly
 * asynchronous).  Synchronous: returns when every byte is in host memory. */
typedef struct hlm_stream_stats {
    double wall_ms;        /* first launch -> last byte in host memory */
    double kernel_ms;      /* sum of the window kernels (HIP events on the compute stream) */
    double copy_ms;        /* sum of the window D2H copies (HIP events on the copy stream) */
    int windows;
    uint64_t dense_bytes;  /* ns * nq * 5 * 8 */
} hlm_stream_stats;
int hlm_rk45_integrate_streamed(const hlm_device_job* job, int window_queries, double* h_dense,
                                void* hip_stream, hlm_stream_stats* stats);
/* The same pipeline for dense output that does not fit in host memory either (BASELINE config 5 on its real
 * size: 500 000 hillslopes x 43 201 queries = 864 GB per GPU): every window lands in one of two PINNED host
 * staging buffers the library owns, and `sink` is called on the calling thread as soon as its bytes are there,
 * h_rows = [ns][query_end - query_begin][K] (K = N_EQ or the job's output selection), valid until the sink
 * returns -- while the next window integrates and the one after is copied.  A consumer writes the rows to its
 * file (the reference: write_dense_netcdf, I_O/output_series.cpp:18-72, one call for the whole array).
 * sink == NULL discards the rows (measurement of the pipeline itself).  A non-zero return of the sink ends the
 * run with HLM_ERR_STATE.  stats->dense_bytes counts the bytes delivered. */
typedef int (*hlm_window_sink)(void* user, int query_begin, int query_end, const double* h_rows);
int hlm_rk45_integrate_streamed_sink(const hlm_device_job* job, int window_queries, hlm_window_sink sink, void* user,
                                     void* hip_stream, hlm_stream_stats* stats);
/* run_rk45<M> (rk45_api.hpp:273-313) for dense output that fits neither HBM nor a host array: the host-buffer form
 * of hlm_rk45_integrate_streamed_sink.  Registered parameters, forcings (resident: hlm_set_forcings) and output
 * selection apply; the rows reach `sink` window by window ([ns][W][K], see above) instead of an h_dense array;
 * finals, flags and counters come back as from hlm_run_rk45.  Not combinable with the Radau fallback (which needs
 * the whole query range in one launch). */
int hlm_run_rk45_streamed(int uid, const double* h_y0, int num_systems, double t0, double tf, const double* h_query_times,
                          int num_queries, const void* d_sp, int window_queries, hlm_window_sink sink, void* user,
                          double* h_final, int32_t* h_stiff, uint32_t* h_n_attempt, uint32_t* h_n_accept);
/* Forcing series streamed through HBM in time chunks (BASELINE config 3: "NetCDF precip+t2m
 * forcings streamed").  h_tables[j]: HOST float [job->forc_nT[j]][job->forc_ld] of forcing j
 * (pinned memory makes the uploads asynchronous); job->d_forc is ignored.  The window [t0, tf]
 * is cut into chunks of chunk_minutes; while chunk k integrates on `hip_stream` (hillslopes
 * pause at its end time), the sample rows chunk k+1 needs are uploaded on a second stream into
 * the other of two device buffers.  Results are bit-identical to the run with all forcings
 * resident.  Synchronous.  *chunks (nullable) receives the number of launches.
 * chunk_minutes < 0 (ABI 4): GROWING chunks -- the first |chunk_minutes| long, each next one twice as long up to 8 x
 * (1, 2, 4, 8, 8, ... days for -1440): only the first, short upload is not hidden behind an integration, and the
 * long chunks run in the queue kernel's launch shape (hlm_device_job.time_pieces, pieces cut from the chunk) instead
 * of one launch per day.  The two device buffers then hold the longest chunk each. */
int hlm_rk45_integrate_forcing_streamed(const hlm_device_job* job, const float* const* h_tables,
                                        double chunk_minutes, void* hip_stream, int* chunks);

/* Transfer path of hlm_rk45_integrate_streamed.  0 (default): linear device windows +
 * hipMemcpy2DAsync on the copy engine (fastest end to end: 54.8 GB/s measured).  2: tile-major
 * device windows (the layout of dense_tiled = 2, coalesced kernel stores) + a transposing kernel that writes
 * pinned host memory directly (51.5 GB/s; falls back to path 0 when h_dense is not pinned).  (1 was the tile-row
 * layout of rounds 1-4, removed in ABI 5.) */
int hlm_set_stream_path(int path);
int hlm_host_alloc_pinned(void** hptr, size_t bytes);
int hlm_host_free_pinned(void* hptr);

/* Self-test of the device elementary functions the FAST math mode substitutes (the analogue of
 * the reference's printf debug kernels, main.cpp:184-246): evaluates function `which` on n
 * doubles.  0: x^(2/3) FAST, 1: x^(-1/5) FAST, 2: 1/x FAST, 3: x^(2/3) STRICT, 4: x^(-1/5) STRICT, 5 / 6: n / d by
 * the invariant divisor's reciprocal / IEEE (divisors in d_in[n .. 2n)), 7 / 8: the seed residual inside 3 / 4's power, 9: (1/x)^(1/4) STRICT (the Radau pass's controller). */
int hlm_selftest_math(int which, const double* d_in, double* d_out, int n, void* hip_stream);

/* Single device evaluations of Model::rhs (the reference's debugRHS kernel, main.cpp:219-246):
 * d_y [n][5], d_sp [n] SpatialParams, d_F [n][2] float forcing samples, d_out [n][5].
 * fast_path = 1 evaluates the model's dry-surface RHS instead, fast_path = 2 its unsaturated
 * ("no-rain") RHS, fast_path = 3 (FAST arithmetic) the unsaturated RHS with a wet surface store
 * (each valid where its precondition holds); they write their "precondition broke" flag to
 * d_broke[n] (nullable). */
int hlm_selftest_rhs(int uid, int math_mode, int fast_path, const double* d_y, const void* d_sp,
                     const float* d_F, double* d_out, int32_t* d_broke, int n, void* hip_stream);

/* ---- one process, the GPUs of a node (ABI 4) -----------------------------------------------------------------
 * Replaces the reference's rank-0 split over worker ranks (main.cpp:258-309: contiguous chunks, base = ns / G, the
 * first ns % G of them one hillslope longer; 2 x MPI_Send / MPI_Recv per worker, :300-303, :357-368) by a single
 * process driving `num_shards` shards on `num_devices` HIP devices (shard s on devices[s % num_devices]; one HIP
 * stream per device, shards of a device one after the other), and adds what the reference does not have: an
 * end-of-run GATHER of the results -- final states, stiff flags and both step counters, 52 bytes per hillslope for
 * N_EQ = 5 -- to devices[0] over RCCL (one group of ncclSend / ncclRecv pairs: xGMI point-to-point between the GPUs
 * of a node), from where one copy per array delivers them to the host arrays in hillslope order.
 * Everything is a HOST buffer of the whole domain; solver parameters, math mode, attempt cap and launch shape are the
 * registered ones (hlm_set_model_parameters ...); no dense output (each GPU streams its own dense rows: the host API
 * above, per device).  Forcings: nforc tables h_forc_tables[j] = float [forc_nT[j]][forc_ld] and either
 * h_forc_col[ns] (column of each hillslope in the tables: compact per-grid-cell tables, every device gets the tables
 * and its slice of the index) or NULL with forc_ld == num_systems (the reference's per-stream expansion,
 * main.cpp:527-549: every device gets its columns). */
typedef struct hlm_sharded_run {
    int uid;
    int num_systems;
    double t0, tf;
    const double* h_y0;               /* [ns][N_EQ] */
    const void* h_sp;                 /* [ns] hlm_spatial_params (136 B each); NULL for a model without parameters */
    const float* const* h_forc_tables;
    double forc_dt_hours[HLM_MAX_FORCINGS];
    uint64_t forc_nT[HLM_MAX_FORCINGS];
    int nforc;
    int64_t forc_ld;
    const int32_t* h_forc_col;
} hlm_sharded_run;

#define HLM_GATHER_HOST 0 /* every device copies its shards' results to the host itself (no exchange between GPUs) */
#define HLM_GATHER_RCCL 1 /* RCCL send / recv to devices[0], one copy per array from there */

#define HLM_MAX_SHARD_DEVICES 16 /* devices of one node a sharded run may name (per-device fields of hlm_sharded_stats) */

typedef struct hlm_sharded_stats {
    double upload_ms;     /* host clock: the inputs of all shards in device memory (the slowest device's uploads) */
    double integrate_ms;  /* HIP events around each device's launches: the slowest device */
    double wall_to_integrated_ms; /* host clock: first launch -> all devices finished (includes the communicator's start) */
    double gather_ms;     /* HLM_GATHER_RCCL: the send / recv group on the root's stream (HIP events); else 0 */
    double download_ms;   /* host clock: results in the host arrays */
    double comm_init_ms;  /* HLM_GATHER_RCCL: getting the communicator, beside the running kernels (host clock; ~1 s the
                           * first time RCCL comes up in a process, ~0 afterwards: one communicator per device list is kept) */
    uint64_t gather_bytes; /* bytes that crossed the communicator (incl. the root's own shards) */
    int rccl_version;     /* ncclGetVersion, 0 without RCCL */
    int num_shards, num_devices;
    /* (ABI 5) */
    int rccl_nranks;      /* ncclCommCount of the gather's communicator: how many ranks RCCL saw (0 without RCCL) */
    int upload_mode;      /* HLM_UPLOAD_* the run used */
    int device_ids[HLM_MAX_SHARD_DEVICES];          /* the HIP devices, in rank order (rank 0 = the gather's root) */
    double upload_ms_device[HLM_MAX_SHARD_DEVICES];    /* per device: its uploads, first byte to completion */
    double integrate_ms_device[HLM_MAX_SHARD_DEVICES]; /* per device: HIP events around its launches (streamed: its pipeline's wall time) */
    uint64_t dense_bytes;    /* hlm_run_rk45_sharded_streamed: bytes of dense rows delivered to the sink, all devices */
    double dense_kernel_ms;  /* hlm_run_rk45_sharded_streamed: sum of the window kernels over all devices */
} hlm_sharded_stats;

/* How the inputs reach the devices.  THREADS (default): one host thread per device issues that device's copies, so the
 * uploads to D devices run side by side (a copy from the caller's pageable arrays occupies the calling thread; one
 * thread feeding D devices -- SINGLE_THREAD, round 4's path -- uploads to them one after the other).  STAGED_THREADS:
 * the threads additionally pack the inputs through a pair of pinned buffers each; slower than the runtime's own
 * pageable copies where those run at the link's rate (measured: 55 GB/s against 27-34 GB/s), selectable for hosts
 * where they do not. */
#define HLM_UPLOAD_SINGLE_THREAD 0
#define HLM_UPLOAD_THREADS 1
#define HLM_UPLOAD_STAGED_THREADS 2
int hlm_set_sharded_upload(int mode);

/* The end-of-run gather of hlm_run_rk45_sharded as a LIST (no device is touched): piece k moves `bytes` bytes of
 * shard `shard`'s result array `array` (0 = y_final, N_EQ doubles per hillslope; 1 = stiff; 2 = n_attempt; 3 = n_accept)
 * from rank `from_rank` (= shard % num_devices) into the root's gathered array at byte offset `dst_offset_bytes`
 * (source offset 0: a shard's arrays are its own).  Four pieces per non-empty shard, in shard order.  pieces == NULL
 * only counts (*num_pieces = the length of the plan); else *num_pieces holds the number of entries `pieces` has room
 * for on entry (too few: HLM_ERR_INVALID_ARGUMENT, *num_pieces = the length needed) and the length of the plan on
 * return.  The run builds its ncclSend / ncclRecv group from exactly this list. */
typedef struct hlm_gather_piece {
    int shard, from_rank, array;
    uint64_t dst_offset_bytes, bytes;
} hlm_gather_piece;
int hlm_gather_plan(int num_systems, int num_shards, int num_devices, int neq, hlm_gather_piece* pieces, int* num_pieces);

/* devices == NULL: devices 0 .. num_devices - 1.  The devices must be distinct.  num_shards >= 1 (more shards than
 * devices rehearses a larger node's split on a smaller one).  h_stiff / h_n_attempt / h_n_accept / stats are optional.
 * Results are bit-identical to hlm_run_rk45 of the whole domain on one device, whatever the split.  hlm_set_radau_fallback
 * applies: every shard's flagged systems go through the implicit pass on the shard's device before the gather (as each of
 * the reference's ranks relaunches for its own, rk45_api.hpp:189-247). */
int hlm_run_rk45_sharded(const hlm_sharded_run* run, int num_shards, const int* devices, int num_devices,
                         int gather_mode, double* h_y_final, int32_t* h_stiff, uint32_t* h_n_attempt,
                         uint32_t* h_n_accept, hlm_sharded_stats* stats);
/* (ABI 5) BASELINE config 5 on the GPUs of a node in one call: hlm_run_rk45_sharded with dense output.  Every device
 * streams the rows of ITS shards through its own two device windows and two pinned staging buffers
 * (hlm_rk45_integrate_streamed_sink per shard, one host thread per device) into `sink` -- no gather of dense rows
 * between GPUs (6.9 TB for config 5); the reference's ranks each write their own dense file (main.cpp:796-823).
 * sink(user, shard, sys_begin, sys_count, query_begin, query_end, rows): rows = [sys_count][query_end - query_begin][K]
 * of the hillslopes sys_begin .. sys_begin + sys_count - 1 (K = N_EQ or the registered output selection), valid until
 * it returns.  It is called from the device's host thread: calls of one device come one after the other, calls of
 * different devices may overlap.  NULL discards the rows.  Finals, flags and counters arrive in the host arrays by
 * one copy per shard (HLM_GATHER_HOST).  Rows are bit-identical to the one-launch run of the whole domain. */
typedef int (*hlm_shard_window_sink)(void* user, int shard, int sys_begin, int sys_count, int query_begin, int query_end,
                                     const double* h_rows);
int hlm_run_rk45_sharded_streamed(const hlm_sharded_run* run, int num_shards, const int* devices, int num_devices,
                                  const double* h_query_times, int num_queries, int window_queries,
                                  hlm_shard_window_sink sink, void* user, double* h_y_final, int32_t* h_stiff,
                                  uint32_t* h_n_attempt, uint32_t* h_n_accept, hlm_sharded_stats* stats);
/* The reference's split rule (main.cpp:275-307 with G shards): first hillslope and length of shard s. */
int hlm_shard_bounds(int num_systems, int num_shards, int shard, int* start, int* count);

/* One Dormand-Prince attempt (the reference's rk45_step, solver/rk45_step_dense.cuh:33-145, preceded by its
 * k0 = rhs(t, y, F), rk45_kernel.cu:114) per (state d_y[n][N_EQ], parameters d_sp[n], forcing pair d_F[n][2], step
 * d_h[n]) on the device: d_ynew[n][N_EQ] = the 5th-order solution, d_err_jump[n][2] = {scaled error norm,
 * |k0 - k1|_inf (event_detector.cuh:46-53)}.  path = HLM_ATTEMPT_SELECT_NODENSE: exactly what a launch without dense
 * output executes per attempt (per-hillslope choice of the cheaper exact RHS forms, linear stores in closed form in
 * FAST arithmetic); HLM_ATTEMPT_SELECT_DENSE: what the dense kernels execute; 0..3: the general / dry-surface /
 * unsaturated / unsaturated-wet-surface form for every hillslope (valid where d_broke[n] comes back 0), `closed` != 0
 * with the linear stores in closed form (forms 2, 3; FAST).  The unit-level parity hook under the integrator's loop
 * (the reference's counterpart: debugRHS, main.cpp:219-246). */
#define HLM_ATTEMPT_SELECT_NODENSE (-1)
#define HLM_ATTEMPT_SELECT_DENSE (-2)
int hlm_selftest_attempt(int uid, int math_mode, int path, int closed, const double* d_y, const void* d_sp,
                         const float* d_F, const double* d_h, double rtol, double atol, double* d_ynew,
                         double* d_err_jump, int32_t* d_broke, int n, void* hip_stream);

/* Static facts about the compiled kernels (for DESIGN.md / bench output / the register-budget
 * test): VGPRs per lane, scratch (spill) bytes per lane, static LDS bytes per workgroup,
 * maximum workgroup size.  168 VGPRs = 3 waves per SIMD, 256 = 2. */
int hlm_kernel_info(int uid, int dense, int math_mode, int* vgprs, int* scratch_bytes,
                    int* lds_bytes, int* max_threads);

#ifdef __cplusplus
}
#endif
#endif /* HLM_HIP_H */
