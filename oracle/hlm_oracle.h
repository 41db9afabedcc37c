/*
 * hlm_oracle.h -- CPU restatement of the reference's per-hillslope RK45 path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (tiger_hlm_gpu_amd/, include/,
 * tools/) may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py use it, as the checker / reported CPU baseline.
 *
 * Parity pinning: the restatement is checked against the reference's own committed
 * result artifacts (tests/golden/, made by tests/golden/make_golden.py):
 *   G1  src/final.csv + src/dense.csv            (DummyModel, 6 significant digits)
 *   G2  src/final_example.nc + dense_example.nc  (Model 204, <= 1e-13 relative)
 * The reference itself cannot be compiled in this image (its sources need
 * <cuda_runtime.h>, nvcc, MPI and netcdf; writing stand-ins is not allowed), so there
 * is no oracle/_ref build; see DESIGN.md "Oracle".
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */
#ifndef HLM_ORACLE_H
#define HLM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HLM_ORACLE_NEQ 5   /* the largest N_EQ (array bound); a model's own N_EQ: hlm_oracle_neq(uid) */
#define HLM_ORACLE_MAX_FORCINGS 16 /* I_O/forcing_data.h:5 */

/* Model UIDs.  204 is models/model_204.hpp:18.  The DummyModel source is absent from
 * the reference tree (README.md:41-42 names it; its RHS survives only in
 * model_dummy_python.ipynb:65-89); the build assigns it UID 0. */
#define HLM_ORACLE_UID_DUMMY 0
#define HLM_ORACLE_UID_204 204
#define HLM_ORACLE_UID_TOY3 3   /* the build's 3-state example model (not a reference model) */

/* Per-system status written to stiff[] (rk45_kernel.cu:160-170 uses 0/1 only;
 * 2 is the build's attempt-cap guard against the slope-jump livelock). */
#define HLM_ORACLE_OK 0
#define HLM_ORACLE_STIFF 1
#define HLM_ORACLE_CAPPED 2

/* I_O/parameters_loader.hpp:19-37 -- 2 x long + 15 x double = 136 bytes, no padding. */
typedef struct {
    long stream;
    long next_stream;
    double c1, infil, perco, Hu, lat, sw, ss, n_mann, slope, L, A_h, alpha3, alpha4,
        melt_f, temp_thr;
} hlm_oracle_spatial_params;

/* models/model_204.hpp:22-30 */
typedef struct {
    double initialStep, rtol, atol, safety, minScale, maxScale;
} hlm_oracle_solver_params;

typedef struct {
    int uid;                 /* HLM_ORACLE_UID_* */
    int ns;                  /* number of systems */
    int nq;                  /* number of query times */
    double t0, tf;
    const double* y0;        /* [ns][5] AoS */
    const double* query;     /* [nq], ascending */
    const hlm_oracle_spatial_params* sp; /* [ns], may be NULL for the DummyModel */
    const float* forc;       /* [sum_j nT_j][ns] forcing-major, main.cpp:527-549 */
    const double* forc_dt_h; /* [nforc] hours */
    const uint64_t* forc_nT; /* [nforc] */
    int nforc;
    /* Optional [nforc] constant DOUBLE forcings replacing the float gather: reproduces
     * the stub-forcing revision (model_204.hpp:76-77) that produced G2.  NULL = gather. */
    const double* forc_stub_f64;
    hlm_oracle_solver_params p;
    double fixed_step;       /* > 0: controller off, h = fixed_step (build feature, C2) */
    uint32_t attempt_cap;    /* 0 = 2^32-1 */
    int nthreads;            /* OpenMP threads; <= 0 = omp default */
    /* outputs (all caller-allocated; zero-filled by the oracle before use) */
    double* y_final;         /* [ns][5] AoS; untouched (0) for flagged systems */
    double* dense_raw;       /* [ns][5][nq] device layout of rk45_kernel.cu:145; may be NULL if nq==0 */
    int32_t* stiff;          /* [ns] */
    uint32_t* n_attempt;     /* [ns] nullable */
    uint32_t* n_accept;      /* [ns] nullable */
    double* t_reached;       /* [ns] nullable: time at which the loop stopped */
    double* y_reached;       /* [ns][N_EQ] nullable: the state at t_reached (what a flagged system had reached) */
    /* hlm_oracle_radau_run only.  0: every flagged system restarts from (t0, y0), as the reference's kernel does
     * (radau_kernel.cu:44-52).  1 (build option, needs t_reached and y_reached of the RK45 run): it CONTINUES
     * from the state at which RK45 gave up -- the rows RK45 wrote up to there stay, the implicit method takes
     * over from (t_reached, y_reached) with the first step initialStep. */
    int radau_continue;
    /* hlm_oracle_rk45_run only: SEGMENT REPLAY (tests/conftest.py, third tier of the parity yardstick).  start_t /
     * start_h (nullable, [ns]): system s enters the loop of rk45_kernel.cu:53 at (start_t[s], y0[s]) with the step
     * start_h[s] -- a state another run of the same loop had reached -- instead of (t0, initialStep); t0 keeps its role
     * in the minimum-step rule (:160).  use_t_stop: a system leaves the loop before the first attempt whose start time
     * has reached t_stop (status HLM_ORACLE_PAUSED; t_reached / y_reached / h_reached hold the state): the pause
     * protocol of the GPU's time chunks (hlm_device_job.t_stop), so that the two can be compared chunk by chunk. */
    const double* start_t;
    const double* start_h;
    int use_t_stop;
    double t_stop;
    double* h_reached;       /* [ns] nullable: the step the next attempt would have used */
    /* hlm_oracle_radau_run only, diagnostic (tests/radau_wave_sim.py: what a wave of the GPU's state machine is made of):
     * one byte per implicit attempt of system s at radau_trace[s * radau_trace_cap + k], k < radau_trace_cap -- low nibble
     * the simplified-Newton iterations it took, 0x10 accepted and landed on a forcing sample boundary, 0x20 Newton did
     * not converge, 0x40 rejected by the error test.  NULL = no trace. */
    uint8_t* radau_trace;
    int64_t radau_trace_cap;
} hlm_oracle_job;
#define HLM_ORACLE_PAUSED 3

/* The whole integration, rk45_kernel.cu:17-176.  Returns 0, or -1 on bad arguments. */
int hlm_oracle_rk45_run(const hlm_oracle_job* job);

/* Radau-IIA fallback: every system whose stiff[] entry equals
 * HLM_ORACLE_STIFF is re-integrated from (t0, y0) with the implicit 3-stage Radau IIA method
 * (order 5) and its y_final / dense rows / n_attempt are overwritten; stiff[] stays 1 on
 * success (as rk45_api.hpp:189-247 leaves d_stiff) and becomes HLM_ORACLE_CAPPED if the
 * attempt cap is hit.  The reference's radau_step_dense.cuh cannot be restated as written
 * (inconsistent dense interpolant, ad-hoc error weights, wrong forcing indexing, results never
 * copied back -- SURVEY A.3), so this is the standard RADAU5 formulation of Hairer & Wanner
 * re-derived on the reference's skeleton (radau_kernel.cu:20-140): finite-difference Jacobian
 * as approx_jacobian (radau_step_dense.cuh:13-31), <= 10 simplified-Newton iterations on the
 * 15x15 system (pivoted LU), the RADAU5 error estimator and collocation dense output, the same
 * forcing hold / query / controller conventions as the RK45 loop, plus the build's own: steps clipped to the forcing
 * sample boundaries, and a step memory across changes of a held sample (hlm_oracle_radau.c).  Parity for this row is
 * NOT pinned by any reference artifact; it is validated against SciPy's Radau. */
int hlm_oracle_radau_run(const hlm_oracle_job* job);

/* rk45_api.hpp:255-267: [sys][comp][q] -> [sys][q][comp]. */
void hlm_oracle_reorder_dense(const double* raw, double* out, int ns, int nq);

/* Single evaluations for known-answer tests. */
void hlm_oracle_rhs(int uid, double t, const double* y, double* dydt,
                    const hlm_oracle_spatial_params* sp, const double* F, int nforc);
int hlm_oracle_neq(int uid);          /* Model::N_EQ (5, or 3 for the toy model); 0 = unknown uid */
int hlm_oracle_has_jacobian(int uid); /* 0: the Radau pass differentiates numerically (radau_step_dense.cuh:13-31) */
void hlm_oracle_reorder_dense_n(const double* raw, double* out, int ns, int nq, int neq);
/* exact one-sided d(rhs)/dy (row-major J[i][j] = d dydt_i / d y_j) used by the Radau-IIA pass */
void hlm_oracle_jacobian(int uid, const double* y, double Jm[HLM_ORACLE_NEQ][HLM_ORACLE_NEQ],
                         const hlm_oracle_spatial_params* sp, const double* F, int nforc);
/* rk45_step_dense.cuh:33-145.  k[0] must hold rhs(t,y).  F are the (double-converted)
 * forcing values held for all stages. */
void hlm_oracle_rk45_step(int uid, double t, const double* y, double h, double rtol,
                          double atol, const hlm_oracle_spatial_params* sp,
                          const double* F, int nforc, double k[7][HLM_ORACLE_NEQ],
                          double* y_out, double* err_norm);
/* rk45_step_dense.cuh:171-244 */
void hlm_oracle_rk45_dense(const double* y_n, double k[7][HLM_ORACLE_NEQ], double h,
                           double theta, double* y_dense);

/* DP5(4) tableau exported for coefficient known-answer tests. */
const double* hlm_oracle_tableau_c(void);      /* [7]   */
const double* hlm_oracle_tableau_a(void);      /* [7][6] row-major */
const double* hlm_oracle_tableau_b(void);      /* [7]   */
const double* hlm_oracle_tableau_balt(void);   /* [7]   */
const double* hlm_oracle_tableau_P(void);      /* [7][4] row-major */

#ifdef __cplusplus
}
#endif
#endif
