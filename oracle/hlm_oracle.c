/*
 * hlm_oracle.c -- CPU restatement (plain C) of the reference's hot path.
 * TEST INFRASTRUCTURE ONLY: see hlm_oracle.h.  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fopenmp: IEEE evaluation in source order, no FMA).
 *
 * The floating-point evaluation ORDER below is part of the contract: the accept /
 * reject sequence of an adaptive integrator depends on last-bit rounding, so every
 * sum is accumulated in the order the reference writes it, zero tableau entries
 * included.
 */
#include "hlm_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEQ HLM_ORACLE_NEQ

/* event_detector.cuh:11,15 */
static const double SLOPE_JUMP_THRESH = 100.0;
static const double MIN_STEP_FRACTION = 1e-6;

/* rk45_step_dense.cuh:54-83 */
static const double TAB_C[7] = {0.0, 1.0 / 5.0, 3.0 / 10.0, 4.0 / 5.0, 8.0 / 9.0, 1.0, 1.0};
static const double TAB_A[7][6] = {
    {0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {1.0 / 5.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {3.0 / 40.0, 9.0 / 40.0, 0.0, 0.0, 0.0, 0.0},
    {44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0, 0.0, 0.0, 0.0},
    {19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0, 0.0, 0.0},
    {9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0, 0.0},
    {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0}};
static const double TAB_B[7] = {35.0 / 384.0,      0.0,         500.0 / 1113.0, 125.0 / 192.0,
                                -2187.0 / 6784.0, 11.0 / 84.0, 0.0};
static const double TAB_BALT[7] = {5179.0 / 57600.0,    0.0,            7571.0 / 16695.0, 393.0 / 640.0,
                                   -92097.0 / 339200.0, 187.0 / 2100.0, 1.0 / 40.0};
/* rk45_step_dense.cuh:193-219 (SciPy's DP dense-output matrix) */
static const double TAB_P[7][4] = {
    {1.0, -8048581381.0 / 2820520608.0, 8663915743.0 / 2820520608.0, -12715105075.0 / 11282082432.0},
    {0.0, 0.0, 0.0, 0.0},
    {0.0, 131558114200.0 / 32700410799.0, -68118460800.0 / 10900136933.0, 87487479700.0 / 32700410799.0},
    {0.0, -1754552775.0 / 470086768.0, 14199869525.0 / 1410260304.0, -10690763975.0 / 1880347072.0},
    {0.0, 127303824393.0 / 49829197408.0, -318862633887.0 / 49829197408.0, 701980252875.0 / 199316789632.0},
    {0.0, -282668133.0 / 205662961.0, 2019193451.0 / 616988883.0, -1453857185.0 / 822651844.0},
    {0.0, 40617522.0 / 29380423.0, -110615467.0 / 29380423.0, 69997945.0 / 29380423.0}};

const double* hlm_oracle_tableau_c(void) { return TAB_C; }
const double* hlm_oracle_tableau_a(void) { return &TAB_A[0][0]; }
const double* hlm_oracle_tableau_b(void) { return TAB_B; }
const double* hlm_oracle_tableau_balt(void) { return TAB_BALT; }
const double* hlm_oracle_tableau_P(void) { return &TAB_P[0][0]; }

/* models/model_204.hpp:43-114.  F holds the forcing values already widened to double
 * (the reference widens float F[0], F[1] at :82-83). */
static void rhs_204(const double* y, double* dydt, const hlm_oracle_spatial_params* P,
                    const double* F, int nforc)
{
    const double h_snow = y[0], h_stat = y[1], h_surf = y[2], h_grav = y[3], h_aq = y[4];
    const double rainfall = nforc > 0 ? F[0] : 0.0;
    const double temperature = nforc > 1 ? F[1] : 0.0;

    /* snow store, :87-91 */
    const double snowmelt = (temperature >= P->temp_thr) ? fmin(h_snow, temperature * P->melt_f) : 0.0;
    const double x1 = rainfall + snowmelt;
    dydt[0] = rainfall - snowmelt;

    /* static store, :94-98 */
    const double x2 = fmax(0.0, x1 + h_stat - P->Hu);
    const double d1 = x1 - x2;
    const double Emax = fmin(0.1 * temperature, h_stat);
    const double s = h_stat / P->Hu;
    dydt[1] = d1 - s * Emax;

    /* surface store, :101-105.  pow of a negative h_surf is NaN and fmin(1.0, NaN) is
     * 1.0 (IEEE minNum): the NaN never leaves the RHS (SURVEY A.1 note 6). */
    const double x3 = fmin(x2, P->infil);
    const double d2 = x2 - x3;
    const double alfa2 = (1.0 / P->n_mann) * pow(h_surf, 2.0 / 3.0) * sqrt(P->slope);
    const double w = fmin(1.0, alfa2 * P->L / P->A_h * 60.0);
    dydt[2] = d2 - h_surf * w;

    /* gravitational store, :108-110 */
    const double x4 = fmin(x3, P->perco);
    const double d3 = x3 - x4;
    dydt[3] = d3 - (P->alpha3 >= 1.0 ? h_grav / P->alpha3 : 0.0);

    /* aquifer, :113 */
    dydt[4] = x4 - (P->alpha4 >= 1.0 ? h_aq / P->alpha4 : 0.0);
}

/* ToyModel3 (the build's "Add a New Model" example, csrc/hlm_models.h: ToyDev): 3 stores, 1 forcing,
 * a feedback from store 2 to store 0.  Not a reference model: the oracle states it so that the generic
 * kernels (N_EQ = 3, NF = 1, finite-difference Jacobian + pivoted LU in the Radau pass) have a checker. */
static void rhs_toy3(const double* y, double* dydt, const hlm_oracle_spatial_params* P, const double* F, int nforc)
{
    const double k1 = P && P->alpha3 > 0.0 ? 1.0 / P->alpha3 : 0.05, k3 = P && P->alpha4 > 0.0 ? 1.0 / P->alpha4 : 0.01;
    const double inflow = nforc > 0 ? F[0] : 0.0;
    const double out0 = k1 * y[0], up = 0.3 * y[0] * y[2], out1 = 0.02 * y[1] * y[1];
    dydt[0] = inflow - out0 - up;
    dydt[1] = out0 - out1;
    dydt[2] = out1 - k3 * y[2] + up;
}

/* Model::N_EQ of a uid (solver/rk45.h:51-66: the reference's kernels are templates over it); 0 = unknown */
int hlm_oracle_neq(int uid)
{
    return uid == HLM_ORACLE_UID_204 || uid == HLM_ORACLE_UID_DUMMY ? 5 : (uid == HLM_ORACLE_UID_TOY3 ? 3 : 0);
}

/* model_dummy_python.ipynb:65-89 (the code cell, I2 = 0.6*H1). */
static void rhs_dummy(const double* y, double* dydt)
{
    const double H0 = y[0], H1 = y[1], H3 = y[3];
    const double X0 = 1.0, Y0 = 0.5 * H0, X1 = 1.2, X2 = 0.3 * H1;
    const double Y1 = 0.4, Y2 = 0.2, Y3 = 0.3, Y4 = 0.1;
    const double I2 = 0.6 * H1, I3 = 0.4 * H3;
    dydt[0] = X0 - Y0;
    dydt[1] = X1 + Y0 - X2 - Y1 - I2;
    dydt[2] = X2 - Y2;
    dydt[3] = I2 - I3 - Y3;
    dydt[4] = I3 - Y4;
}

void hlm_oracle_rhs(int uid, double t, const double* y, double* dydt,
                    const hlm_oracle_spatial_params* sp, const double* F, int nforc)
{
    (void)t; /* model_204.hpp:78 computes doy from t and never uses it */
    if (uid == HLM_ORACLE_UID_204)
        rhs_204(y, dydt, sp, F, nforc);
    else if (uid == HLM_ORACLE_UID_TOY3)
        rhs_toy3(y, dydt, sp, F, nforc);
    else
        rhs_dummy(y, dydt);
}

/* d(rhs)/dy of the two models above, for the Radau-IIA pass (the reference differentiates
 * numerically, radau_step_dense.cuh:13-31: 5 more RHS calls per attempt; the models' RHS are piecewise
 * linear apart from the Manning term, so the exact one-sided derivative is a handful of comparisons).
 * Every min / max contributes the derivative of the operand it returns (a < b picks a), exactly the
 * choice the RHS itself makes; w = min(1, c h^(2/3)) gives d(h w)/dh = 5/3 c h^(2/3) below the cap and
 * 1 at it (also for h < 0, where the RHS drains at rate 1: SURVEY A.1 note 6). */
int hlm_oracle_has_jacobian(int uid) { return uid == HLM_ORACLE_UID_204 || uid == HLM_ORACLE_UID_DUMMY; }

void hlm_oracle_jacobian(int uid, const double* y, double Jm[NEQ][NEQ],
                         const hlm_oracle_spatial_params* P, const double* F, int nforc)
{
    for (int i = 0; i < NEQ; ++i)
        for (int j = 0; j < NEQ; ++j) Jm[i][j] = 0.0;
    if (uid != HLM_ORACLE_UID_204) {  /* DummyModel: constant */
        Jm[0][0] = -0.5;
        Jm[1][0] = 0.5; Jm[1][1] = -0.9;
        Jm[2][1] = 0.3;
        Jm[3][1] = 0.6; Jm[3][3] = -0.4;
        Jm[4][3] = 0.4;
        return;
    }
    const double h_snow = y[0], h_stat = y[1], h_surf = y[2];
    const double rainfall = nforc > 0 ? F[0] : 0.0;
    const double temperature = nforc > 1 ? F[1] : 0.0;
    const double melt_pot = temperature * P->melt_f;
    const int warm = temperature >= P->temp_thr;
    const double snowmelt = warm ? fmin(h_snow, melt_pot) : 0.0;
    const double m = (warm && h_snow < melt_pot) ? 1.0 : 0.0;       /* d snowmelt / d h_snow */
    const double x1 = rainfall + snowmelt;
    const double over = x1 + h_stat - P->Hu;
    const double x2 = fmax(0.0, over);
    const double a = over > 0.0 ? 1.0 : 0.0;                        /* d x2 / d (x1 + h_stat) */
    const double e_pot = 0.1 * temperature;
    const double Emax = fmin(e_pot, h_stat);
    const double e = h_stat < e_pot ? 1.0 : 0.0;                    /* d Emax / d h_stat */
    const double s = h_stat / P->Hu;
    const double x3 = fmin(x2, P->infil);
    const double b = x2 < P->infil ? 1.0 : 0.0;                     /* d x3 / d x2 */
    const double c = x3 < P->perco ? 1.0 : 0.0;                     /* d x4 / d x3 */
    const double alfa2 = (1.0 / P->n_mann) * pow(h_surf, 2.0 / 3.0) * sqrt(P->slope);
    const double w_lin = alfa2 * P->L / P->A_h * 60.0;              /* NaN for h_surf < 0: the cap applies */
    Jm[0][0] = -m;
    Jm[1][0] = m - a * m;
    Jm[1][1] = -a - (Emax / P->Hu + s * e);
    Jm[2][0] = (1.0 - b) * a * m;
    Jm[2][1] = (1.0 - b) * a;
    Jm[2][2] = -(w_lin < 1.0 ? (5.0 / 3.0) * w_lin : 1.0);
    Jm[3][0] = (1.0 - c) * b * a * m;
    Jm[3][1] = (1.0 - c) * b * a;
    Jm[3][3] = -(P->alpha3 >= 1.0 ? 1.0 / P->alpha3 : 0.0);
    Jm[4][0] = c * b * a * m;
    Jm[4][1] = c * b * a;
    Jm[4][4] = -(P->alpha4 >= 1.0 ? 1.0 / P->alpha4 : 0.0);
}

/* rk45_step_dense.cuh:94-142 */
void hlm_oracle_rk45_step(int uid, double t, const double* y, double h, double rtol,
                          double atol, const hlm_oracle_spatial_params* sp,
                          const double* F, int nforc, double k[7][NEQ], double* y_out,
                          double* err_norm)
{
    const int n = hlm_oracle_neq(uid);
    double y_temp[NEQ];
    for (int s = 1; s < 7; ++s) { /* :94-107 */
        for (int i = 0; i < n; ++i) {
            double acc = y[i];
            for (int j = 0; j < s; ++j) acc += h * TAB_A[s][j] * k[j][i];
            y_temp[i] = acc;
        }
        hlm_oracle_rhs(uid, t + TAB_C[s] * h, y_temp, k[s], sp, F, nforc);
    }
    for (int i = 0; i < n; ++i) { /* :111-117 */
        double acc = y[i];
        for (int s = 0; s < 7; ++s) acc += h * TAB_B[s] * k[s][i];
        y_out[i] = acc;
    }
    double max_ratio = 0.0; /* :123-142; "if (ratio > max)" ignores NaN ratios */
    for (int i = 0; i < n; ++i) {
        double y_err = 0.0;
        for (int s = 0; s < 7; ++s) y_err += h * (TAB_B[s] - TAB_BALT[s]) * k[s][i];
        const double ymax = fmax(fabs(y[i]), fabs(y_out[i]));
        const double tol = atol + rtol * ymax;
        const double ratio = fabs(y_err / tol);
        if (ratio > max_ratio) max_ratio = ratio;
    }
    *err_norm = max_ratio;
}

/* rk45_step_dense.cuh:222-243 */
static void rk45_dense_n(const double* y_n, double k[7][NEQ], double h, double theta, double* y_dense, int n);
void hlm_oracle_rk45_dense(const double* y_n, double k[7][NEQ], double h, double theta,
                           double* y_dense)
{
    rk45_dense_n(y_n, k, h, theta, y_dense, NEQ);
}
static void rk45_dense_n(const double* y_n, double k[7][NEQ], double h, double theta, double* y_dense, int n)
{
    double Q[4][NEQ];
    for (int m = 0; m < 4; ++m)
        for (int i = 0; i < n; ++i) {
            double sum = 0.0;
            for (int j = 0; j < 7; ++j) sum += TAB_P[j][m] * k[j][i];
            Q[m][i] = sum;
        }
    for (int i = 0; i < n; ++i) {
        double poly = 0.0, thp = theta;
        for (int m = 0; m < 4; ++m) {
            poly += Q[m][i] * thp;
            thp *= theta;
        }
        y_dense[i] = y_n[i] + h * poly;
    }
}

/* event_detector.cuh:46-53 */
static double norm_inf_diff(const double* a, const double* b, int n)
{
    double m = 0.0;
    for (int i = 0; i < n; ++i) {
        const double f = fabs(a[i] - b[i]);
        if (f > m) m = f;
    }
    return m;
}

/* One system, rk45_kernel.cu:44-175. */
static void integrate_one(const hlm_oracle_job* J, int sys)
{
    const hlm_oracle_solver_params* P = &J->p;
    const uint32_t cap = J->attempt_cap ? J->attempt_cap : 0xFFFFFFFFu;
    const int fixed = J->fixed_step > 0.0;
    const int n = hlm_oracle_neq(J->uid);
    double y[NEQ], y_next[NEQ], k[7][NEQ], err;
    for (int i = 0; i < n; ++i) y[i] = J->y0[(size_t)sys * n + i];

    int next_q = 0, reject_count = 0, stiff = 0, capped = 0, paused = 0;
    uint32_t n_attempt = 0, n_accept = 0;
    double t = J->start_t ? J->start_t[sys] : J->t0;   /* (segment replay: see hlm_oracle.h) */
    double h = fixed ? J->fixed_step : (J->start_h ? J->start_h[sys] : P->initialStep);

    while (t < J->tf && !stiff) {
        if (J->use_t_stop && t >= J->t_stop) { paused = 1; break; }
        if (n_attempt >= cap) { capped = 1; break; }
        if (t + h > J->tf) h = J->tf - t; /* :54 */

        /* forcing gather at the step's START time, held for all stages, :85-106 */
        double F[HLM_ORACLE_MAX_FORCINGS];
        for (int j = 0; j < J->nforc; ++j) {
            if (J->forc_stub_f64) { F[j] = J->forc_stub_f64[j]; continue; }
            const double dt_min = J->forc_dt_h[j] * 60.0;
            const double r = t / dt_min;
            const size_t nS = (size_t)J->forc_nT[j];
            const size_t idx = (r < 0.0) ? 0 : (r >= (double)nS ? nS - 1 : (size_t)r);
            size_t base = 0;
            for (int kk = 0; kk < j; ++kk) base += (size_t)J->forc_nT[kk] * (size_t)J->ns;
            F[j] = (double)J->forc[base + idx * (size_t)J->ns + (size_t)sys];
        }
        const hlm_oracle_spatial_params* sp = J->sp ? &J->sp[sys] : NULL;

        ++n_attempt;
        hlm_oracle_rhs(J->uid, t, y, k[0], sp, F, J->nforc); /* :114, no FSAL */
        hlm_oracle_rk45_step(J->uid, t, y, h, P->rtol, P->atol, sp, F, J->nforc, k, y_next, &err);

        if (fixed || err <= 1.0) { /* :129 */
            reject_count = 0;
            if (!fixed) {
                const double jump = norm_inf_diff(k[0], k[1], n); /* :132-136 */
                if (jump > SLOPE_JUMP_THRESH) {
                    h = fmax(h * 0.5, P->initialStep * MIN_STEP_FRACTION);
                    continue;
                }
            }
            const double t1 = t + h;
            while (next_q < J->nq && J->query[next_q] <= t1) { /* :139-148 */
                const double tq = J->query[next_q];
                if (tq > t) {
                    const double th = (tq - t) / h;
                    double yd[NEQ];
                    rk45_dense_n(y, k, h, th, yd, n);
                    for (int c = 0; c < n; ++c)
                        J->dense_raw[(size_t)sys * n * (size_t)J->nq + (size_t)c * (size_t)J->nq + (size_t)next_q] = yd[c];
                }
                ++next_q;
            }
            for (int i = 0; i < n; ++i) y[i] = y_next[i];
            t = t1;
            ++n_accept;
            if (!fixed) { /* :151-152 */
                const double fac = P->safety * pow(1.0 / (err + 1e-16), 0.2);
                h *= fmin(fmax(fac, P->minScale), P->maxScale);
            } else {
                h = J->fixed_step;
            }
        } else { /* :153-163 */
            ++reject_count;
            double fac = P->safety * pow(1.0 / (err + 1e-16), 0.2);
            fac = fmin(fac, 1.0);
            fac = fmin(fmax(fac, P->minScale), P->maxScale);
            h *= fac;
            if (reject_count > 5 || h < (J->tf - J->t0) * MIN_STEP_FRACTION) stiff = 1;
        }
    }

    if (J->n_attempt) J->n_attempt[sys] = n_attempt;
    if (J->n_accept) J->n_accept[sys] = n_accept;
    if (J->t_reached) J->t_reached[sys] = t;
    if (J->h_reached) J->h_reached[sys] = h;
    if (J->y_reached)
        for (int i = 0; i < n; ++i) J->y_reached[(size_t)sys * n + i] = y[i];
    if (paused) { J->stiff[sys] = HLM_ORACLE_PAUSED; return; }
    if (capped && t < J->tf) { J->stiff[sys] = HLM_ORACLE_CAPPED; return; }
    if (stiff && t < J->tf) { J->stiff[sys] = HLM_ORACLE_STIFF; return; } /* :167-170 */
    for (int i = 0; i < n; ++i) J->y_final[(size_t)sys * n + i] = y[i]; /* :173-175 */
}

int hlm_oracle_rk45_run(const hlm_oracle_job* J)
{
    if (!J || J->ns < 0 || J->nq < 0 || J->nforc < 0 || J->nforc > HLM_ORACLE_MAX_FORCINGS) return -1;
    if (J->ns > 0 && (!J->y0 || !J->y_final || !J->stiff)) return -1;
    if (J->nq > 0 && (!J->query || !J->dense_raw)) return -1;
    const int n = hlm_oracle_neq(J->uid);
    if (n == 0) return -1;
    if (J->uid == HLM_ORACLE_UID_204 && J->ns > 0 && !J->sp) return -1;
    if (J->nforc > 0 && !J->forc_stub_f64 && (!J->forc || !J->forc_dt_h || !J->forc_nT)) return -1;

    /* SURVEY A.1 note 11: never-written slots read as 0 in every committed artifact. */
    memset(J->y_final, 0, sizeof(double) * (size_t)J->ns * n);
    memset(J->stiff, 0, sizeof(int32_t) * (size_t)J->ns);
    if (J->nq > 0) memset(J->dense_raw, 0, sizeof(double) * (size_t)J->ns * n * (size_t)J->nq);

    int sys;
#ifdef _OPENMP
    const int nt = J->nthreads > 0 ? J->nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
#endif
    for (sys = 0; sys < J->ns; ++sys) integrate_one(J, sys);
    return 0;
}

void hlm_oracle_reorder_dense_n(const double* raw, double* out, int ns, int nq, int n)
{
    for (size_t s = 0; s < (size_t)ns; ++s)
        for (size_t q = 0; q < (size_t)nq; ++q)
            for (size_t c = 0; c < (size_t)n; ++c)
                out[(s * (size_t)nq + q) * (size_t)n + c] = raw[s * (size_t)n * (size_t)nq + c * (size_t)nq + q];
}
void hlm_oracle_reorder_dense(const double* raw, double* out, int ns, int nq) { hlm_oracle_reorder_dense_n(raw, out, ns, nq, NEQ); }
