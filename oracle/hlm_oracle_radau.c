/*
 * hlm_oracle_radau.c -- CPU restatement of the Radau-IIA fallback.  TEST INFRASTRUCTURE ONLY
 * (see hlm_oracle.h).  The product's HIP kernel (csrc/hlm_radau_kernel.h) follows this text
 * operation for operation.
 */
#include <math.h>
#include <string.h>

#include "hlm_oracle.h"

#define NEQ HLM_ORACLE_NEQ
#define NS 3

#define SQRT6 2.449489742783178
/* Radau IIA, 3 stages, order 5 (radau_step_dense.cuh:57-67 holds the same A and c).  With
 * Lambda = TI A^-1 T = [[g,0,0],[0,a,b],[0,-b,a]] the 15x15 Newton system
 *   (I - h A (x) J) dZ = -(Z - h (A (x) I) F(Z))
 * becomes, for W = (TI (x) I) Z, one real and one complex 5x5 system (Hairer & Wanner, Solving
 * ODEs II, IV.8):
 *   (g/h I - J) dW0            = (TI F)_0 - (g/h) W0
 *   ((a - ib)/h I - J)(dW1 + i dW2) = (TI F)_1 + i (TI F)_2 - ((a - ib)/h)(W1 + i W2)          */
static const double RC[NS] = {(4.0 - SQRT6) / 10.0, (4.0 + SQRT6) / 10.0, 1.0};
#define MU_REAL 3.637834252744496    /* g = 3 + 3^(2/3) - 3^(1/3) */
#define MU_CRE 2.6810828736277523    /* a = 3 + (3^(1/3) - 3^(2/3)) / 2 */
#define MU_CIM 3.050430199247411     /* b = sqrt(3) (3^(1/3) + 3^(2/3)) / 2; the eigenvalue used is a - ib */
static const double RT[NS][NS] = {{0.09443876248897524, -0.1412552950209542, 0.03002919410514742},
                                  {0.2502131229653333, 0.20412935229379994, -0.3829421127572619},
                                  {1.0, 1.0, 0.0}};
static const double RTI[NS][NS] = {{4.178718591551904, 0.32768282076106237, 0.5233764454994495},
                                   {-4.178718591551904, -0.32768282076106237, 0.47662355450055044},
                                   {0.5028726349457868, -2.571926949855605, 0.5960392048282249}};
/* RADAU5 error estimator: err = (g/h I - J)^-1 (f0 + E.Z / h) */
static const double RE[NS] = {(-13.0 - 7.0 * SQRT6) / 3.0, (-13.0 + 7.0 * SQRT6) / 3.0, -1.0 / 3.0};
/* collocation polynomial: y(t + theta h) = y + sum_m Q_m theta^(m+1), Q_m = sum_i P[i][m] Z_i */
static const double RP[NS][3] = {
    {13.0 / 3.0 + 7.0 * SQRT6 / 3.0, -23.0 / 3.0 - 22.0 * SQRT6 / 3.0, 10.0 / 3.0 + 5.0 * SQRT6},
    {13.0 / 3.0 - 7.0 * SQRT6 / 3.0, -23.0 / 3.0 + 22.0 * SQRT6 / 3.0, 10.0 / 3.0 - 5.0 * SQRT6},
    {1.0 / 3.0, -8.0 / 3.0, 10.0 / 3.0}};

#define NEWTON_MAXITER 10   /* radau_step_dense.cuh:87 */
#define NEWTON_TOL 1e-3     /* in units of the error tolerance (RADAU5: min(0.03, sqrt(rtol))) */
#define FD_EPS 1e-8         /* radau_step_dense.cuh:20: sqrt(1e-16) */

/* real 5x5 LU with partial pivoting; the reciprocal pivots are kept for the solves.
 * Returns 0 if a pivot is (numerically) zero. */
static int lu_real(double a[NEQ][NEQ], int piv[NEQ], double inv[NEQ], int n)
{
    int ok = 1;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(a[k][k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(a[i][k]) > best) { best = fabs(a[i][k]); p = i; }
        piv[k] = p;
        if (!(best > 1e-300)) ok = 0;
        for (int j = 0; j < n; ++j) { const double tmp = a[k][j]; a[k][j] = a[p][j]; a[p][j] = tmp; }
        inv[k] = 1.0 / a[k][k];
        for (int i = k + 1; i < n; ++i) {
            const double m = a[i][k] * inv[k];
            a[i][k] = m;
            for (int j = k + 1; j < n; ++j) a[i][j] -= m * a[k][j];
        }
    }
    return ok;
}

static void solve_real(double a[NEQ][NEQ], const int piv[NEQ], const double inv[NEQ], double b[NEQ], int n)
{
    for (int k = 0; k < n; ++k) {
        const double tmp = b[k]; b[k] = b[piv[k]]; b[piv[k]] = tmp;
        for (int i = k + 1; i < n; ++i) b[i] -= a[i][k] * b[k];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < n; ++j) s -= a[i][j] * b[j];
        b[i] = s * inv[i];
    }
}

/* the same for a complex matrix held as (re, im); pivot on |re| + |im| */
static int lu_cplx(double ar[NEQ][NEQ], double ai[NEQ][NEQ], int piv[NEQ], double ir[NEQ], double ii[NEQ], int n)
{
    int ok = 1;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(ar[k][k]) + fabs(ai[k][k]);
        for (int i = k + 1; i < n; ++i) {
            const double v = fabs(ar[i][k]) + fabs(ai[i][k]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (!(best > 1e-300)) ok = 0;
        for (int j = 0; j < n; ++j) {
            double tmp = ar[k][j]; ar[k][j] = ar[p][j]; ar[p][j] = tmp;
            tmp = ai[k][j]; ai[k][j] = ai[p][j]; ai[p][j] = tmp;
        }
        const double d = 1.0 / (ar[k][k] * ar[k][k] + ai[k][k] * ai[k][k]);
        ir[k] = ar[k][k] * d;
        ii[k] = -ai[k][k] * d;
        for (int i = k + 1; i < n; ++i) {
            const double mr = ar[i][k] * ir[k] - ai[i][k] * ii[k];
            const double mi = ar[i][k] * ii[k] + ai[i][k] * ir[k];
            ar[i][k] = mr;
            ai[i][k] = mi;
            for (int j = k + 1; j < n; ++j) {
                ar[i][j] -= mr * ar[k][j] - mi * ai[k][j];
                ai[i][j] -= mr * ai[k][j] + mi * ar[k][j];
            }
        }
    }
    return ok;
}

static void solve_cplx(double ar[NEQ][NEQ], double ai[NEQ][NEQ], const int piv[NEQ], const double ir[NEQ],
                       const double ii[NEQ], double br[NEQ], double bi[NEQ], int n)
{
    for (int k = 0; k < n; ++k) {
        double tmp = br[k]; br[k] = br[piv[k]]; br[piv[k]] = tmp;
        tmp = bi[k]; bi[k] = bi[piv[k]]; bi[piv[k]] = tmp;
        for (int i = k + 1; i < n; ++i) {
            br[i] -= ar[i][k] * br[k] - ai[i][k] * bi[k];
            bi[i] -= ar[i][k] * bi[k] + ai[i][k] * br[k];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double sr = br[i], si = bi[i];
        for (int j = i + 1; j < n; ++j) {
            sr -= ar[i][j] * br[j] - ai[i][j] * bi[j];
            si -= ar[i][j] * bi[j] + ai[i][j] * br[j];
        }
        br[i] = sr * ir[i] - si * ii[i];
        bi[i] = sr * ii[i] + si * ir[i];
    }
}

static void radau_one(const hlm_oracle_job* J, int sys)
{
    const hlm_oracle_solver_params* P = &J->p;
    const uint32_t cap = J->attempt_cap ? J->attempt_cap : 0xFFFFFFFFu;
    const hlm_oracle_spatial_params* sp = J->sp ? &J->sp[sys] : NULL;
    const size_t nq = (size_t)J->nq;
    double* dense = J->nq ? J->dense_raw + (size_t)sys * hlm_oracle_neq(J->uid) * nq : NULL;
    const int n = hlm_oracle_neq(J->uid);
    double y[NEQ];
    double t = J->t0, h = P->initialStep;
    if (J->radau_continue) {
        /* continue from where RK45 gave up: its rows up to there (queries <= t_reached) stay as they are, the
         * later ones are still zero */
        t = J->t_reached[sys];
        for (int i = 0; i < n; ++i) y[i] = J->y_reached[(size_t)sys * n + i];
    } else {
        for (int i = 0; i < n; ++i) y[i] = J->y0[(size_t)sys * n + i];
        if (dense) memset(dense, 0, sizeof(double) * n * nq);
    }
    int next_q = 0, capped = 0;
    uint32_t n_attempt = 0, n_accept = 0;
    while (next_q < J->nq && J->query[next_q] <= t) ++next_q; /* never produced: stay 0 */
    /* diagnostic trace (hlm_oracle.h): one byte per attempt */
    uint8_t* const trace = J->radau_trace ? J->radau_trace + (size_t)sys * (size_t)J->radau_trace_cap : NULL;
#define TRACE(byte) do { if (trace && (int64_t)n_attempt <= J->radau_trace_cap) trace[n_attempt - 1] = (uint8_t)(byte); } while (0)

    /* STEP MEMORY ACROSS FORCING CHANGES.  A step that lands on a sample boundary keeps the controller's proposal (h_ctl
     * below), typically the whole sample interval; when the next sample differs, the store it feeds starts a transient of
     * minutes and that proposal is rejected three or four times in a row (min scale 0.2 each) before a step fits --
     * measured on SURVEY's cold cells: 10.6 % of all implicit attempts are these rejects.  The first attempt after a
     * CHANGE of any held sample therefore starts from the step that was accepted first after the previous change
     * (h_change; none yet: the proposal stands): rejected attempts 12 % -> 1.5 %, attempts per system -6 %.  Every
     * accepted step still passes the error test; only the controller's first guess differs. */
    double F_held[HLM_ORACLE_MAX_FORCINGS], h_change = 0.0;
    int have_held = 0, change_pending = 0;

    while (t < J->tf) {
        if (n_attempt >= cap) { capped = 1; break; }

        /* forcing at the step's start, held over the step (as rk45_kernel.cu:85-106); unlike the
         * explicit loop an implicit step can span many samples, so the step is also clipped to the
         * next sample boundary t_b of any stream (the hold is then exact, not an approximation) */
        double F[HLM_ORACLE_MAX_FORCINGS], t_b = INFINITY;
        for (int j = 0; j < J->nforc; ++j) {
            if (J->forc_stub_f64) { F[j] = J->forc_stub_f64[j]; continue; }
            const double dt_min = J->forc_dt_h[j] * 60.0, r = t / dt_min;
            const size_t nS = (size_t)J->forc_nT[j];
            size_t idx = (r < 0.0) ? 0 : (r >= (double)nS ? nS - 1 : (size_t)r);
            /* nudge: a step clipped to a boundary lands exactly on it; make sure the interval
             * of idx really contains t even if t / dt_min rounded below the integer */
            while (idx + 1 < nS && (double)(idx + 1) * dt_min <= t) ++idx;
            size_t base = 0;
            for (int kk = 0; kk < j; ++kk) base += (size_t)J->forc_nT[kk] * (size_t)J->ns;
            F[j] = (double)J->forc[base + idx * (size_t)J->ns + (size_t)sys];
            if (idx + 1 < nS) t_b = fmin(t_b, (double)(idx + 1) * dt_min);
        }
        if (!J->forc_stub_f64) {
            int changed = 0;
            for (int j = 0; j < J->nforc; ++j) {
                if (have_held && memcmp(&F[j], &F_held[j], sizeof(double)) != 0) changed = 1; /* (by bits: a NaN sample that stays is no change) */
                F_held[j] = F[j];
            }
            have_held = 1;
            if (changed) {
                if (h_change > 0.0 && h > h_change) h = h_change;
                change_pending = 1;
            }
        }
        const double h_ctl = h; /* the controller's proposal, restored after a clipped step */
        double t_end = t + h;   /* the step lands exactly on tf / t_b when clipped to them */
        int clipped = 0, on_boundary = 0;
        if (t_end > J->tf) { t_end = J->tf; clipped = 1; }
        if (t_end > t_b) { t_end = t_b; clipped = 1; on_boundary = 1; }
        if (clipped) h = t_end - t;
        ++n_attempt;

        /* f0 and the Jacobian at (t, y).  The reference differentiates numerically
         * (approx_jacobian, radau_step_dense.cuh:13-31: delta_j = 1e-8 max(1, |y_j|), 5 more RHS
         * calls); the models here supply the exact one-sided derivative (hlm_oracle_jacobian). */
        double f0[NEQ], Jm[NEQ][NEQ];
        hlm_oracle_rhs(J->uid, t, y, f0, sp, F, J->nforc);
        if (hlm_oracle_has_jacobian(J->uid)) {
            hlm_oracle_jacobian(J->uid, y, Jm, sp, F, J->nforc);
        } else { /* approx_jacobian, radau_step_dense.cuh:13-31 */
            for (int j = 0; j < n; ++j) {
                double yp[NEQ], f1[NEQ];
                const double d = FD_EPS * fmax(1.0, fabs(y[j]));
                for (int i = 0; i < n; ++i) yp[i] = y[i];
                yp[j] += d;
                hlm_oracle_rhs(J->uid, t, yp, f1, sp, F, J->nforc);
                for (int i = 0; i < n; ++i) Jm[i][j] = (f1[i] - f0[i]) / d;
            }
        }

        /* LU of (g/h) I - J (Newton's real block AND the error estimator) and of ((a - ib)/h) I - J */
        double Mr[NEQ][NEQ], Cr[NEQ][NEQ], Ci[NEQ][NEQ], inv_r[NEQ], inv_cr[NEQ], inv_ci[NEQ];
        int piv_r[NEQ], piv_c[NEQ];
        const double g_h = MU_REAL / h, a_h = MU_CRE / h, b_h = MU_CIM / h;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                Mr[i][j] = ((i == j) ? g_h : 0.0) - Jm[i][j];
                Cr[i][j] = ((i == j) ? a_h : 0.0) - Jm[i][j];
                Ci[i][j] = (i == j) ? -b_h : 0.0;
            }
        int ok = lu_real(Mr, piv_r, inv_r, n);
        ok = lu_cplx(Cr, Ci, piv_c, inv_cr, inv_ci, n) && ok;

        /* simplified Newton on W = (TI (x) I) Z, Z_i = Y_i - y */
        double Z[NS][NEQ], W[NS][NEQ];
        memset(Z, 0, sizeof(Z));
        memset(W, 0, sizeof(W));
        int converged = 0, its = 0;
        double prev = INFINITY;
        for (int it = 0; ok && it < NEWTON_MAXITER; ++it) {
            double Fs[NS][NEQ], d0[NEQ], d1[NEQ], d2[NEQ];
            ++its;
            for (int s = 0; s < NS; ++s) {
                double Y[NEQ];
                for (int i = 0; i < n; ++i) Y[i] = y[i] + Z[s][i];
                hlm_oracle_rhs(J->uid, t + RC[s] * h, Y, Fs[s], sp, F, J->nforc);
            }
            for (int i = 0; i < n; ++i) {
                d0[i] = (RTI[0][0] * Fs[0][i] + RTI[0][1] * Fs[1][i] + RTI[0][2] * Fs[2][i]) - g_h * W[0][i];
                d1[i] = (RTI[1][0] * Fs[0][i] + RTI[1][1] * Fs[1][i] + RTI[1][2] * Fs[2][i]) - (a_h * W[1][i] + b_h * W[2][i]);
                d2[i] = (RTI[2][0] * Fs[0][i] + RTI[2][1] * Fs[1][i] + RTI[2][2] * Fs[2][i]) - (a_h * W[2][i] - b_h * W[1][i]);
            }
            solve_real(Mr, piv_r, inv_r, d0, n);
            solve_cplx(Cr, Ci, piv_c, inv_cr, inv_ci, d1, d2, n);
            double nrm = 0.0, chk = 0.0;
            for (int i = 0; i < n; ++i) {
                chk += d0[i] + d1[i] + d2[i];
                W[0][i] += d0[i];
                W[1][i] += d1[i];
                W[2][i] += d2[i];
                const double sc = P->atol + P->rtol * fabs(y[i]);
                const double r = fmax(fmax(fabs(d0[i]), fabs(d1[i])), fabs(d2[i])) / sc;
                if (r > nrm) nrm = r;
                for (int s = 0; s < NS; ++s) Z[s][i] = RT[s][0] * W[0][i] + RT[s][1] * W[1][i] + RT[s][2] * W[2][i];
            }
            if (!(chk - chk == 0.0)) break;              /* a NaN / Inf increment */
            if (it > 0 && nrm > 2.0 * prev) break;       /* diverging */
            prev = nrm;
            if (nrm <= NEWTON_TOL) { converged = 1; break; }
        }
        if (!converged) { TRACE(its | 0x20); h *= 0.5; continue; }

        /* stiffly accurate: y_new = y + Z_3; RADAU5 error estimate */
        double y_new[NEQ], ev[NEQ];
        for (int i = 0; i < n; ++i) {
            y_new[i] = y[i] + Z[2][i];
            ev[i] = f0[i] + (RE[0] * Z[0][i] + RE[1] * Z[1][i] + RE[2] * Z[2][i]) / h;
        }
        solve_real(Mr, piv_r, inv_r, ev, n);
        double err = 0.0;
        for (int i = 0; i < n; ++i) {
            const double r = fabs(ev[i]) / (P->atol + P->rtol * fmax(fabs(y[i]), fabs(y_new[i])));
            if (r > err) err = r;
        }
        if (!(err == err)) { TRACE(its | 0x20); h *= 0.5; continue; } /* NaN estimate: treat like a failed Newton */
        const double fac = P->safety * pow(1.0 / (err + 1e-16), 0.25);
        if (err <= 1.0) {
            const double t1 = t_end;
            while (next_q < J->nq && J->query[next_q] <= t1) { /* collocation polynomial */
                const double th = (J->query[next_q] - t) / h;
                for (int c = 0; c < n; ++c) {
                    double acc = y[c], thp = th;
                    for (int m = 0; m < 3; ++m) {
                        acc += (RP[0][m] * Z[0][c] + RP[1][m] * Z[1][c] + RP[2][m] * Z[2][c]) * thp;
                        thp *= th;
                    }
                    dense[(size_t)c * nq + (size_t)next_q] = acc;
                }
                ++next_q;
            }
            for (int i = 0; i < n; ++i) y[i] = y_new[i];
            t = t1;
            ++n_accept;
            TRACE(its | (on_boundary ? 0x10 : 0));
            if (change_pending) { h_change = h; change_pending = 0; } /* the first step the new sample accepted */
            h *= fmin(fmax(fac, P->minScale), P->maxScale);
            if (clipped && h < h_ctl) h = h_ctl;
        } else {
            TRACE(its | 0x40);
            h *= fmin(fmax(fmin(fac, 1.0), P->minScale), P->maxScale);
        }
    }
#undef TRACE

    if (J->n_attempt) J->n_attempt[sys] += n_attempt;
    if (J->n_accept) J->n_accept[sys] += n_accept;
    if (J->t_reached) J->t_reached[sys] = t;
    if (capped && t < J->tf) {
        J->stiff[sys] = HLM_ORACLE_CAPPED;
        for (int i = 0; i < n; ++i) J->y_final[(size_t)sys * n + i] = 0.0;
        return;
    }
    for (int i = 0; i < n; ++i) J->y_final[(size_t)sys * n + i] = y[i];
}

int hlm_oracle_radau_run(const hlm_oracle_job* J)
{
    if (!J || J->ns < 0 || !J->stiff || !J->y_final || !J->y0) return -1;
    if (J->nq > 0 && (!J->query || !J->dense_raw)) return -1;
    if (J->radau_continue && (!J->t_reached || !J->y_reached)) return -1;
    int sys;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (sys = 0; sys < J->ns; ++sys)
        if (J->stiff[sys] == HLM_ORACLE_STIFF) radau_one(J, sys);
    return 0;
}
