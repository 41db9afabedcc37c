// Device-side model plug-ins for the RK45 kernel (one struct per Model::UID).
//
// A device model provides
//   UID, N_EQ (state variables; <= HLM_MAX_NEQ), NF (how many forcings rhs reads; <= HLM_KERNEL_MAX_NF),
//   struct Local (per-system constants kept in VGPRs for the whole integration), struct Forc (values
//   derived from the held forcing samples), load(sp, sys), prep_forcing(P, F[NF]), rhs<MATH>(P, F, y,
//   dydt) and optionally (HAS_FASTPATH / HAS_NORAINPATH) cheaper RHS variants that are exact while a
//   per-lane precondition holds, (HAS_JACOBIAN) the exact d(rhs)/dy for the Radau-IIA pass.
// The kernels are templates over the model: N_EQ and NF are compile-time, every state array lives in
// named registers (reference: rk45_kernel_multi<Model>, solver/rk45.h:51-66, Model::N_EQ).
// This is the device half of the reference's plug-in concept (Model::UID / SP_TYPE /
// rhs, reference src/models/model_204.hpp:15-115); the host half is include/hlm/models/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hlm_hip.h"
#include "hlm_strict_pow.h"

namespace hlm {

constexpr int HLM_KERNEL_MAX_NF = 4;  // forcings a model's rhs may read (kernel-argument arrays are this long)
constexpr int MATH_STRICT = HLM_MATH_STRICT;
constexpr int MATH_FAST = HLM_MATH_FAST;

// IEEE minNum / maxNum as single VALU instructions.  fmin()/fmax() lower to
// llvm.minnum/maxnum, for which hipcc (IEEE mode on) first quiets possibly-signalling
// inputs with an extra "v_max_f64 x, x, x" whenever it cannot prove the operand came out
// of an arithmetic instruction (27 of them per loop iteration in this kernel).  No
// signalling NaN can reach these values (loads of user doubles are only ever combined
// arithmetically first), and for quiet NaNs the bare instruction already returns the
// non-NaN operand -- the behaviour the reference relies on for fmin(1.0, NaN) == 1.0.
__device__ __forceinline__ double vmin(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// fmin(fmax(x, lo), hi) with WAVE-UNIFORM bounds (kernel arguments): two instructions, each bound as the one scalar
// operand a VOP3 instruction may take.  Written with fmin / fmax the compiler first quiets all three operands
// ("v_max_f64 lo, lo, lo" ...): three more VALU instructions per accepted step.  Same NaN behaviour as those
// functions: a NaN x comes out as lo.
__device__ __forceinline__ double vclamp_uniform(double x, double lo, double hi)
{
    double a, r;
    asm("v_max_f64 %0, %1, %2" : "=v"(a) : "v"(x), "s"(lo));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(hi));
    return r;
}
// max(|a|, |b|) with the absolute values as source modifiers
__device__ __forceinline__ double vmax_abs(double a, double b)
{
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max(m, |b|): "if (fabs(b) > m) m = fabs(b)" of the reference's norm loops
// (rk45_step_dense.cuh:135-140, event_detector.cuh:46-53); like those comparisons it
// ignores a NaN b.
__device__ __forceinline__ double vmax_abs2(double m, double b)
{
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(b));
    return r;
}

// n / d for a divisor whose correctly rounded reciprocal rd = RN(1/d) is at hand (per-system
// constants: Hu, alpha3, alpha4, A_h): q = RN(n*rd) is within 2 ulp of n/d, the residual
// n - q*d is exact in one FMA, and q + rem*rd rounds to RN(n/d) (Markstein's division step).
// Three FMA-class instructions instead of the ~11 + v_rcp_f64 of the IEEE expansion.
// Verified bit-for-bit against '/' on the GPU for 4e8 operand pairs incl. adversarial
// significands (tests/test_parity_gpu.py::test_division_by_invariant_is_ieee_exact); operands
// and quotient must be normal, finite and clear of the overflow/underflow margins, which the
// caller guarantees by falling back to '/' otherwise.
__device__ __forceinline__ double div_by_invariant(double n, double d, double rd)
{
    const double q = n * rd;
    const double rem = __builtin_fma(-q, d, n);
    return __builtin_fma(rem, rd, q);
}

// IEEE n / d for STRICT mode: the 3-instruction form when it is provably exact, '/' otherwise
// (zero, subnormal, huge, Inf or NaN numerators; a divisor outside [1e-100, 1e100]).
__device__ __forceinline__ double strict_div(double n, double d, double rd, bool divisor_ok)
{
    const double an = fabs(n);
    if (divisor_ok && an >= 1e-290 && an <= 1e290) return div_by_invariant(n, d, rd);
    return n / d;
}
__device__ __forceinline__ bool divisor_in_safe_range(double d)
{
    const double ad = fabs(d);
    return ad >= 1e-100 && ad <= 1e100;
}

// x^(-1/n) refinement shared by the FAST roots: given a seed w ~ x^(-1/n) with residual
// e = 1 - x w^n, the exact root is w (1-e)^(-1/n) = w (1 + c1 e + c2 e^2 + c3 e^3 + ...).

// x^(2/3) for the Manning term (model_204.hpp:103).
// STRICT: the correctly rounded power for the double exponent 0.6666666666666666 (hlm_strict_pow.h; what glibc's pow
//         returns in 999 of 1 000 cases -- the reference makes the library call, through libdevice on the GPU).
// FAST:   x * x^(-1/3) with x^(-1/3) from an integer seed and two series refinements, all in
//         ordinary VALU instructions -- no f32 log/exp (quarter rate), no frexp/ldexp/convert
//         shuffling:
//           seed   hi32(w0) = 0x553ee979 - hi32(x)/3          |1 - x w0^3| <= 0.103 (all normal x)
//           step 1 w1 = w0 (1 + e0/3 + 2/9 e0^2 + 14/81 e0^3 + 35/243 e0^4)   |e1| <= 4.4e-6
//           step 2 r  = x w1 (1 + e1/3 + 2/9 e1^2)                            remainder 14/81 e1^3 <= 1.5e-17
//         Result within 2 ulp of cbrt(x)^2 (hlm_selftest_math, tests).  x < 0 must stay NaN like
//         pow() so that fmin(1.0, NaN) == 1.0 keeps the reference's behaviour for a negative
//         surface store (SURVEY A.1 note 6); zero and subnormal x give a value <= 1.8e-205 (the true
//         one is below that as well; the RHS multiplies it by x again, so x == 0 still contributes 0).
template <int MATH>
__device__ __forceinline__ double pow_two_thirds(double x)
{
    if constexpr (MATH == MATH_STRICT) {
        return strict_pow_two_thirds(x);
    } else {
        // zero and subnormal arguments (the seed would be out of range) are lifted to the smallest normal
        // number: the result is then <= 1.8e-205 instead of the true value, which is below that too -- one
        // v_max_f64 instead of a compare and two selects; a NaN argument is lifted as well, harmlessly
        // (the caller multiplies the result by the argument again)
        const double xs = vmax(x, 2.3e-308);
        const unsigned hi = (unsigned)__double2hiint(xs);
        const unsigned whi = 0x553ee979u - __umulhi(hi, 0x55555556u);
        const double w0 = __hiloint2double((int)whi, 0);
        const double e0 = __builtin_fma(-xs, w0 * w0 * w0, 1.0);
        double p = __builtin_fma(e0, 35.0 / 243.0, 14.0 / 81.0);
        p = __builtin_fma(e0, p, 2.0 / 9.0);
        p = __builtin_fma(e0, p, 1.0 / 3.0);
        const double w1 = __builtin_fma(w0 * e0, p, w0);
        const double e1 = __builtin_fma(-xs, w1 * w1 * w1, 1.0);   // |e1| <= 4.4e-6
        const double q = __builtin_fma(e1, 2.0 / 9.0, 1.0 / 3.0);  // the next term, 14/81 e1^3 <= 1.5e-17, is dropped
        const double xw = xs * w1;
        const double r = __builtin_fma(xw * e1, q, xw);
        return x < 0.0 ? __builtin_nan("") : r;
    }
}

// ------------------------------------------------------------------------------------
// Model 204: snow / static / surface / gravitational / aquifer stores.
// Reference: src/models/model_204.hpp:43-114; parameter record
// src/I_O/parameters_loader.hpp:19-37 (136-byte AoS, read once per system here).
// ------------------------------------------------------------------------------------
struct Model204Dev {
    static constexpr int UID = HLM_UID_204;
    static constexpr int N_EQ = 5;  // Model204::N_EQ, model_204.hpp:19
    static constexpr int NF = 2;    // F[0] = rainfall, F[1] = temperature (:82-83)

    struct Local {
        double infil, perco, Hu, L, A_h, alpha3, alpha4, melt_f, temp_thr;
        double inv_n;       // 1.0 / n_mann   (same operation as :103, hoisted: bit-identical)
        double sqrt_slope;  // sqrt(slope)    (idem)
        // FAST: reciprocals / products replacing the per-call divides.  STRICT: inv_* are the
        // correctly rounded reciprocals RN(1/d) that strict_div() needs for its exact 3-FMA form.
        double inv_Hu, manning, inv_alpha3, inv_alpha4, inv_A_h;
        bool div_ok;  // STRICT: all four divisors are within [1e-100, 1e100]
        bool nonneg;  // infil >= 0 && perco >= 0: min(0, infil) == min(0, perco) == 0 (no-rain path)
    };

    // Values that depend only on the held forcing samples; recomputed when a sample
    // changes (about once per forcing hour) instead of in each of the 7 RHS calls.  They
    // are the reference's own sub-expressions (:87-88, :96), so hoisting is bit-identical.
    struct Forc {
        double rain;     // F[0]
        double melt_pot; // temperature * melt_f
        double e_pot;    // 0.1 * temperature   (the linear ET stub, :96)
        unsigned long long warm_mask;  // all ones if temperature >= temp_thr, else 0
    };

    template <int MATH>
    static __device__ __forceinline__ Local load(const void* sp_base, int sys)
    {
        const hlm_spatial_params& S = static_cast<const hlm_spatial_params*>(sp_base)[sys];
        Local P;
        P.infil = S.infil;
        P.perco = S.perco;
        P.Hu = S.Hu;
        P.L = S.L;
        P.A_h = S.A_h;
        P.alpha3 = S.alpha3;
        P.alpha4 = S.alpha4;
        P.melt_f = S.melt_f;
        P.temp_thr = S.temp_thr;
        P.inv_n = 1.0 / S.n_mann;
        P.sqrt_slope = sqrt(S.slope);
        P.nonneg = S.infil >= 0.0 && S.perco >= 0.0;
        if constexpr (MATH == MATH_FAST) {
            P.inv_Hu = 1.0 / S.Hu;
            P.manning = P.inv_n * P.sqrt_slope * (S.L / S.A_h * 60.0);
            // the reference drains only when alpha >= 1 (:110,:113)
            P.inv_alpha3 = S.alpha3 >= 1.0 ? 1.0 / S.alpha3 : 0.0;
            P.inv_alpha4 = S.alpha4 >= 1.0 ? 1.0 / S.alpha4 : 0.0;
            P.inv_A_h = 0.0;
            P.div_ok = false;
        } else {
            P.manning = 0.0;
            P.inv_Hu = 1.0 / S.Hu;
            P.inv_alpha3 = 1.0 / S.alpha3;
            P.inv_alpha4 = 1.0 / S.alpha4;
            P.inv_A_h = 1.0 / S.A_h;
            P.div_ok = divisor_in_safe_range(S.Hu) && divisor_in_safe_range(S.alpha3) &&
                       divisor_in_safe_range(S.alpha4) && divisor_in_safe_range(S.A_h);
        }
        return P;
    }

    static __device__ __forceinline__ Forc prep_forcing(const Local& P, const double (&held)[NF])
    {
        const double rainfall = held[0], temperature = held[1];
        Forc F;
        F.rain = rainfall;
        F.melt_pot = temperature * P.melt_f;
        F.e_pot = 0.1 * temperature;
        F.warm_mask = temperature >= P.temp_thr ? ~0ull : 0ull;
        return F;
    }

    // Fast path: while the surface store is exactly 0, pow(0, 2/3) == 0 makes alfa2 == 0,
    // w == 0 and dydt[2] == d2 - 0 (model_204.hpp:103-105), so the Manning term can be
    // left out bit-exactly.  rhs_fast reports d2 != 0 (inflow: the next stage state is wet).
    static constexpr bool HAS_FASTPATH = true;
    static constexpr int FAST_ZERO_COMP = 2;  // rhs_fast: dydt[2] == 0 unless it reports "broke"
    static __device__ __forceinline__ bool fastpath_ok(const double (&y)[N_EQ]) { return y[2] == 0.0; }

    // "No-rain" path -- since round 3 really the UNSATURATED path: a dry surface store and, rain or not, a static
    // store that stays below its capacity.  While x1 + h_stat <= Hu, x2 = max(0, x1 + h_stat - Hu) = 0 takes x3, x4,
    // d2, d3 with it (model_204.hpp:94-113, given infil, perco >= 0): the cascade below the static store is not
    // evaluated at all.  rhs_norain reports x1 + h_stat > Hu (every call: that is the regime test).  The expressions
    // are the general ones with x2, x3, x4, d2, d3 replaced by the literal 0.0 they evaluate to: same values.
    // (Until round 3 the path also required F.rain == 0 and dropped the rain from x1; rain that the static store
    // absorbs -- most of it, with Hu of 100-250 -- then went the dry path's longer way.)
    static constexpr bool HAS_NORAINPATH = true;
    // On the no-rain path (x3 = x4 = 0) the gravitational and the aquifer store only drain: y' = -y / alpha, linear
    // and homogeneous -- the kernels may take them in closed form (rk_stages, CLOSED).  FAST arithmetic: the rate is
    // the reciprocal the RHS multiplies by (0 where the reference does not drain, alpha < 1).
    static constexpr int NORAIN_LINEAR = 2;
    static constexpr int NORAIN_LINEAR_COMP[2] = {3, 4};
    static __device__ __forceinline__ double norain_rate(const Local& P, int k) { return k == 0 ? -P.inv_alpha3 : -P.inv_alpha4; }
    static __device__ __forceinline__ bool norainpath_ok(const Local& P, const Forc& F, const double (&y)[N_EQ])
    {
        (void)F;  // (round 3: rain or not -- what the path needs is x2 == 0, which rhs_norain tests in every call)
        return y[2] == 0.0 && P.nonneg;
    }
    template <int MATH>
    static __device__ __forceinline__ bool rhs_norain(const Local& P, const Forc& F, const double (&y)[N_EQ],
                                                      double (&dydt)[N_EQ])
    {
        const double h_snow = y[0], h_stat = y[1], h_grav = y[3], h_aq = y[4];
        const double snowmelt =
            __longlong_as_double(__double_as_longlong(vmin(h_snow, F.melt_pot)) & (long long)F.warm_mask);
        const double x1 = F.rain + snowmelt;  // (:87-91; with no rain in the sample: 0 + snowmelt, the same value)
        dydt[0] = F.rain - snowmelt;
        // x2 = max(0, x1 + h_stat - Hu) > 0; (a - b > 0) == (a > b) exactly in IEEE arithmetic
        const bool wet = x1 + h_stat > P.Hu;
        const double Emax = vmin(F.e_pot, h_stat);
        dydt[2] = 0.0;
        if constexpr (MATH == MATH_FAST) {
            const double s = h_stat * P.inv_Hu;
            dydt[1] = x1 - s * Emax;
            dydt[3] = 0.0 - h_grav * P.inv_alpha3;
            dydt[4] = 0.0 - h_aq * P.inv_alpha4;
        } else {
            const double s = strict_div(h_stat, P.Hu, P.inv_Hu, P.div_ok);
            dydt[1] = x1 - s * Emax;
            dydt[3] = 0.0 - (P.alpha3 >= 1.0 ? strict_div(h_grav, P.alpha3, P.inv_alpha3, P.div_ok) : 0.0);
            dydt[4] = 0.0 - (P.alpha4 >= 1.0 ? strict_div(h_aq, P.alpha4, P.inv_alpha4, P.div_ok) : 0.0);
        }
        return wet;
    }

    // The unsaturated regime with a WET surface store: as rhs_norain, and the surface store drains by the Manning term
    // (model_204.hpp:101-105 with d2 = 0); the gravitational and the aquifer store still only drain (closed form).
    // FAST arithmetic, launches without dense output (attempt_per_lane_path).
    static __device__ __forceinline__ bool wetupath_ok(const Local& P, const Forc&, const double (&y)[N_EQ]) { return y[2] > 0.0 && P.nonneg; }
    template <int MATH>
    static __device__ __forceinline__ bool rhs_wetu(const Local& P, const Forc& F, const double (&y)[N_EQ], double (&dydt)[N_EQ])
    {
        const double h_snow = y[0], h_stat = y[1], h_surf = y[2], h_grav = y[3], h_aq = y[4];
        const double snowmelt =
            __longlong_as_double(__double_as_longlong(vmin(h_snow, F.melt_pot)) & (long long)F.warm_mask);
        const double x1 = F.rain + snowmelt;
        dydt[0] = F.rain - snowmelt;
        const bool wet = x1 + h_stat > P.Hu;
        const double Emax = vmin(F.e_pot, h_stat);
        const double s = h_stat * P.inv_Hu;
        dydt[1] = x1 - s * Emax;
        const double w = vmin(1.0, P.manning * pow_two_thirds<MATH>(h_surf));
        dydt[2] = 0.0 - h_surf * w;
        dydt[3] = 0.0 - h_grav * P.inv_alpha3;
        dydt[4] = 0.0 - h_aq * P.inv_alpha4;
        return wet;
    }

    template <int MATH>
    static __device__ __forceinline__ void rhs(const Local& P, const Forc& F, const double (&y)[N_EQ],
                                               double (&dydt)[N_EQ])
    {
        (void)rhs_impl<MATH, false>(P, F, y, dydt);
    }
    // For the hybrid attempt (rk_attempt_hybrid): may this hillslope take an unsaturated form at all (norainpath_ok ||
    // wetupath_ok), is its static store overflowing at this state under the held forcing -- the test rhs_norain and
    // rhs_wetu make in every call, same expression, same decision -- and the general RHS reporting that test.
    static __device__ __forceinline__ bool unsat_eligible(const Local& P, const double (&y)[N_EQ]) { return y[2] >= 0.0 && P.nonneg; }
    static __device__ __forceinline__ bool saturated_at(const Local& P, const Forc& F, const double (&y)[N_EQ])
    {
        const double snowmelt = __longlong_as_double(__double_as_longlong(vmin(y[0], F.melt_pot)) & (long long)F.warm_mask);
        const double x1 = F.rain + snowmelt;
        return x1 + y[1] > P.Hu;
    }
    template <int MATH>
    static __device__ __forceinline__ bool rhs_report_wet(const Local& P, const Forc& F, const double (&y)[N_EQ], double (&dydt)[N_EQ])
    {
        (void)rhs_impl<MATH, false>(P, F, y, dydt);
        return saturated_at(P, F, y);   // (the same sub-expressions as inside the RHS: formed once)
    }
    template <int MATH>
    static __device__ __forceinline__ bool rhs_fast(const Local& P, const Forc& F, const double (&y)[N_EQ],
                                                    double (&dydt)[N_EQ])
    {
        return rhs_impl<MATH, true>(P, F, y, dydt);
    }

    // d(rhs)/dy for the Radau-IIA pass: the exact one-sided derivative (every min / max contributes the
    // derivative of the operand it returns; w = min(1, c h^(2/3)) gives d(h w)/dh = 5/3 c h^(2/3) below
    // the cap and 1 at it, also for h < 0).  Reference: model_204.hpp:86-113 differentiated term by term.
    static constexpr bool HAS_JACOBIAN = true;
    static constexpr bool JACOBIAN_LOWER_TRIANGULAR = true;  // a cascade: store i only feels stores j <= i
    // entries below the diagonal that are zero whatever the state (jacobian() never writes them): the gravitational
    // and the aquifer store feel the snow and the static store only -- the Radau pass's forward substitutions leave
    // those products out (FAST)
    static constexpr bool jacobian_zero(int i, int j) { return (i == 3 && j == 2) || (i == 4 && (j == 2 || j == 3)); }
    template <int MATH>
    static __device__ __forceinline__ void jacobian(const Local& P, const Forc& F, const double (&y)[N_EQ],
                                                    double (&J)[N_EQ][N_EQ])
    {
#pragma unroll
        for (int i = 0; i < N_EQ; ++i)
#pragma unroll
            for (int j = 0; j < N_EQ; ++j) J[i][j] = 0.0;
        const double h_snow = y[0], h_stat = y[1], h_surf = y[2];
        const bool warm = F.warm_mask != 0ull;
        const double snowmelt = warm ? fmin(h_snow, F.melt_pot) : 0.0;
        const double m = (warm && h_snow < F.melt_pot) ? 1.0 : 0.0;
        const double x1 = F.rain + snowmelt;
        const double over = x1 + h_stat - P.Hu;
        const double x2 = fmax(0.0, over);
        const double a = over > 0.0 ? 1.0 : 0.0;
        const double Emax = fmin(F.e_pot, h_stat);
        const double e = h_stat < F.e_pot ? 1.0 : 0.0;
        const double x3 = fmin(x2, P.infil);
        const double b = x2 < P.infil ? 1.0 : 0.0;
        const double c = x3 < P.perco ? 1.0 : 0.0;
        double w_lin, s, dEmax;
        if constexpr (MATH == MATH_FAST) {
            w_lin = P.manning * pow_two_thirds<MATH>(h_surf);
            s = h_stat * P.inv_Hu;
            dEmax = Emax * P.inv_Hu;
            J[3][3] = -P.inv_alpha3;
            J[4][4] = -P.inv_alpha4;
        } else {
            const double alfa2 = P.inv_n * pow_two_thirds<MATH>(h_surf) * P.sqrt_slope;
            w_lin = alfa2 * P.L / P.A_h * 60.0;
            s = h_stat / P.Hu;
            dEmax = Emax / P.Hu;
            J[3][3] = -(P.alpha3 >= 1.0 ? 1.0 / P.alpha3 : 0.0);
            J[4][4] = -(P.alpha4 >= 1.0 ? 1.0 / P.alpha4 : 0.0);
        }
        J[0][0] = -m;
        J[1][0] = m - a * m;
        J[1][1] = -a - (dEmax + s * e);
        J[2][0] = (1.0 - b) * a * m;
        J[2][1] = (1.0 - b) * a;
        J[2][2] = -(w_lin < 1.0 ? (5.0 / 3.0) * w_lin : 1.0);
        J[3][0] = (1.0 - c) * b * a * m;
        J[3][1] = (1.0 - c) * b * a;
        J[4][0] = c * b * a * m;
        J[4][1] = c * b * a;
    }

    // rhs and jacobian of the same state in one pass (the Radau pass's set-up evaluates both): the shared
    // sub-expressions -- snowmelt, x1, the overflow, Emax, and above all h_surf^(2/3) -- are formed once.  Same
    // expressions as rhs<MATH> and jacobian<MATH>, hence the same values.
    static constexpr bool HAS_RHS_JACOBIAN = true;
    template <int MATH>
    static __device__ __forceinline__ void rhs_jacobian(const Local& P, const Forc& F, const double (&y)[N_EQ], double (&dydt)[N_EQ],
                                                        double (&J)[N_EQ][N_EQ])
    {
#pragma unroll
        for (int i = 0; i < N_EQ; ++i)
#pragma unroll
            for (int j = 0; j < N_EQ; ++j) J[i][j] = 0.0;
        const double h_snow = y[0], h_stat = y[1], h_surf = y[2], h_grav = y[3], h_aq = y[4];
        const bool warm = F.warm_mask != 0ull;
        const double snowmelt = __longlong_as_double(__double_as_longlong(vmin(h_snow, F.melt_pot)) & (long long)F.warm_mask);
        const double m = (warm && h_snow < F.melt_pot) ? 1.0 : 0.0;
        const double x1 = F.rain + snowmelt;
        dydt[0] = F.rain - snowmelt;
        const double over = x1 + h_stat - P.Hu;
        const double x2 = vmax(0.0, over);
        const double a = over > 0.0 ? 1.0 : 0.0;
        const double d1 = x1 - x2;
        const double Emax = vmin(F.e_pot, h_stat);
        const double e = h_stat < F.e_pot ? 1.0 : 0.0;
        const double x3 = vmin(x2, P.infil);
        const double b = x2 < P.infil ? 1.0 : 0.0;
        const double c = x3 < P.perco ? 1.0 : 0.0;
        const double d2 = x2 - x3;
        const double x4 = vmin(x3, P.perco);
        const double d3 = x3 - x4;
        const double p23 = pow_two_thirds<MATH>(h_surf);
        double w_lin, s, dEmax;
        if constexpr (MATH == MATH_FAST) {
            w_lin = P.manning * p23;
            s = h_stat * P.inv_Hu;
            dEmax = Emax * P.inv_Hu;
            J[3][3] = -P.inv_alpha3;
            J[4][4] = -P.inv_alpha4;
            dydt[3] = d3 - h_grav * P.inv_alpha3;
            dydt[4] = x4 - h_aq * P.inv_alpha4;
            dydt[2] = d2 - h_surf * vmin(1.0, w_lin);
        } else {
            const double alfa2 = P.inv_n * p23 * P.sqrt_slope;
            w_lin = alfa2 * P.L / P.A_h * 60.0;   // (the Jacobian's form, as jacobian<STRICT> writes it)
            s = strict_div(h_stat, P.Hu, P.inv_Hu, P.div_ok);
            dEmax = Emax / P.Hu;
            J[3][3] = -(P.alpha3 >= 1.0 ? 1.0 / P.alpha3 : 0.0);
            J[4][4] = -(P.alpha4 >= 1.0 ? 1.0 / P.alpha4 : 0.0);
            dydt[3] = d3 - (P.alpha3 >= 1.0 ? strict_div(h_grav, P.alpha3, P.inv_alpha3, P.div_ok) : 0.0);
            dydt[4] = x4 - (P.alpha4 >= 1.0 ? strict_div(h_aq, P.alpha4, P.inv_alpha4, P.div_ok) : 0.0);
            dydt[2] = d2 - h_surf * vmin(1.0, strict_div(alfa2 * P.L, P.A_h, P.inv_A_h, P.div_ok) * 60.0);   // (the RHS's form)
        }
        dydt[1] = d1 - s * Emax;
        J[0][0] = -m;
        J[1][0] = m - a * m;
        J[1][1] = -a - (dEmax + s * e);
        J[2][0] = (1.0 - b) * a * m;
        J[2][1] = (1.0 - b) * a;
        J[2][2] = -(w_lin < 1.0 ? (5.0 / 3.0) * w_lin : 1.0);
        J[3][0] = (1.0 - c) * b * a * m;
        J[3][1] = (1.0 - c) * b * a;
        J[4][0] = c * b * a * m;
        J[4][1] = c * b * a;
    }

    template <int MATH, bool DRY>
    static __device__ __forceinline__ bool rhs_impl(const Local& P, const Forc& F, const double (&y)[N_EQ],
                                                    double (&dydt)[N_EQ])
    {
        const double h_snow = y[0], h_stat = y[1], h_surf = y[2], h_grav = y[3], h_aq = y[4];

        // snow (:87-91)
        // (T >= thr) ? fmin(h_snow, T*melt_f) : 0.0 as min + bit-mask: no exec-mask branch
        const double snowmelt =
            __longlong_as_double(__double_as_longlong(vmin(h_snow, F.melt_pot)) & (long long)F.warm_mask);
        const double x1 = F.rain + snowmelt;
        dydt[0] = F.rain - snowmelt;

        // static (:94-98)
        const double x2 = vmax(0.0, x1 + h_stat - P.Hu);
        const double d1 = x1 - x2;
        const double Emax = vmin(F.e_pot, h_stat);
        double s;
        if constexpr (MATH == MATH_FAST) s = h_stat * P.inv_Hu;
        else s = strict_div(h_stat, P.Hu, P.inv_Hu, P.div_ok);  // == h_stat / Hu, bit for bit
        dydt[1] = d1 - s * Emax;

        // surface (:101-105).  pow(0, 2/3) == 0 exactly, hence alfa2 == 0, w == 0 and
        // dydt[2] == d2 - 0: a dry surface store needs no transcendental.  Waves whose 64
        // hillslopes are all dry take the DRY variant (rk_stages' fast path); the general
        // variant evaluates the term for every lane WITHOUT a per-lane branch -- a dry lane in
        // a mixed wave gets w == 0 from the arithmetic itself (pow_two_thirds(0) == 0 in both
        // math modes), and seven branches per attempt cost more than they could ever skip.
        const double x3 = vmin(x2, P.infil);
        const double d2 = x2 - x3;
        double out2 = d2;
        if constexpr (!DRY) {
            double w;
            if constexpr (MATH == MATH_FAST) {
                w = vmin(1.0, P.manning * pow_two_thirds<MATH>(h_surf));
            } else {
                const double alfa2 = P.inv_n * pow_two_thirds<MATH>(h_surf) * P.sqrt_slope;
                w = vmin(1.0, strict_div(alfa2 * P.L, P.A_h, P.inv_A_h, P.div_ok) * 60.0);
            }
            out2 = d2 - h_surf * w;
        }
        dydt[2] = out2;

        // gravitational (:108-110) and aquifer (:113)
        const double x4 = vmin(x3, P.perco);
        const double d3 = x3 - x4;
        if constexpr (MATH == MATH_FAST) {
            dydt[3] = d3 - h_grav * P.inv_alpha3;
            dydt[4] = x4 - h_aq * P.inv_alpha4;
        } else {
            dydt[3] = d3 - (P.alpha3 >= 1.0 ? strict_div(h_grav, P.alpha3, P.inv_alpha3, P.div_ok) : 0.0);
            dydt[4] = x4 - (P.alpha4 >= 1.0 ? strict_div(h_aq, P.alpha4, P.inv_alpha4, P.div_ok) : 0.0);
        }
        return DRY && d2 != 0.0;
    }
};

// ------------------------------------------------------------------------------------
// DummyModel: the 5-equation linear test system of the reference's README; its source
// is absent from the tree, the RHS is the notebook's (model_dummy_python.ipynb:65-89,
// the code cell: I2 = 0.6*H1).  No parameters, no forcing.
// ------------------------------------------------------------------------------------
struct DummyDev {
    static constexpr int UID = HLM_UID_DUMMY;
    static constexpr int N_EQ = 5;
    static constexpr int NF = 0;
    struct Local {};
    struct Forc {};
    static constexpr bool HAS_FASTPATH = false;
    static constexpr bool HAS_NORAINPATH = false;
    static constexpr int FAST_ZERO_COMP = 0;
    static constexpr int NORAIN_LINEAR = 0;
    static constexpr int NORAIN_LINEAR_COMP[1] = {0};
    static __device__ __forceinline__ double norain_rate(const Local&, int) { return 0.0; }
    static __device__ __forceinline__ bool norainpath_ok(const Local&, const Forc&, const double (&)[N_EQ]) { return false; }
    template <int MATH>
    static __device__ __forceinline__ bool rhs_norain(const Local&, const Forc&, const double (&)[N_EQ],
                                                      double (&)[N_EQ]) { return true; }
    static __device__ __forceinline__ bool fastpath_ok(const double (&)[N_EQ]) { return false; }
    template <int MATH>
    static __device__ __forceinline__ bool rhs_fast(const Local&, const Forc&, const double (&)[N_EQ],
                                                    double (&)[N_EQ]) { return true; }
    static constexpr bool HAS_JACOBIAN = true;  // constant: the system is linear
    static constexpr bool HAS_RHS_JACOBIAN = false;
    static constexpr bool JACOBIAN_LOWER_TRIANGULAR = true;
    static constexpr bool jacobian_zero(int, int) { return false; }
    template <int MATH>
    static __device__ __forceinline__ void jacobian(const Local&, const Forc&, const double (&)[N_EQ],
                                                    double (&J)[N_EQ][N_EQ])
    {
        const double M[N_EQ][N_EQ] = {{-0.5, 0, 0, 0, 0}, {0.5, -0.9, 0, 0, 0}, {0, 0.3, 0, 0, 0},
                                            {0, 0.6, 0, -0.4, 0}, {0, 0, 0, 0.4, 0}};
#pragma unroll
        for (int i = 0; i < N_EQ; ++i)
#pragma unroll
            for (int j = 0; j < N_EQ; ++j) J[i][j] = M[i][j];
    }

    template <int MATH>
    static __device__ __forceinline__ Local load(const void*, int) { return Local{}; }
    static __device__ __forceinline__ Forc prep_forcing(const Local&, const double (&)[1]) { return Forc{}; }

    template <int MATH>
    static __device__ __forceinline__ void rhs(const Local&, const Forc&, const double (&y)[N_EQ],
                                               double (&dydt)[N_EQ])
    {
        // Every product-sum below is written as the ONE operation sequence it is to compile to (explicit FMAs: a
        // product that feeds an fma intrinsic cannot be fused any further -- the pragma below only documents the
        // intent, -ffp-contract=fast on the command line disregards it): the kernels inline this function several times (k0 / k1..k6, first attempt / loop body,
        // resumed / fresh lane) and all copies must round alike, or a resumed run differs from the uninterrupted
        // one in the last bit.  Left to -ffp-contract=fast, "I2 - I3" (two products, one sum) was fused as
        // fma(-0.4, H3, I2) in one copy and as fma(0.6, H1, -I3) in another.  The forms are the ones the compiler
        // had chosen for the majority of the copies.
#pragma clang fp contract(off)
        const double H0 = y[0], H1 = y[1], H3 = y[3];
        dydt[0] = __builtin_fma(-0.5, H0, 1.0);                                                          // 1.0 - Y0
        dydt[1] = __builtin_fma(-0.6, H1, __builtin_fma(-0.3, H1, __builtin_fma(0.5, H0, 1.2)) - 0.4);   // 1.2 + Y0 - X2 - 0.4 - I2
        dydt[2] = __builtin_fma(0.3, H1, -0.2);                                                          // X2 - 0.2
        dydt[3] = __builtin_fma(-0.4, H3, 0.6 * H1) - 0.3;                                               // I2 - I3 - 0.3
        dydt[4] = __builtin_fma(0.4, H3, -0.1);                                                          // I3 - 0.1
    }
};

// ------------------------------------------------------------------------------------
// ToyModel3: a three-store model that exists to prove the plug-in surface is generic (the reference's
// README "Add a New Model": a struct with UID / N_EQ / Parameters / rhs registered in model_registry):
// N_EQ = 3, one forcing, no cheap RHS variants, no analytic Jacobian -- and a Jacobian that is NOT
// triangular (store 2 feeds back into store 0), so the Radau pass takes its general route for it: finite
// differences (radau_step_dense.cuh:13-31) and pivoted LU.
//   dy0 = F0 - k1 y0 - c y0 y2        inflow, linear outflow, uptake by store 2
//   dy1 = k1 y0 - k2 y1^2             quadratic outflow
//   dy2 = k2 y1^2 - k3 y2 + c y0 y2
// k1 = 1 / alpha3, k2 = 0.02, k3 = 1 / alpha4 read from the SpatialParams record (or k1 = 0.05, k3 = 0.01
// without one, or where the record holds no positive value), c = 0.3.
// ------------------------------------------------------------------------------------
struct ToyDev {
    static constexpr int UID = HLM_UID_TOY3;
    static constexpr int N_EQ = 3;
    static constexpr int NF = 1;
    struct Local { double k1, k3; };
    struct Forc { double inflow; };
    static constexpr bool HAS_FASTPATH = false;
    static constexpr bool HAS_NORAINPATH = false;
    static constexpr bool HAS_JACOBIAN = false;
    static constexpr bool HAS_RHS_JACOBIAN = false;
    static constexpr bool JACOBIAN_LOWER_TRIANGULAR = false;
    static constexpr int FAST_ZERO_COMP = 0;
    static constexpr int NORAIN_LINEAR = 0;
    static constexpr int NORAIN_LINEAR_COMP[1] = {0};
    static __device__ __forceinline__ double norain_rate(const Local&, int) { return 0.0; }
    static __device__ __forceinline__ bool norainpath_ok(const Local&, const Forc&, const double (&)[N_EQ]) { return false; }
    template <int MATH>
    static __device__ __forceinline__ bool rhs_norain(const Local&, const Forc&, const double (&)[N_EQ], double (&)[N_EQ]) { return true; }
    static __device__ __forceinline__ bool fastpath_ok(const double (&)[N_EQ]) { return false; }
    template <int MATH>
    static __device__ __forceinline__ bool rhs_fast(const Local&, const Forc&, const double (&)[N_EQ], double (&)[N_EQ]) { return true; }
    template <int MATH>
    static __device__ __forceinline__ void jacobian(const Local&, const Forc&, const double (&)[N_EQ], double (&)[N_EQ][N_EQ]) {}

    template <int MATH>
    static __device__ __forceinline__ Local load(const void* sp_base, int sys)
    {
        Local P{0.05, 0.01};
        if (sp_base) {
            const hlm_spatial_params& S = static_cast<const hlm_spatial_params*>(sp_base)[sys];
            if (S.alpha3 > 0.0) P.k1 = 1.0 / S.alpha3;
            if (S.alpha4 > 0.0) P.k3 = 1.0 / S.alpha4;
        }
        return P;
    }
    static __device__ __forceinline__ Forc prep_forcing(const Local&, const double (&held)[NF]) { return Forc{held[0]}; }

    template <int MATH>
    static __device__ __forceinline__ void rhs(const Local& P, const Forc& F, const double (&y)[N_EQ], double (&dydt)[N_EQ])
    {
        // (one operation sequence per statement, explicit FMAs: see DummyDev::rhs.  "out0 - out1" and
        // "out1 - k3 y2" are sums of two products.)
#pragma clang fp contract(off)
        const double u = 0.3 * y[0], out1 = (0.02 * y[1]) * y[1];
        dydt[0] = __builtin_fma(-y[2], u, __builtin_fma(-P.k1, y[0], F.inflow));   // F0 - out0 - up
        dydt[1] = __builtin_fma(P.k1, y[0], -out1);                                // out0 - out1
        dydt[2] = __builtin_fma(y[2], u, __builtin_fma(-P.k3, y[2], out1));        // out1 - k3 y2 + up
    }
};

}  // namespace hlm
