// STRICT-mode powers: pow(x, 2.0/3.0) (model_204.hpp:103), pow(u, 0.2) (rk45_kernel.cu:151,156) and the Radau pass's
// pow(u, 0.25) as the reference's host library evaluates them -- correctly rounded results for the DOUBLE exponents
// the source spells (0.6666666666666666 = 2/3 - 2^-53/3, 0.2000000000000000111 = 1/5 + 2^-54/5, 0.25), in ~35 VALU
// instructions instead of the ~215 of the general pow(x, y).
//
// Until round 4 STRICT called OCML's pow: measured at up to 12 ulp from the correctly rounded value on
// [1e-12, 10] (tests/test_parity_gpu.py::test_fast_math_elementary_functions) and 70 % of the instructions of a
// STRICT attempt on a wet surface store.  glibc's pow -- what the CPU build of the reference and the oracle
// call -- is correctly rounded in all but ~1e-3 of the cases; these functions are too, so STRICT now differs
// from the oracle in the last bit of a power about once in a thousand calls instead of once in three.
//
// The scheme, for x^(p/q) with a root r = x^(-1/q):
//   seed     r0 = exp2(-log2(x)/q) in f32 (v_log_f32 / v_exp_f32); as a double r0 has 24 significant bits, hence
//            r0^2 is exact
//   residual e = 1 - x r0^q with r0^q as an unevaluated sum (two-product FMAs), known to ~1e-22
//   series   x^(p/q) = x r0^(q-p) (1 - e)^(-(q-p)/q) = B (1 + s(e)),  B = x r0^(q-p) as a double-double
//   exponent the source's constant c is p/q + d with |d| = 2^-53/3 or 2^-54/5: x^c = x^(p/q) (1 + d ln x), and
//            ln x is at hand from the seed (needed to 1e-2 only)
//   result   B_hi + (B_hi (s + d ln x) + B_lo): ONE rounding, of a sum whose small term is exact to ~1e-5 ulp.
// The argument is first reduced by an exact power of 2^q (x = x1 2^(qk), x1 in [0.5, 2^(q-1))), so the f32 seed
// sees a small exponent (|e| < 2e-6) and every positive finite double -- subnormals included -- takes the same
// path; zeros, negatives, infinities and NaNs are selected afterwards (pow()'s values).
//
// The same source compiles for the host (tests/test_strict_pow_cpu.py checks it against 50-digit decimal
// arithmetic without a GPU); the device build uses the hardware's log2 / exp2.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HLM_POW_HD __host__ __device__ __forceinline__
#else
#define HLM_POW_HD static inline
#endif
// The sums below are compensated: a product's rounding error is taken with an FMA and added back later.  The library
// is built with -ffp-contract=fast, which fuses such a product into the final addition -- b + q with b = x w becomes
// fma(x, w, q) -- and counts its error twice (seen: every fourth result one ulp off; `#pragma clang fp contract(off)`
// is ignored under =fast).  The leading product therefore passes through an empty asm: the compiler no longer sees a
// multiplication feeding the addition.  (Host builds get the same barrier, through memory.)
#if defined(__HIP_DEVICE_COMPILE__)
#define HLM_POW_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define HLM_POW_OPAQUE(x) asm volatile("" : "+m"(x))   // (host builds: the CPU test also compiles this header with contraction on)
#endif

namespace hlm {

// x = m 2^ex, m in [0.5, 1) (subnormals included; garbage for 0 / Inf / NaN, which the callers select away)
HLM_POW_HD double strict_pow_frexp(double x, int* ex)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *ex = __builtin_amdgcn_frexp_exp(x);
    return __builtin_amdgcn_frexp_mant(x);
#else
    return frexp(x, ex);
#endif
}
HLM_POW_HD float strict_pow_log2f(float m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_logf(m);
#else
    return log2f(m);
#endif
}
HLM_POW_HD float strict_pow_exp2f(float a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(a);
#else
    return exp2f(a);
#endif
}
HLM_POW_HD double strict_pow_ldexp(double x, int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexp(x, n);
#else
    return ldexp(x, n);
#endif
}

// RN(x^c), c = 0.6666666666666666 (the double 2.0/3.0), for every positive finite x
HLM_POW_HD double strict_pow_two_thirds_core(double x, double* residual = nullptr)
{
    // x = x1 8^k with x1 = m 2^r in [0.5, 4): x^c = 4^k x1^(2/3) (1 - 2^-53/3 ln x), the power of four exact
    int ex;
    const double m = strict_pow_frexp(x, &ex);
    const unsigned e3 = (unsigned)(ex + 1080);                       // 1080 = 3 * 360; ex >= -1073
    const unsigned k3 = (e3 * 43691u) >> 17;                         // floor(e3 / 3) for e3 < 2^15
    const int r = (int)(e3 - 3u * k3), k = (int)k3 - 360;
    const double x1 = strict_pow_ldexp(m, r);
    const float L1 = (float)r + strict_pow_log2f((float)m);          // log2 x1 in [-1, 2)
    const double w = (double)strict_pow_exp2f(L1 * (-1.0f / 3.0f));  // ~ x1^(-1/3), 24 bits, relative error ~2e-7
    const double w2 = w * w;                                         // exact
    const double w3 = w2 * w;
    const double w3e = __builtin_fma(w2, w, -w3);                    // w^3 = w3 + w3e exactly
    double e = __builtin_fma(-x1, w3, 1.0);
    e = __builtin_fma(-x1, w3e, e);                                  // 1 - x1 w^3, |e| < 1e-6
    if (residual) *residual = e;
    // (1 - e)^(-1/3) - 1 = e/3 + 2/9 e^2 + 14/81 e^3 (+ 35/243 e^4 < 2e-25)
    double s = __builtin_fma(e, 14.0 / 81.0, 2.0 / 9.0);
    s = __builtin_fma(e, s, 1.0 / 3.0);
    const double lg = (double)(L1 + (float)(3 * k));                 // log2 x, needed to 1e-2 only
    const double t = __builtin_fma(e, s, -2.5651598643722064e-17 * lg);   // 2^-53/3 ln 2
    double b = x1 * w;
    HLM_POW_OPAQUE(b);
    const double be = __builtin_fma(x1, w, -b);
    return strict_pow_ldexp(b + __builtin_fma(b, t, be), 2 * k);
}

// RN(u^c), c = 0.2 (the double: 1/5 + 2^-54/5), for every positive finite u
HLM_POW_HD double strict_pow_fifth_core(double u, double* residual = nullptr)
{
    // u = u1 32^k with u1 = m 2^r in [0.5, 16): u^c = 2^k u1^(1/5) (1 + 2^-54/5 ln u)
    int ex;
    const double m = strict_pow_frexp(u, &ex);
    const unsigned e5 = (unsigned)(ex + 1075);                       // 1075 = 5 * 215
    const unsigned k5 = (e5 * 52429u) >> 18;                         // floor(e5 / 5) for e5 < 2^14
    const int r = (int)(e5 - 5u * k5), k = (int)k5 - 215;
    const double u1 = strict_pow_ldexp(m, r);
    const float L1 = (float)r + strict_pow_log2f((float)m);          // log2 u1 in [-1, 4)
    const double z = (double)strict_pow_exp2f(L1 * -0.2f);           // ~ u1^(-1/5), 24 bits
    const double z2 = z * z;                                         // exact
    const double z4 = z2 * z2;
    const double z4e = __builtin_fma(z2, z2, -z4);                   // z^4 = z4 + z4e exactly
    const double z5 = z4 * z;
    double z5e = __builtin_fma(z4, z, -z5);
    z5e = __builtin_fma(z4e, z, z5e);                                // z^5 = z5 + z5e (to ~1e-32)
    double e = __builtin_fma(-u1, z5, 1.0);
    e = __builtin_fma(-u1, z5e, e);                                  // 1 - u1 z^5, |e| < 2e-6
    if (residual) *residual = e;
    // u1^(1/5) = u1 z^4 (1 - e)^(-4/5);  (1 - e)^(-4/5) - 1 = 4/5 e + 18/25 e^2 + 84/125 e^3 (+ 399/625 e^4 < 1e-23)
    double s = __builtin_fma(e, 84.0 / 125.0, 18.0 / 25.0);
    s = __builtin_fma(e, s, 0.8);
    const double lg = (double)(L1 + (float)(5 * k));
    const double t = __builtin_fma(e, s, 7.69547959311662e-18 * lg);      // 2^-54/5 ln 2
    double b = u1 * z4;
    HLM_POW_OPAQUE(b);
    double be = __builtin_fma(u1, z4, -b);
    be = __builtin_fma(u1, z4e, be);
    return strict_pow_ldexp(b + __builtin_fma(b, t, be), k);
}

// RN(u^(1/4)) -- pow(u, 0.25) of the Radau pass's controller (as its CPU restatement writes it; 0.25 is exact, no
// exponent term) -- for every positive finite u
HLM_POW_HD double strict_pow_fourth_core(double u)
{
    // u = u1 16^k with u1 = m 2^r in [0.5, 8): u^(1/4) = 2^k u1^(1/4)
    int ex;
    const double m = strict_pow_frexp(u, &ex);
    const unsigned e4 = (unsigned)(ex + 1076);                       // 1076 = 4 * 269
    const int r = (int)(e4 & 3u), k = (int)(e4 >> 2) - 269;
    const double u1 = strict_pow_ldexp(m, r);
    const float L1 = (float)r + strict_pow_log2f((float)m);          // log2 u1 in [-1, 3)
    const double z = (double)strict_pow_exp2f(L1 * -0.25f);          // ~ u1^(-1/4), 24 bits
    const double z2 = z * z;                                         // exact
    const double z3 = z2 * z;
    const double z3e = __builtin_fma(z2, z, -z3);                    // z^3 = z3 + z3e exactly
    const double z4 = z2 * z2;
    const double z4e = __builtin_fma(z2, z2, -z4);                   // z^4 = z4 + z4e exactly
    double e = __builtin_fma(-u1, z4, 1.0);
    e = __builtin_fma(-u1, z4e, e);                                  // 1 - u1 z^4
    // u1^(1/4) = u1 z^3 (1 - e)^(-3/4);  (1 - e)^(-3/4) - 1 = 3/4 e + 21/32 e^2 + 77/128 e^3 (+ 1155/2048 e^4 < 1e-23)
    double s = __builtin_fma(e, 77.0 / 128.0, 21.0 / 32.0);
    s = __builtin_fma(e, s, 0.75);
    const double t = e * s;
    double b = u1 * z3;
    HLM_POW_OPAQUE(b);
    double be = __builtin_fma(u1, z3, -b);
    be = __builtin_fma(u1, z3e, be);
    return strict_pow_ldexp(b + __builtin_fma(b, t, be), k);
}

// pow(x, 2.0 / 3.0), pow(u, 0.2) and pow(u, 0.25) for every argument: +-0 -> 0, finite x < 0 and NaN -> NaN, +-Inf -> +Inf
HLM_POW_HD double strict_pow_two_thirds(double x)
{
    double r = strict_pow_two_thirds_core(x);
    if (!(x > 0.0)) r = x == 0.0 ? 0.0 : __builtin_nan("");
    if (__builtin_fabs(x) == __builtin_inf()) r = __builtin_inf();
    return r;
}
HLM_POW_HD double strict_pow_fifth(double u)
{
    double r = strict_pow_fifth_core(u);
    if (!(u > 0.0)) r = u == 0.0 ? 0.0 : __builtin_nan("");
    if (__builtin_fabs(u) == __builtin_inf()) r = __builtin_inf();
    return r;
}
HLM_POW_HD double strict_pow_fourth(double u)
{
    double r = strict_pow_fourth_core(u);
    if (!(u > 0.0)) r = u == 0.0 ? 0.0 : __builtin_nan("");
    if (__builtin_fabs(u) == __builtin_inf()) r = __builtin_inf();
    return r;
}

}  // namespace hlm
