// Dormand-Prince 5(4) coefficients as compile-time constants (immediates in the ISA).
// Values: reference src/solver/rk45_step_dense.cuh:54-83 (c, a, b, b_alt) and :193-219
// (dense-output matrix P, SciPy's).  Only the NON-ZERO entries exist here: a zero
// tableau entry contributes h*0*k = 0 to a sum, so it is dropped at compile time
// (differs from the reference only if a stage slope is Inf/NaN).
#pragma once

namespace hlm {
namespace dp {

constexpr double C1 = 1.0 / 5.0, C2 = 3.0 / 10.0, C3 = 4.0 / 5.0, C4 = 8.0 / 9.0;

constexpr double A10 = 1.0 / 5.0;
constexpr double A20 = 3.0 / 40.0, A21 = 9.0 / 40.0;
constexpr double A30 = 44.0 / 45.0, A31 = -56.0 / 15.0, A32 = 32.0 / 9.0;
constexpr double A40 = 19372.0 / 6561.0, A41 = -25360.0 / 2187.0, A42 = 64448.0 / 6561.0,
                 A43 = -212.0 / 729.0;
constexpr double A50 = 9017.0 / 3168.0, A51 = -355.0 / 33.0, A52 = 46732.0 / 5247.0,
                 A53 = 49.0 / 176.0, A54 = -5103.0 / 18656.0;
// row 6 of a == b (stage 6 is evaluated at the 5th-order solution), b[1] = b[6] = 0
constexpr double B0 = 35.0 / 384.0, B2 = 500.0 / 1113.0, B3 = 125.0 / 192.0,
                 B4 = -2187.0 / 6784.0, B5 = 11.0 / 84.0;
// embedded 4th-order weights
constexpr double BA0 = 5179.0 / 57600.0, BA2 = 7571.0 / 16695.0, BA3 = 393.0 / 640.0,
                 BA4 = -92097.0 / 339200.0, BA5 = 187.0 / 2100.0, BA6 = 1.0 / 40.0;
// error weights e_s = b_s - b_alt_s (the reference forms the difference in double too)
constexpr double E0 = B0 - BA0, E2 = B2 - BA2, E3 = B3 - BA3, E4 = B4 - BA4, E5 = B5 - BA5,
                 E6 = 0.0 - BA6;

// dense output: Q_m = sum_j P[j][m] k_j ; P[1][*] = 0, P[j>0][0] = 0, P[0][0] = 1
constexpr double P01 = -8048581381.0 / 2820520608.0, P02 = 8663915743.0 / 2820520608.0,
                 P03 = -12715105075.0 / 11282082432.0;
constexpr double P21 = 131558114200.0 / 32700410799.0, P22 = -68118460800.0 / 10900136933.0,
                 P23 = 87487479700.0 / 32700410799.0;
constexpr double P31 = -1754552775.0 / 470086768.0, P32 = 14199869525.0 / 1410260304.0,
                 P33 = -10690763975.0 / 1880347072.0;
constexpr double P41 = 127303824393.0 / 49829197408.0, P42 = -318862633887.0 / 49829197408.0,
                 P43 = 701980252875.0 / 199316789632.0;
constexpr double P51 = -282668133.0 / 205662961.0, P52 = 2019193451.0 / 616988883.0,
                 P53 = -1453857185.0 / 822651844.0;
constexpr double P61 = 40617522.0 / 29380423.0, P62 = -110615467.0 / 29380423.0,
                 P63 = 69997945.0 / 29380423.0;

// ---- the method applied to y' = lambda y, in closed form ----------------------------------------------------
// For a linear homogeneous component every stage slope is lambda y p_j(z), z = lambda h, with polynomials p_j fixed by
// the tableau; so are the 5th-order solution y R(z) (R = p_6: the stability polynomial) and the error estimate
// y E(z).  Derived from the tableau at compile time and checked against the closed expressions the kernels use:
// R = 1 + z + z^2/2 + z^3/6 + z^4/24 + z^5/120 + z^6/600,  E = -97/120000 z^5 + 13/40000 z^6 - 1/24000 z^7
// (E starts at z^5: the embedded solution is of order 4).
struct Poly8 { double c[8]; };
constexpr Poly8 poly8_zero() { return Poly8{{0, 0, 0, 0, 0, 0, 0, 0}}; }
constexpr Poly8 poly8_axpy(Poly8 acc, double a, const Poly8& p)  // acc + a p
{
    for (int k = 0; k < 8; ++k) acc.c[k] += a * p.c[k];
    return acc;
}
constexpr Poly8 poly8_z_times(const Poly8& p)
{
    Poly8 r = poly8_zero();
    for (int k = 0; k < 7; ++k) r.c[k + 1] = p.c[k];
    return r;
}
constexpr Poly8 poly8_one_plus_z_times(const Poly8& p)  // 1 + z p
{
    Poly8 r = poly8_z_times(p);
    r.c[0] = 1.0;
    return r;
}
struct LinearTest { Poly8 R, E; };
constexpr LinearTest linear_test_polynomials()
{
    Poly8 p0 = poly8_zero();
    p0.c[0] = 1.0;
    const Poly8 p1 = poly8_one_plus_z_times(poly8_axpy(poly8_zero(), A10, p0));
    const Poly8 p2 = poly8_one_plus_z_times(poly8_axpy(poly8_axpy(poly8_zero(), A20, p0), A21, p1));
    const Poly8 p3 = poly8_one_plus_z_times(poly8_axpy(poly8_axpy(poly8_axpy(poly8_zero(), A30, p0), A31, p1), A32, p2));
    const Poly8 p4 = poly8_one_plus_z_times(poly8_axpy(poly8_axpy(poly8_axpy(poly8_axpy(poly8_zero(), A40, p0), A41, p1), A42, p2), A43, p3));
    const Poly8 p5 = poly8_one_plus_z_times(
        poly8_axpy(poly8_axpy(poly8_axpy(poly8_axpy(poly8_axpy(poly8_zero(), A50, p0), A51, p1), A52, p2), A53, p3), A54, p4));
    const Poly8 p6 = poly8_one_plus_z_times(
        poly8_axpy(poly8_axpy(poly8_axpy(poly8_axpy(poly8_axpy(poly8_zero(), B0, p0), B2, p2), B3, p3), B4, p4), B5, p5));
    const Poly8 e = poly8_z_times(poly8_axpy(
        poly8_axpy(poly8_axpy(poly8_axpy(poly8_axpy(poly8_axpy(poly8_zero(), E0, p0), E2, p2), E3, p3), E4, p4), E5, p5), E6, p6));
    return LinearTest{p6, e};
}
constexpr LinearTest LINEAR = linear_test_polynomials();
constexpr double LIN_R[7] = {1.0, 1.0, 1.0 / 2.0, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 600.0};
constexpr double LIN_E5 = -97.0 / 120000.0, LIN_E6 = 13.0 / 40000.0, LIN_E7 = -1.0 / 24000.0;
constexpr bool near_c(double a, double b) { return (a - b < 0 ? b - a : a - b) < 1e-15; }
static_assert(near_c(LINEAR.R.c[0], LIN_R[0]) && near_c(LINEAR.R.c[1], LIN_R[1]) && near_c(LINEAR.R.c[2], LIN_R[2]) &&
                  near_c(LINEAR.R.c[3], LIN_R[3]) && near_c(LINEAR.R.c[4], LIN_R[4]) && near_c(LINEAR.R.c[5], LIN_R[5]) &&
                  near_c(LINEAR.R.c[6], LIN_R[6]) && LINEAR.R.c[7] == 0.0,
              "stability polynomial of the Dormand-Prince pair");
static_assert(near_c(LINEAR.E.c[0], 0) && near_c(LINEAR.E.c[1], 0) && near_c(LINEAR.E.c[2], 0) && near_c(LINEAR.E.c[3], 0) &&
                  near_c(LINEAR.E.c[4], 0) && near_c(LINEAR.E.c[5], LIN_E5) && near_c(LINEAR.E.c[6], LIN_E6) && near_c(LINEAR.E.c[7], LIN_E7),
              "error polynomial of the Dormand-Prince pair");

}  // namespace dp

// reference src/solver/event_detector.cuh:11,15
constexpr double SLOPE_JUMP_THRESH = 100.0;
constexpr double MIN_STEP_FRACTION = 1e-6;

}  // namespace hlm
