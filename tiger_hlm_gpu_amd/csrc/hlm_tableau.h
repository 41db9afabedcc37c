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

}  // namespace dp

// reference src/solver/event_detector.cuh:11,15
constexpr double SLOPE_JUMP_THRESH = 100.0;
constexpr double MIN_STEP_FRACTION = 1e-6;

}  // namespace hlm
