// hlm/models/model_204.hpp -- host half of the Model 204 plug-in (UID / N_EQ / SP_TYPE /
// Parameters / rhs), mirroring reference src/models/model_204.hpp:15-115.  The device half
// (what the HIP kernel evaluates) is tiger_hlm_gpu_amd/csrc/hlm_models.h; this host rhs is
// what a driver uses for its initial-step guess (reference main.cpp:615-631).
#pragma once

#include <cmath>

#include "hlm/parameters_loader.hpp"

struct Model204 {
    using SP_TYPE = SpatialParams;
    static constexpr unsigned short UID = HLM_UID_204;
    static constexpr int N_EQ = HLM_NEQ;

    struct Parameters {
        double initialStep = 0.01;
        double rtol = 1e-6;
        double atol = 1e-9;
        double safety = 0.9;
        double minScale = 0.2;
        double maxScale = 10.0;
    };

    // y = {h_snow, h_static, h_surface, h_grav, h_aquifer}; F[0] = rainfall [m/min],
    // F[1] = temperature [deg C].  Same argument list as the reference's rhs.
    static void rhs(double /*t*/, const double* y, double* dydt, int /*n*/, int sys, const SP_TYPE* sp_ptr,
                    const float* F, int nForc)
    {
        const SP_TYPE& P = sp_ptr[sys];
        const double rain = nForc > 0 ? (double)F[0] : 0.0;
        const double temp = nForc > 1 ? (double)F[1] : 0.0;
        const double melt = temp >= P.temp_thr ? std::fmin(y[0], temp * P.melt_f) : 0.0;
        const double x1 = rain + melt;
        dydt[0] = rain - melt;
        const double x2 = std::fmax(0.0, x1 + y[1] - P.Hu);
        dydt[1] = (x1 - x2) - (y[1] / P.Hu) * std::fmin(0.1 * temp, y[1]);
        const double x3 = std::fmin(x2, P.infil);
        const double alfa2 = (1.0 / P.n_mann) * std::pow(y[2], 2.0 / 3.0) * std::sqrt(P.slope);
        const double w = std::fmin(1.0, alfa2 * P.L / P.A_h * 60.0);
        dydt[2] = (x2 - x3) - y[2] * w;
        const double x4 = std::fmin(x3, P.perco);
        dydt[3] = (x3 - x4) - (P.alpha3 >= 1.0 ? y[3] / P.alpha3 : 0.0);
        dydt[4] = x4 - (P.alpha4 >= 1.0 ? y[4] / P.alpha4 : 0.0);
    }
};
