// hlm/models/model_toy3.hpp -- ToyModel3, the build's "Add a New Model" example (reference README.md,
// plug-in section: a struct with UID / N_EQ / SP_TYPE / Parameters / rhs, registered once).  It is not a
// reference model; it exists to show -- and to test -- that the engine is generic over the model:
// three state variables, one forcing, and a feedback from the last store to the first, so that its
// Jacobian is not triangular.  The device half is csrc/hlm_models.h: ToyDev (same right-hand side).
//   dy0 = F0 - k1 y0 - 0.3 y0 y2      dy1 = k1 y0 - 0.02 y1^2      dy2 = 0.02 y1^2 - k3 y2 + 0.3 y0 y2
// k1 = 1 / alpha3, k3 = 1 / alpha4 from the SpatialParams record (0.05 and 0.01 without one).
//
// Adding a model to this build, step by step (what the three existing ones consist of):
//   1. a host struct like this one under include/hlm/models/ (UID, N_EQ, Parameters, rhs);
//   2. a device struct in tiger_hlm_gpu_amd/csrc/hlm_models.h (N_EQ, NF, Local, Forc, load, prep_forcing,
//      rhs; optionally cheap RHS variants and jacobian) and its line in with_model() in hlm_abi.hip;
//   3. tiger_hlm_gpu_amd/models.py for the Python mirror (and a CPU restatement in the test infrastructure if it is to be checked).
#pragma once

#include "hlm/parameters_loader.hpp"

struct ToyModel3 {
    using SP_TYPE = SpatialParams;
    static constexpr unsigned short UID = HLM_UID_TOY3;
    static constexpr int N_EQ = 3;

    struct Parameters {
        double initialStep = 0.01;
        double rtol = 1e-6;
        double atol = 1e-9;
        double safety = 0.9;
        double minScale = 0.2;
        double maxScale = 10.0;
    };

    static void rhs(double /*t*/, const double* y, double* dydt, int /*n*/, int sys, const SP_TYPE* sp, const float* F, int nForc)
    {
        const double k1 = sp && sp[sys].alpha3 > 0.0 ? 1.0 / sp[sys].alpha3 : 0.05, k3 = sp && sp[sys].alpha4 > 0.0 ? 1.0 / sp[sys].alpha4 : 0.01;
        const double inflow = nForc > 0 ? (double)F[0] : 0.0;
        const double out0 = k1 * y[0], up = 0.3 * y[0] * y[2], out1 = 0.02 * y[1] * y[1];
        dydt[0] = inflow - out0 - up;
        dydt[1] = out0 - out1;
        dydt[2] = out1 - k3 * y[2] + up;
    }
};
