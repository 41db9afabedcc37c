// hlm/models/model_dummy.hpp -- the README's DummyModel (reference README.md:41-42,90).  Its
// source is absent from the reference tree; the right-hand side is the one in
// src/model_dummy_python.ipynb:65-89 (I2 = 0.6*H1), which the committed src/final.csv /
// src/dense.csv agree with.  No spatial parameters, no forcing.  The build assigns UID 0.
#pragma once

#include "hlm/parameters_loader.hpp"

struct DummyModel {
    using SP_TYPE = SpatialParams;
    static constexpr unsigned short UID = HLM_UID_DUMMY;
    static constexpr int N_EQ = HLM_NEQ;

    struct Parameters {
        double initialStep = 0.01;
        double rtol = 1e-6;
        double atol = 1e-9;
        double safety = 0.9;
        double minScale = 0.2;
        double maxScale = 10.0;
    };

    static void rhs(double /*t*/, const double* y, double* dydt, int /*n*/, int /*sys*/, const SP_TYPE* /*sp*/,
                    const float* /*F*/, int /*nForc*/)
    {
        const double Y0 = 0.5 * y[0], X2 = 0.3 * y[1], I2 = 0.6 * y[1], I3 = 0.4 * y[3];
        dydt[0] = 1.0 - Y0;
        dydt[1] = 1.2 + Y0 - X2 - 0.4 - I2;
        dydt[2] = X2 - 0.2;
        dydt[3] = I2 - I3 - 0.3;
        dydt[4] = I3 - 0.1;
    }
};
