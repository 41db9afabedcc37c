// Exercises the header-only mirror of the reference's host API (include/hlm/*.hpp) the way
// reference src/main.cpp:640-732 uses it: setModelParameters<M>, the three phases
// setup_gpu_buffers / launch_rk45_kernel / retrieve_and_free, and run_rk45<M>; plus the error
// behaviour (std::runtime_error).  Prints "OK <what>" lines; driven by tests/test_parity_gpu.py.
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include "hlm/model_registry.hpp"
#include "hlm/models/model_204.hpp"
#include "hlm/models/model_dummy.hpp"
#include "hlm/rk45_api.hpp"
#include "hlm/stream.hpp"

#define REQUIRE(c)                                                      \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("FAILED %s (line %d)\n", #c, __LINE__);         \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    using namespace rk45_api;

    // ---- DummyModel: three phases == run_rk45, golden finals of the reference's src/final.csv --------
    DummyModel::Parameters dp;  // defaults of model_204.hpp:22-30
    setModelParameters<DummyModel>(dp);
    const int ns = 130, nq = 25;
    std::vector<double> y0((size_t)ns * DummyModel::N_EQ, 1.0), q(nq);
    for (int i = 0; i < nq; ++i) q[i] = 5.0 * (i + 1) / (nq + 1);
    auto [d_y0, d_yf, d_q, d_dense, d_stiff, n_sys, n_q] = setup_gpu_buffers<DummyModel>(y0, q);
    REQUIRE(n_sys == ns && n_q == nq && d_y0 && d_yf && d_q && d_dense && d_stiff);
    launch_rk45_kernel<DummyModel>(d_y0, d_yf, d_q, d_dense, d_stiff, n_sys, n_q, 0.0, 5.0, nullptr);
    auto [f1, dn1] = retrieve_and_free<DummyModel>(d_y0, d_yf, d_q, d_dense, d_stiff, n_sys, n_q, 0.0, 5.0, nullptr);
    REQUIRE(f1.size() == (size_t)ns * 5 && dn1.size() == (size_t)ns * nq * 5);
    const double gold[5] = {1.91791, 1.90017, 2.39397, 1.71872, 3.06922};
    for (int s = 0; s < ns; ++s)
        for (int c = 0; c < 5; ++c) REQUIRE(std::fabs(f1[(size_t)s * 5 + c] - gold[c]) < 6e-6);
    RunInfo info = last_run_info();
    REQUIRE(info.stiff.size() == (size_t)ns && info.n_attempt.size() == (size_t)ns && info.n_attempt[7] == 12 && info.stiff[7] == 0);
    auto [f2, dn2] = run_rk45<DummyModel>(y0, 0.0, 5.0, q, nullptr);
    REQUIRE(f1 == f2 && dn1 == dn2);
    // dense index (s*nq + q)*N_EQ + c (rk45_api.hpp:255-267): last query is close to the final state
    REQUIRE(std::fabs(dn1[((size_t)3 * nq + (nq - 1)) * 5 + 0] - f1[3 * 5 + 0]) < 2e-2);
    std::printf("OK dummy three-phase == run_rk45, finals match src/final.csv\n");

    // ---- Model 204 from the parameter CSV: Stream<Model> records, d_sp, forcings, parameters ------------
    const std::vector<SpatialParams> sp = loadSpatialParams(argv[1]);
    REQUIRE(sp.size() == 10 && sp[0].stream == 420555774 && sp[0].alpha3 == 2880.0);
    std::vector<Stream<Model204>> streams;
    for (const auto& p : sp) streams.emplace_back(p, std::array<double, 5>{0.01, 3.0, 0.0, 5.0, 0.2});
    REQUIRE(streams[9].id == sp[9].stream && streams[9].next_id == sp[9].next_stream);
    SpatialParams* d_sp = device_upload(sp.data(), sp.size());
    std::vector<float> forc(50 * 10);
    for (int t = 0; t < 48; ++t) for (int s = 0; s < 10; ++s) forc[t * 10 + s] = 0.001f;
    for (int t = 48; t < 50; ++t) for (int s = 0; s < 10; ++s) forc[t * 10 + s] = 1.0f;
    float* d_forc = device_upload(forc.data(), forc.size());
    set_forcings(d_forc, {1.0, 24.0}, {48, 2}, 10);
    Model204::Parameters hp;
    hp.initialStep = 1e-6;  // what main.cpp:616-631 yields
    setModelParameters<Model204>(hp);
    std::vector<double> y204(50), q204;
    for (int s = 0; s < 10; ++s) for (int c = 0; c < 5; ++c) y204[s * 5 + c] = streams[s].y0[c];
    for (double t = 0.0; t <= 2880.0; t += 60.0) q204.push_back(t);  // main.cpp:653-656
    auto [yf, dn] = run_rk45<Model204>(y204, 0.0, 2880.0, q204, d_sp);
    const double g2[5] = {0.0010000003349848149, 3.450738061754886, 0.0, 1.8393972058572117, 0.19285791579294506};
    for (int s = 0; s < 10; ++s)
        for (int c = 0; c < 5; ++c) REQUIRE(std::fabs(yf[s * 5 + c] - g2[c]) <= 1e-9 + 1e-6 * std::fabs(g2[c]));
    REQUIRE(dn[0] == 0.0 && dn[4] == 0.0);  // the query at t0 is never produced
    std::printf("OK model 204 run_rk45 matches src/final_example.nc\n");

    // ---- errors are std::runtime_error, numerical trouble is data -------------------------------------------
    bool threw = false;
    try { run_rk45<Model204>(y204, 0.0, 10.0, q204, nullptr); } catch (const std::runtime_error& e) { threw = std::string(e.what()).find("d_sp") != std::string::npos; }
    REQUIRE(threw);
    threw = false;
    Model204::Parameters bad; bad.initialStep = -1.0;
    try { setModelParameters<Model204>(bad); } catch (const std::runtime_error&) { threw = true; }
    REQUIRE(threw);
    hlm_set_attempt_cap(3);
    auto [fc, dc] = run_rk45<DummyModel>(y0, 0.0, 5.0, q, nullptr);
    hlm_set_attempt_cap(0);
    REQUIRE(last_run_info().stiff[0] == HLM_SYS_CAPPED && fc[0] == 0.0);
    std::printf("OK errors throw std::runtime_error; flags are data\n");

    device_free(d_sp);
    hlm_clear_forcings();
    device_free(d_forc);
    setModelParameters<Model204>(Model204::Parameters{});
    return 0;
}
