// hlm/model_registry.hpp -- rk45_api::setModelParameters<Model>().
// Drop-in for reference src/model_registry.hpp:6-15 / model_registry.cpp:18-60: the
// reference copies the struct into the __constant__ symbol devParams; here it is handed to
// the library through the C ABI (hlm_set_model_parameters) and travels to the kernel as an
// argument.  Throws std::runtime_error on failure, like the reference.
#pragma once

#include <stdexcept>
#include <string>

#include "hlm_hip.h"

namespace rk45_api {

namespace detail {
inline void check(int rc, const char* what)
{
    if (rc != HLM_SUCCESS) throw std::runtime_error(std::string(what) + " failed: " + hlm_last_error());
}
}  // namespace detail

template <typename Model>
void setModelParameters(const typename Model::Parameters& p)
{
    const double v[6] = {p.initialStep, p.rtol, p.atol, p.safety, p.minScale, p.maxScale};
    detail::check(hlm_set_model_parameters(Model::UID, v), "setModelParameters");
}

}  // namespace rk45_api
