// hlm/stream.hpp -- Stream<Model>: one hillslope record {id, next_id, sp, y0}
// (reference src/stream.hpp:29-51; next_id is carried but no routing exists).
#pragma once

#include <array>

#include "hlm/parameters_loader.hpp"

template <class Model>
struct Stream {
    long id;
    long next_id;
    SpatialParams sp;
    std::array<double, Model::N_EQ> y0;

    Stream(const SpatialParams& sp_, const std::array<double, Model::N_EQ>& y0_)
        : id(sp_.stream), next_id(sp_.next_stream), sp(sp_), y0(y0_) {}
};
