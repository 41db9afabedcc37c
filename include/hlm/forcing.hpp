// hlm/forcing.hpp -- lookup CSV reader and host-side assembly of the forcing block.
// LookupMapper: drop-in for reference src/I_O/forcing_loader.cpp:14-64 (header skipped,
// "stream,lat_index,lon_index", unknown id -> {-1,-1}).  ForcingBlock: the flattened
// float[sum_j nT_j][ld] array + dt/nT metadata that main.cpp:527-549 builds, from in-memory per-cell
// tables or constant values.  The NetCDF (time, lat, lon) reader that fills such tables is
// hlm/netcdf4_io.hpp (on the HDF5 C library).
#pragma once

#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

class LookupMapper {
public:
    explicit LookupMapper(const std::string& filepath) : path_(filepath) {}

    bool load()
    {
        std::ifstream in(path_);
        if (!in.is_open()) return false;
        std::string line;
        std::getline(in, line);  // header
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            const size_t a = line.find(','), b = line.find(',', a + 1);
            if (a == std::string::npos || b == std::string::npos) continue;
            const long long id = std::stoll(line.substr(0, a));
            map_[id] = {std::stoi(line.substr(a + 1, b - a - 1)), std::stoi(line.substr(b + 1))};
        }
        return true;
    }
    bool hasStream(long long id) const { return map_.count(id) != 0; }
    std::pair<int, int> getLatLon(long long id) const
    {
        auto it = map_.find(id);
        return it == map_.end() ? std::make_pair(-1, -1) : it->second;
    }
    size_t size() const { return map_.size(); }

private:
    std::string path_;
    std::unordered_map<long long, std::pair<int, int>> map_;
};

// One forcing column per DISTINCT grid cell instead of one per stream (main.cpp:527-549 expands every
// forcing to [time][stream]; the sample lookup maps 41 274 streams to 88 cells).  For each stream id, in
// order: the column of its cell (lat * lon_size + lon, main.cpp:503-504), columns numbered in order of
// first appearance; cell_of_col[c] is the flat grid index column c holds.  A stream without a valid cell
// (the reference would index out of bounds with the {-1,-1} it gets back, main.cpp:503-504) is an error.
struct CellColumns {
    std::vector<int32_t> col_of_stream;
    std::vector<size_t> cell_of_col;
};
template <class IdAt>
inline CellColumns build_cell_columns(const LookupMapper& lm, size_t n_streams, IdAt id_at, size_t lat_size, size_t lon_size,
                                      const std::string& lookup_name = "the lookup table")
{
    CellColumns cc;
    cc.col_of_stream.resize(n_streams);
    std::unordered_map<size_t, int32_t> col_of_cell;
    for (size_t s = 0; s < n_streams; ++s) {
        const long long id = id_at(s);
        const auto [lat, lon] = lm.getLatLon(id);
        if (lat < 0 || lon < 0 || (size_t)lat >= lat_size || (size_t)lon >= lon_size)
            throw std::runtime_error("stream " + std::to_string(id) + " has no valid grid cell in " + lookup_name);
        const size_t cell = (size_t)lat * lon_size + (size_t)lon;
        auto it = col_of_cell.find(cell);
        if (it == col_of_cell.end()) {
            it = col_of_cell.emplace(cell, (int32_t)cc.cell_of_col.size()).first;
            cc.cell_of_col.push_back(cell);
        }
        cc.col_of_stream[s] = it->second;
    }
    return cc;
}

struct ForcingBlock {
    std::vector<float> data;       // [sum nT][ld]
    std::vector<double> dt_hours;  // per forcing
    std::vector<size_t> nT;
    long long ld = 0;

    // append one forcing given as a per-column table [nT][ld]
    void add(const std::vector<float>& table, size_t n_times, double dt_h)
    {
        if (ld <= 0 || table.size() != n_times * (size_t)ld) throw std::runtime_error("ForcingBlock::add: bad table size");
        data.insert(data.end(), table.begin(), table.end());
        dt_hours.push_back(dt_h);
        nT.push_back(n_times);
    }
    // constant forcing (e.g. the stub rainfall 0.001 / temperature 1.0 of model_204.hpp:76-77)
    void add_constant(float value, size_t n_times, double dt_h)
    {
        add(std::vector<float>(n_times * (size_t)ld, value), n_times, dt_h);
    }
};
