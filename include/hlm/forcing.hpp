// hlm/forcing.hpp -- lookup CSV reader and host-side assembly of the forcing block.
// LookupMapper: drop-in for reference src/I_O/forcing_loader.cpp:14-64 (header skipped,
// "stream,lat_index,lon_index", unknown id -> {-1,-1}).  ForcingBlock: the flattened
// float[sum_j nT_j][ld] array + dt/nT metadata that main.cpp:527-549 builds; NetCDF input is
// a later row (no libnetcdf in this image), so the sources here are
// in-memory per-cell tables or constant values.
#pragma once

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
