// hlm/parameters_loader.hpp -- SpatialParams + the parameter-CSV reader.
// Drop-in for reference src/I_O/parameters_loader.hpp:19-46 / .cpp:8-107: the same 136-byte
// record (field order is ABI), columns found by header name, the same required set, unit
// conversions and std::runtime_error conditions.  Header-only.
#pragma once

#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "hlm_hip.h"

// One definition shared with the C ABI: 2 x long + 15 x double, no padding.
using SpatialParams = hlm_spatial_params;
static_assert(sizeof(SpatialParams) == 136, "SpatialParams must stay 136 bytes (reference ABI)");

namespace hlm_detail {
inline std::vector<std::string> split_csv_line(const std::string& line)
{
    std::vector<std::string> out;
    std::string::size_type a = 0;
    while (true) {
        const std::string::size_type b = line.find(',', a);
        if (b == std::string::npos) { out.emplace_back(line.substr(a)); break; }
        out.emplace_back(line.substr(a, b - a));
        a = b + 1;
    }
    // std::getline(ss, cell, ',') yields no trailing empty field: mirror that
    if (!out.empty() && out.back().empty() && !line.empty() && line.back() == ',') out.pop_back();
    return out;
}
inline void strip_cr(std::string& s) { if (!s.empty() && s.back() == '\r') s.pop_back(); }
}  // namespace hlm_detail

inline std::vector<SpatialParams> loadSpatialParams(const std::string& csv_path)
{
    std::ifstream in(csv_path);
    if (!in.is_open()) throw std::runtime_error("Failed to open parameter file: " + csv_path);
    std::string line;
    if (!std::getline(in, line)) throw std::runtime_error("Empty parameter file: " + csv_path);
    hlm_detail::strip_cr(line);
    const std::vector<std::string> headers = hlm_detail::split_csv_line(line);
    std::unordered_map<std::string, int> col;
    for (int i = 0; i < (int)headers.size(); ++i) col[headers[i]] = i;

    static const char* const required[] = {"stream", "next_stream", "i2", "i3", "hu", "centroid_lat",
                                           "sw", "ss", "n", "slope", "length_km", "drainage_area_km2",
                                           "melt", "t_thres", "res_ss", "res_gw"};
    for (const char* name : required)
        if (col.find(name) == col.end())
            throw std::runtime_error("Missing column '" + std::string(name) + "' in " + csv_path);

    constexpr double c_1 = 0.001 / 60.0;  // mm/hr -> m/min
    std::vector<SpatialParams> rows;
    while (std::getline(in, line)) {
        hlm_detail::strip_cr(line);
        if (line.empty()) continue;
        const std::vector<std::string> f = hlm_detail::split_csv_line(line);
        if (f.size() < headers.size()) throw std::runtime_error("Bad row with too few fields in " + csv_path);
        auto num = [&](const char* name) { return std::stod(f[col[name]]); };
        SpatialParams p{};
        p.stream = std::stol(f[col["stream"]]);
        p.next_stream = std::stol(f[col["next_stream"]]);
        p.c1 = c_1;
        p.Hu = num("hu");
        p.lat = num("centroid_lat");
        p.sw = num("sw");
        p.ss = num("ss");
        p.n_mann = num("n");
        p.slope = num("slope");
        p.L = num("length_km");
        p.A_h = num("drainage_area_km2");
        p.melt_f = num("melt");
        p.temp_thr = num("t_thres");
        p.infil = num("i2") * c_1;
        p.perco = num("i3") * c_1;
        p.alpha3 = num("res_ss") * 24.0 * 60.0;  // days -> minutes
        p.alpha4 = num("res_gw") * 24.0 * 60.0;
        rows.push_back(p);
    }
    return rows;
}
