// hlm/output_csv.hpp -- final.csv / dense.csv writers in the reference's two historical
// layouts:
//   Dummy era (src/final.csv:1, src/dense.csv:1): header "Var0..Var4" / "time,Var{c}_sys{s}",
//       default ostream formatting (6 significant digits);
//   204 era (main.cpp:734-773, src/final_204_a.csv:1, src/dense_204_a.csv:1-3): header
//       "h_snow,var1..var4" / "time,var{c}_sys{s}", time fixed with 8 decimals and values with
//       9 (std::fixed stays set after the first time column, as in the reference).
// dense index = (s*nq + q)*N_EQ + c (main.cpp:766).
#pragma once

#include <fstream>
#include <iomanip>
#include <stdexcept>
#include <string>
#include <vector>

namespace hlm_io {

enum class CsvStyle { Dummy, Model204 };

inline void write_final_csv(const std::string& path, const std::vector<double>& y_final, int num_systems, int n_eq,
                            CsvStyle style)
{
    std::ofstream f(path);
    if (!f.is_open()) throw std::runtime_error("Failed to open output file: " + path);
    if (style == CsvStyle::Dummy) {
        for (int i = 0; i < n_eq; ++i) f << (i ? "," : "") << "Var" << i;
    } else {
        f << "h_snow";
        for (int i = 1; i < n_eq; ++i) f << ",var" << i;
    }
    f << "\n";
    for (int s = 0; s < num_systems; ++s) {
        for (int i = 0; i < n_eq; ++i) {
            f << y_final[(size_t)s * n_eq + i];
            if (i + 1 < n_eq) f << ",";
        }
        f << "\n";
    }
}

inline void write_dense_csv(const std::string& path, const std::vector<double>& dense,
                            const std::vector<double>& query_times, int num_systems, int n_eq, CsvStyle style)
{
    std::ofstream f(path);
    if (!f.is_open()) throw std::runtime_error("Failed to open output file: " + path);
    const int nq = (int)query_times.size();
    f << "time";
    for (int s = 0; s < num_systems; ++s)
        for (int i = 0; i < n_eq; ++i) f << (style == CsvStyle::Dummy ? ",Var" : ",var") << i << "_sys" << s;
    f << "\n";
    for (int q = 0; q < nq; ++q) {
        if (style == CsvStyle::Model204) f << std::fixed << std::setprecision(8);
        f << query_times[q];
        for (int s = 0; s < num_systems; ++s)
            for (int i = 0; i < n_eq; ++i) {
                if (style == CsvStyle::Model204) f << "," << std::setprecision(9);
                else f << ",";
                f << dense[((size_t)s * nq + q) * n_eq + i];
            }
        f << "\n";
    }
}

}  // namespace hlm_io
