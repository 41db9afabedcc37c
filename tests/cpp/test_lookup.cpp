// LookupMapper + build_cell_columns (include/hlm/forcing.hpp) on a lookup CSV: prints
//   <streams> <distinct cells> <lat,lon of the first stream> <lat,lon of an unknown id>
// then one line per column "col cell n_streams", then the column of every 5000th stream.
// Usage: test_lookup <lookup.csv> <lat_size> <lon_size>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "hlm/forcing.hpp"

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    LookupMapper lm(argv[1]);
    if (!lm.load()) { std::puts("LOAD_FAILED"); return 1; }
    std::vector<long long> ids;  // in file order, as the parameter table would list them
    {
        std::ifstream in(argv[1]);
        std::string line;
        std::getline(in, line);
        while (std::getline(in, line))
            if (!line.empty()) ids.push_back(std::stoll(line.substr(0, line.find(','))));
    }
    try {
        const CellColumns cc = build_cell_columns(lm, ids.size(), [&](size_t s) { return ids[s]; }, (size_t)std::atoi(argv[2]),
                                                  (size_t)std::atoi(argv[3]), argv[1]);
        const auto first = lm.getLatLon(ids[0]), none = lm.getLatLon(-12345);
        std::printf("%zu %zu %d,%d %d,%d\n", lm.size(), cc.cell_of_col.size(), first.first, first.second, none.first, none.second);
        std::vector<size_t> count(cc.cell_of_col.size(), 0);
        for (int32_t c : cc.col_of_stream) ++count[(size_t)c];
        for (size_t c = 0; c < cc.cell_of_col.size(); ++c) std::printf("col %zu %zu %zu\n", c, cc.cell_of_col[c], count[c]);
        for (size_t s = 0; s < ids.size(); s += 5000) std::printf("stream %lld %d\n", ids[s], cc.col_of_stream[s]);
    } catch (const std::exception& e) {
        std::printf("ERROR %s\n", e.what());
        return 3;
    }
    return 0;
}
