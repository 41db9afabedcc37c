// Test program for include/hlm/parameters_loader.hpp (run by tests/test_host_logic.py, also under ASan+UBSan by
// tools/asan_check.sh): prints every loaded row as hex floats, or "EXC <kind>: <what>".
#include <cstdio>
#include <stdexcept>

#include "hlm/parameters_loader.hpp"

int main(int argc, char** argv)
{
    if (argc != 2) return 2;
    try {
        const std::vector<SpatialParams> rows = loadSpatialParams(argv[1]);
        std::printf("ROWS %zu\n", rows.size());
        for (const SpatialParams& p : rows)
            std::printf("%ld %ld %a %a %a %a %a %a %a %a %a %a %a %a %a %a %a\n", p.stream, p.next_stream, p.c1, p.infil, p.perco, p.Hu, p.lat, p.sw,
                        p.ss, p.n_mann, p.slope, p.L, p.A_h, p.alpha3, p.alpha4, p.melt_f, p.temp_thr);
        return 0;
    } catch (const std::invalid_argument& e) {
        std::printf("EXC invalid_argument: %s\n", e.what());
    } catch (const std::out_of_range& e) {
        std::printf("EXC out_of_range: %s\n", e.what());
    } catch (const std::runtime_error& e) {
        std::printf("EXC runtime_error: %s\n", e.what());
    }
    return 3;
}
