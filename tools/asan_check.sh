#!/bin/bash
# AddressSanitizer + UBSan pass over the CPU side: the oracle's C text under the
# tests that exercise it, and the host-side C++ headers (YAML reader, lookup / forcing block, CSV
# writers, parameter loader) through their test programs.  CPU only; prints PASS/FAIL per part.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
ASAN_LIB=$(gcc -print-file-name=libasan.so)
make -C oracle asan >/dev/null || { echo "FAIL: oracle asan build"; exit 1; }
rc=0
# detect_leaks=0: the python interpreter itself is not leak-clean
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  LD_PRELOAD=$ASAN_LIB HLM_ORACLE_SO=$ROOT/oracle/libhlm_oracle_asan.so \
  python -m pytest tests/test_oracle_golden.py tests/test_radau_oracle.py -q -x -p no:cacheprovider 2>&1 | tail -3
[ ${PIPESTATUS[0]} -eq 0 ] && echo "PASS: oracle under ASan+UBSan" || { echo "FAIL: oracle under ASan+UBSan"; rc=1; }
T=$(mktemp -d)
SAN="-O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -Wextra -Iinclude"
g++ $SAN tests/cpp/test_config_yaml.cpp -o $T/yaml && $T/yaml tools/examples/run_204.yaml >/dev/null && $T/yaml dump tools/examples/run_204.yaml >/dev/null \
  && echo "PASS: config_yaml.hpp" || { echo "FAIL: config_yaml.hpp"; rc=1; }
g++ $SAN tests/cpp/test_lookup.cpp -o $T/lookup && $T/lookup tests/golden/lookup_small_example.csv 64 64 >/dev/null \
  && echo "PASS: forcing.hpp (LookupMapper, build_cell_columns)" || { echo "FAIL: forcing.hpp"; rc=1; }
cat > $T/params.cpp <<'CPP'
#include <cstdio>
#include "hlm/parameters_loader.hpp"
#include "hlm/output_csv.hpp"
int main(int, char** argv) {
    auto v = loadSpatialParams(argv[1]);
    std::vector<double> yf(v.size() * 5, 1.5), dn(v.size() * 3 * 5, 2.5), q = {0.0, 1.0, 2.0};
    hlm_io::write_final_csv(std::string(argv[2]) + "/f.csv", yf, (int)v.size(), 5, hlm_io::CsvStyle::Model204);
    hlm_io::write_dense_csv(std::string(argv[2]) + "/d.csv", dn, q, (int)v.size(), 5, hlm_io::CsvStyle::Model204);
    try { loadSpatialParams("/nonexistent.csv"); } catch (const std::exception&) { std::puts("ok"); }
    return v.size() == 10 ? 0 : 1;
}
CPP
g++ $SAN $T/params.cpp -o $T/params && $T/params tests/golden/small_test.csv $T >/dev/null \
  && echo "PASS: parameters_loader.hpp, output_csv.hpp" || { echo "FAIL: parameters_loader.hpp / output_csv.hpp"; rc=1; }
# the threaded parameter reader through its own test program (rows in order, error paths), under TSan as well
g++ $SAN tests/cpp/test_params_loader.cpp -o $T/ploader && $T/ploader tests/golden/small_test.csv >/dev/null \
  && { printf 'stream,next_stream\n1,2\n' > $T/short.csv; $T/ploader $T/short.csv | grep -q "Missing column"; } \
  && echo "PASS: parameters_loader.hpp (test program, error path)" || { echo "FAIL: parameters_loader.hpp test program"; rc=1; }
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -Iinclude tests/cpp/test_params_loader.cpp -o $T/ploader_tsan 2>/dev/null \
  && python3 - "$T" <<'PY' && $T/ploader_tsan $T/big.csv >/dev/null && echo "PASS: parameters_loader.hpp under TSan (several parser threads)" || { echo "FAIL: parameters_loader.hpp under TSan"; rc=1; }
import sys
sys.path.insert(0, ".")
from tiger_hlm_gpu_amd import synth
synth.write_params_csv(sys.argv[1] + "/big.csv", synth.model204_params(0, 60000))
PY
# the NetCDF writers incl. the windowed one (needs the HDF5 C library of the image; skipped without it)
if [ -f /opt/conda/include/hdf5_hl.h ]; then
  H5="-DHLM_WITH_HDF5 -isystem /opt/conda/include tools/hdf5_libs/libhdf5_hl.so tools/hdf5_libs/libhdf5.so -Wl,-rpath,$ROOT/tools/hdf5_libs"
  [ -f tools/hdf5_libs/libhdf5.so ] || make -C tools hdf5_libs/libhdf5.so >/dev/null
  g++ $SAN tests/cpp/test_netcdf4_io.cpp -o $T/nc $H5 && ASAN_OPTIONS=detect_leaks=0 $T/nc write_dense_windows $T/w.nc 7 500 3 0 97 \
    && ASAN_OPTIONS=detect_leaks=0 $T/nc write_dense $T/d.nc 7 500 3 4 && echo "PASS: netcdf4_io.hpp writers (whole and windowed)" || { echo "FAIL: netcdf4_io.hpp writers"; rc=1; }
fi
rm -rf $T
exit $rc
