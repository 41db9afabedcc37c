// Test helper for include/hlm/netcdf4_io.hpp (driven by tests/test_netcdf4_io.py).
//   read <file> <var> <start> <count> <out.bin>   -> prints "dims T LAT LON", dumps the floats
//   write_dense <out.nc> <ns> <nq> <neq> <level>  -> deterministic values, see value()
//   write_final <out.nc> <ns> <neq> <level>
//   write_forcing <out.nc> <var> <T> <LAT> <LON> <kind>  -> float (time,lat,lon) variable written with
//        plain HDF5 calls (test input, not product code); kind 0: 1e-4*((7t+3lat+lon)%11), kind 1: 1+((t+lat+lon)%5)
// Exceptions are printed as "EXC <type>: <what>" with exit code 3.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "hlm/netcdf4_io.hpp"

static double value(int s, int q, int c) { return s * 1000.0 + q + 0.125 * c; }

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    try {
        if (mode == "read" && argc == 7) {
            NetCDFLoader L(argv[2], argv[3]);
            std::printf("dims %zu %zu %zu loaded=%d\n", L.getTimeSize(), L.getLatSize(), L.getLonSize(), (int)L.isDataLoaded());
            const size_t start = std::strtoull(argv[4], nullptr, 10), count = std::strtoull(argv[5], nullptr, 10);
            auto chunk = L.loadTimeChunk(start, count);
            std::ofstream out(argv[6], std::ios::binary);
            out.write(reinterpret_cast<const char*>(chunk.get()), sizeof(float) * count * L.getLatSize() * L.getLonSize());
            NetCDFLoader M(std::move(L));  // move keeps the handle valid
            std::printf("moved=%d old=%d v000=%.9g\n", (int)M.isDataLoaded(), (int)L.isDataLoaded(),
                        (double)M.getValueFromChunk(chunk, 0, 0, 0, count, M.getLatSize(), M.getLonSize()));
            return 0;
        }
        if (mode == "write_dense" && argc == 7) {
            const int ns = std::atoi(argv[3]), nq = std::atoi(argv[4]), neq = std::atoi(argv[5]), lvl = std::atoi(argv[6]);
            std::vector<double> d((size_t)ns * nq * neq), t(nq);
            std::vector<int> ids(ns), st(neq);
            for (int s = 0; s < ns; ++s) { ids[s] = s; for (int q = 0; q < nq; ++q) for (int c = 0; c < neq; ++c) d[((size_t)s * nq + q) * neq + c] = value(s, q, c); }
            for (int q = 0; q < nq; ++q) t[q] = 60.0 * q;
            for (int c = 0; c < neq; ++c) st[c] = c;
            write_dense_netcdf(argv[2], d.data(), t.data(), ids.data(), st.data(), nq, ns, neq, lvl);
            return 0;
        }
        if (mode == "write_dense_windows" && argc == 8) {  // the same file through DenseNetcdfWriter, W queries per window
            const int ns = std::atoi(argv[3]), nq = std::atoi(argv[4]), neq = std::atoi(argv[5]), lvl = std::atoi(argv[6]), W = std::atoi(argv[7]);
            std::vector<double> t(nq);
            std::vector<int> ids(ns), st(neq);
            for (int s = 0; s < ns; ++s) ids[s] = s;
            for (int q = 0; q < nq; ++q) t[q] = 60.0 * q;
            for (int c = 0; c < neq; ++c) st[c] = c;
            DenseNetcdfWriter w(argv[2], t.data(), ids.data(), st.data(), nq, ns, neq, lvl, W);
            if (!w.ok()) return 4;
            for (int qb = 0; qb < nq; qb += W) {
                const int qe = qb + W < nq ? qb + W : nq;
                std::vector<double> rows((size_t)ns * (qe - qb) * neq);  // [ns][W][neq], as a window sink receives them
                for (int s = 0; s < ns; ++s)
                    for (int q = qb; q < qe; ++q)
                        for (int c = 0; c < neq; ++c) rows[((size_t)s * (qe - qb) + (q - qb)) * neq + c] = value(s, q, c);
                if (!w.write_window(qb, qe, rows.data())) return 4;
            }
            if (w.write_window(nq - 1, nq + 1, nullptr)) return 5;  // a window past the end must be refused
            return 0;
        }
        if (mode == "write_final" && argc == 6) {
            const int ns = std::atoi(argv[3]), neq = std::atoi(argv[4]), lvl = std::atoi(argv[5]);
            std::vector<double> d((size_t)ns * neq);
            std::vector<int> ids(ns), st(neq);
            for (int s = 0; s < ns; ++s) { ids[s] = s; for (int c = 0; c < neq; ++c) d[(size_t)s * neq + c] = value(s, 0, c); }
            for (int c = 0; c < neq; ++c) st[c] = c;
            write_final_netcdf(argv[2], d.data(), ids.data(), st.data(), ns, neq, lvl);
            return 0;
        }
        if (mode == "write_forcing" && argc == 8) {
            const hsize_t T = std::atoi(argv[4]), LA = std::atoi(argv[5]), LO = std::atoi(argv[6]);
            const int kind = std::atoi(argv[7]);
            std::vector<float> v(T * LA * LO);
            for (hsize_t t = 0; t < T; ++t)
                for (hsize_t a = 0; a < LA; ++a)
                    for (hsize_t o = 0; o < LO; ++o)
                        v[(t * LA + a) * LO + o] = kind == 0 ? 1e-4f * (float)((7 * t + 3 * a + o) % 11) : 1.0f + (float)((t + a + o) % 5);
            const hsize_t dims[3] = {T, LA, LO};
            const hid_t f = H5Fcreate(argv[2], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), sp = H5Screate_simple(3, dims, nullptr);
            const hid_t d = H5Dcreate2(f, argv[3], H5T_IEEE_F32LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            const bool ok = d >= 0 && H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data()) >= 0;
            H5Dclose(d); H5Sclose(sp); H5Fclose(f);
            return ok ? 0 : 4;
        }
    } catch (const std::invalid_argument& e) {
        std::printf("EXC invalid_argument: %s\n", e.what());
        return 3;
    } catch (const std::out_of_range& e) {
        std::printf("EXC out_of_range: %s\n", e.what());
        return 3;
    } catch (const std::runtime_error& e) {
        std::printf("EXC runtime_error: %s\n", e.what());
        return 3;
    }
    return 2;
}
