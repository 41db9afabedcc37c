// hlm_forcing_nc -- write a (time, lat, lon) float forcing variable as a NetCDF-4 / HDF5 file that
// NetCDFLoader (include/hlm/netcdf4_io.hpp; reference src/I_O/forcing_loader.cpp:76-218) reads:
//     hlm_forcing_nc OUT.nc VAR T LAT LON RAW.f32
// RAW.f32 holds T*LAT*LON little-endian floats, time-major.  Used by bench.py's driver_e2e leg and by anyone
// who wants to feed rk45_solver --forcing-nc with synthetic series (the image has no netCDF4 / h5py).
#include <hdf5.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

int main(int argc, char** argv)
{
    if (argc != 7) {
        std::fprintf(stderr, "usage: hlm_forcing_nc OUT.nc VAR T LAT LON RAW.f32\n");
        return 2;
    }
    const hsize_t T = std::strtoull(argv[3], nullptr, 10), LA = std::strtoull(argv[4], nullptr, 10), LO = std::strtoull(argv[5], nullptr, 10);
    if (T == 0 || LA == 0 || LO == 0) { std::fprintf(stderr, "hlm_forcing_nc: T, LAT, LON must be > 0\n"); return 2; }
    std::vector<float> v(T * LA * LO);
    std::ifstream in(argv[6], std::ios::binary);
    if (!in.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(float)))) {
        std::fprintf(stderr, "hlm_forcing_nc: %s does not hold %llu floats\n", argv[6], (unsigned long long)v.size());
        return 1;
    }
    const hsize_t dims[3] = {T, LA, LO};
    const hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) { std::fprintf(stderr, "hlm_forcing_nc: cannot create %s\n", argv[1]); return 1; }
    const hid_t sp = H5Screate_simple(3, dims, nullptr);
    const hid_t d = H5Dcreate2(f, argv[2], H5T_IEEE_F32LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    const bool ok = d >= 0 && H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data()) >= 0;
    if (d >= 0) H5Dclose(d);
    H5Sclose(sp);
    H5Fclose(f);
    if (!ok) { std::fprintf(stderr, "hlm_forcing_nc: write failed\n"); return 1; }
    return 0;
}
