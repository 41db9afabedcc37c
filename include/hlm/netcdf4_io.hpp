// hlm/netcdf4_io.hpp -- NetCDF-4 forcing reader and result writers on the HDF5 C library.
//
// Drop-in for the reference's NetCDF I/O:
//   class NetCDFLoader              src/I_O/forcing_loader.hpp:35-89, forcing_loader.cpp:76-218
//   write_dense_netcdf / _final_    src/I_O/output_series.hpp:18-43, output_series.cpp:18-123
// The reference links netcdf-c; this image has no libnetcdf but ships HDF5 1.10 (+ hdf5_hl),
// and a NetCDF-4 file IS an HDF5 file whose dimensions are HDF5 dimension scales.  The reader
// therefore opens the variable as an HDF5 dataset (any numeric type is converted to float,
// as nc_get_vara_float does); the writers create the same objects netcdf-c 4.9 creates
// (checked against the reference's src/dense_example.nc: datasets system/time/variable as
// DIMENSION_SCALEs with NAME, _Netcdf4Dimid, long_name/units; "outputs" with DIMENSION_LIST
// and _Netcdf4Coordinates), so netCDF-4 tools read them.
//
// Optional component: compile with -DHLM_WITH_HDF5 -I<hdf5>/include and link -lhdf5_hl -lhdf5.
#pragma once

#ifndef HLM_WITH_HDF5
#error "hlm/netcdf4_io.hpp needs the HDF5 C library: compile with -DHLM_WITH_HDF5 (see tools/Makefile)"
#endif

#include <hdf5.h>
#include <hdf5_hl.h>

#include <algorithm>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

class NetCDFLoader {
public:
    // Opens `filename` read-only and binds the 3-D variable `varName` (time, lat, lon).
    // Throws std::runtime_error like the reference (forcing_loader.cpp:80-104).
    NetCDFLoader(const std::string& filename, const std::string& varName) : fileName(filename), varName(varName)
    {
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);  // errors are reported through exceptions
        file = H5Fopen(filename.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
        if (file < 0) throw std::runtime_error("Opening file " + filename + ": cannot open as NetCDF-4/HDF5");
        dset = H5Lexists(file, varName.c_str(), H5P_DEFAULT) > 0 ? H5Dopen2(file, varName.c_str(), H5P_DEFAULT) : -1;
        if (dset < 0) {
            close();
            throw std::runtime_error("Variable " + varName + " not found in file");
        }
        const hid_t sp = H5Dget_space(dset);
        const int nd = H5Sget_simple_extent_ndims(sp);
        if (nd != 3) {
            H5Sclose(sp);
            close();
            throw std::runtime_error("Expected 3D variable (time, lat, lon), got " + std::to_string(nd) + "D");
        }
        hsize_t dims[3];
        H5Sget_simple_extent_dims(sp, dims, nullptr);
        H5Sclose(sp);
        timeSize = dims[0];
        latSize = dims[1];
        lonSize = dims[2];
    }
    ~NetCDFLoader() { close(); }
    NetCDFLoader(const NetCDFLoader&) = delete;
    NetCDFLoader& operator=(const NetCDFLoader&) = delete;
    NetCDFLoader(NetCDFLoader&& o) noexcept { *this = std::move(o); }
    NetCDFLoader& operator=(NetCDFLoader&& o) noexcept
    {
        if (this != &o) {
            close();
            file = o.file; dset = o.dset;
            timeSize = o.timeSize; latSize = o.latSize; lonSize = o.lonSize;
            fileName = std::move(o.fileName); varName = std::move(o.varName);
            o.file = o.dset = -1;
            o.timeSize = o.latSize = o.lonSize = 0;
        }
        return *this;
    }

    // [numTimeSteps][lat][lon] floats starting at time index startTime
    // (forcing_loader.cpp:164-196; same exceptions).
    std::unique_ptr<float[]> loadTimeChunk(size_t startTime, size_t numTimeSteps)
    {
        if (numTimeSteps == 0) throw std::invalid_argument("Size of time chunk must be greater than zero");
        if (startTime >= timeSize) throw std::out_of_range("Start time index out of range");
        if (startTime + numTimeSteps > timeSize) throw std::out_of_range("Requested time steps exceed available data");
        const hsize_t start[3] = {startTime, 0, 0}, count[3] = {numTimeSteps, latSize, lonSize};
        std::unique_ptr<float[]> data(new float[numTimeSteps * latSize * lonSize]);
        const hid_t fsp = H5Dget_space(dset), msp = H5Screate_simple(3, count, nullptr);
        herr_t rc = H5Sselect_hyperslab(fsp, H5S_SELECT_SET, start, nullptr, count, nullptr);
        if (rc >= 0) rc = H5Dread(dset, H5T_NATIVE_FLOAT, msp, fsp, H5P_DEFAULT, data.get());
        H5Sclose(msp);
        H5Sclose(fsp);
        if (rc < 0) throw std::runtime_error("Reading variable data: HDF5 read failed");
        return data;
    }

    float getValueFromChunk(const std::unique_ptr<float[]>& chunk, size_t relTime, size_t lat, size_t lon,
                            size_t chunkTimeSize, size_t latSz, size_t lonSz)
    {
        if (relTime >= chunkTimeSize || lat >= latSz || lon >= lonSz) throw std::out_of_range("Chunk indices out of range");
        return chunk[relTime * (latSz * lonSz) + lat * lonSz + lon];
    }

    bool isDataLoaded() const { return file >= 0 && timeSize > 0 && latSize > 0 && lonSize > 0; }
    size_t getTimeSize() const { return timeSize; }
    size_t getLatSize() const { return latSize; }
    size_t getLonSize() const { return lonSize; }
    std::string getVariableName() const { return varName; }
    std::string getFileName() const { return fileName; }

private:
    void close()
    {
        if (dset >= 0) H5Dclose(dset);
        if (file >= 0) H5Fclose(file);
        dset = file = -1;
    }
    hid_t file = -1, dset = -1;
    size_t timeSize = 0, latSize = 0, lonSize = 0;
    std::string fileName, varName;
};

namespace hlm_nc_detail {

inline bool put_text_attr(hid_t obj, const char* name, const std::string& text)
{
    const hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, text.size() ? text.size() : 1);
    H5Tset_strpad(t, H5T_STR_NULLTERM);
    const hid_t s = H5Screate(H5S_SCALAR);
    const hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    const bool ok = a >= 0 && H5Awrite(a, t, text.c_str()) >= 0;
    if (a >= 0) H5Aclose(a);
    H5Sclose(s);
    H5Tclose(t);
    return ok;
}

inline bool put_int_attr(hid_t obj, const char* name, const int* v, int n, bool scalar)
{
    const hsize_t d = (hsize_t)n;
    const hid_t s = scalar ? H5Screate(H5S_SCALAR) : H5Screate_simple(1, &d, nullptr);
    const hid_t a = H5Acreate2(obj, name, H5T_STD_I32LE, s, H5P_DEFAULT, H5P_DEFAULT);
    const bool ok = a >= 0 && H5Awrite(a, H5T_NATIVE_INT, v) >= 0;
    if (a >= 0) H5Aclose(a);
    H5Sclose(s);
    return ok;
}

// one coordinate variable = one HDF5 dimension scale (what nc_def_dim + nc_def_var of the same
// name produce): dataset, CLASS/NAME via H5DSset_scale, _Netcdf4Dimid, _Netcdf4Coordinates
inline hid_t make_coord(hid_t file, const char* name, hid_t ftype, hid_t mtype, const void* vals, hsize_t n, int dimid)
{
    const hid_t s = H5Screate_simple(1, &n, nullptr);
    const hid_t d = H5Dcreate2(file, name, ftype, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Sclose(s);
    if (d < 0) return d;
    if (H5Dwrite(d, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, vals) < 0 || H5DSset_scale(d, name) < 0 ||
        !put_int_attr(d, "_Netcdf4Dimid", &dimid, 1, true) || !put_int_attr(d, "_Netcdf4Coordinates", &dimid, 1, false)) {
        H5Dclose(d);
        return -1;
    }
    return d;
}

// the "outputs" variable: contiguous, or chunked + shuffle + deflate like
// nc_def_var_deflate(ncid, varid, 1, 1, level)
// time_chunk > 0 (3-D, compressed): chunks span that many time steps instead of the whole time axis -- a file written
// window by window (DenseNetcdfWriter) then touches every chunk once instead of recompressing it for every window
inline hid_t make_outputs(hid_t file, int rank, const hsize_t* dims, int compression_level, hsize_t time_chunk = 0)
{
    const hid_t s = H5Screate_simple(rank, dims, nullptr);
    const hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
    if (compression_level > 0) {
        hsize_t chunk[3];
        hsize_t inner = 1;
        for (int i = 1; i < rank; ++i) {
            chunk[i] = dims[i] ? dims[i] : 1;
            if (i == 1 && rank == 3 && time_chunk > 0 && time_chunk < chunk[i]) chunk[i] = time_chunk;
            inner *= chunk[i];
        }
        const hsize_t target = (4u << 20) / sizeof(double);  // ~4 MiB chunks
        chunk[0] = std::max<hsize_t>(1, std::min<hsize_t>(dims[0] ? dims[0] : 1, target / (inner ? inner : 1)));
        H5Pset_chunk(pl, rank, chunk);
        H5Pset_shuffle(pl);
        H5Pset_deflate(pl, (unsigned)std::min(compression_level, 9));
    }
    const hid_t d = H5Dcreate2(file, "outputs", H5T_IEEE_F64LE, s, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Pclose(pl);
    H5Sclose(s);
    return d;
}

#define HLM_NC_CHECK(cond)                                                                         \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            std::cerr << "NetCDF error: " #cond " failed at " << __FILE__ << ":" << __LINE__ << std::endl; \
            goto done;                                                                             \
        }                                                                                          \
    } while (0)

}  // namespace hlm_nc_detail

// outputs[system][time][variable] + coordinate variables system(int) / time(double) / variable(int)
// (output_series.cpp:18-72).  Like the reference it reports failures on stderr and returns.
inline void write_dense_netcdf(const std::string& filename, const double* h_dense, const double* time_vals,
                               const int* linkid_vals, const int* state_vals, int num_queries, int num_systems,
                               int N_EQ, int compression_level = 4)
{
    using namespace hlm_nc_detail;
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    hid_t f = -1, ds = -1, dt = -1, dv = -1, out = -1;
    const hsize_t dims[3] = {(hsize_t)num_systems, (hsize_t)num_queries, (hsize_t)N_EQ};
    const int coords[3] = {0, 1, 2};
    f = H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    HLM_NC_CHECK(f >= 0);
    ds = make_coord(f, "system", H5T_STD_I32LE, H5T_NATIVE_INT, linkid_vals, dims[0], 0);
    HLM_NC_CHECK(ds >= 0);
    dt = make_coord(f, "time", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, time_vals, dims[1], 1);
    HLM_NC_CHECK(dt >= 0);
    dv = make_coord(f, "variable", H5T_STD_I32LE, H5T_NATIVE_INT, state_vals, dims[2], 2);
    HLM_NC_CHECK(dv >= 0);
    HLM_NC_CHECK(put_text_attr(ds, "long_name", "LinkID"));
    HLM_NC_CHECK(put_text_attr(dt, "long_name", "Time"));
    HLM_NC_CHECK(put_text_attr(dt, "units", "minutes since start of simulation"));
    HLM_NC_CHECK(put_text_attr(dv, "long_name", "state variable"));
    HLM_NC_CHECK(put_text_attr(dv, "units", "various units"));
    out = make_outputs(f, 3, dims, compression_level);
    HLM_NC_CHECK(out >= 0);
    HLM_NC_CHECK(H5Dwrite(out, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, h_dense) >= 0);
    HLM_NC_CHECK(H5DSattach_scale(out, ds, 0) >= 0 && H5DSattach_scale(out, dt, 1) >= 0 && H5DSattach_scale(out, dv, 2) >= 0);
    HLM_NC_CHECK(put_int_attr(out, "_Netcdf4Coordinates", coords, 3, false));
done:
    if (out >= 0) H5Dclose(out);
    if (dv >= 0) H5Dclose(dv);
    if (dt >= 0) H5Dclose(dt);
    if (ds >= 0) H5Dclose(ds);
    if (f >= 0) H5Fclose(f);
}

// The same file written window by window: the dense output of a run whose rows do not fit in host memory
// (BASELINE config 5: 864 GB per GPU) arrives in query windows [q_begin, q_end) -- rk45_api::run_rk45_streamed /
// hlm_rk45_integrate_streamed_sink -- and each window is one hyperslab outputs[:, q_begin:q_end, :].  The file is
// what write_dense_netcdf produces for the whole array (same objects, attributes and layout; contiguous unless
// compression is asked for).  Failures are reported on stderr like the reference's writer and make ok() false.
class DenseNetcdfWriter {
public:
    // window_queries: the window length the rows will arrive in (chunk shape of a compressed file; 0 = whole time axis)
    DenseNetcdfWriter(const std::string& filename, const double* time_vals, const int* linkid_vals, const int* state_vals,
                      int num_queries, int num_systems, int N_EQ, int compression_level = 0, int window_queries = 0)
        : ns_((hsize_t)num_systems), nq_((hsize_t)num_queries), k_((hsize_t)N_EQ)
    {
        using namespace hlm_nc_detail;
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
        const hsize_t dims[3] = {ns_, nq_, k_};
        const int coords[3] = {0, 1, 2};
        hid_t ds = -1, dt = -1, dv = -1;
        f_ = H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        HLM_NC_CHECK(f_ >= 0);
        ds = make_coord(f_, "system", H5T_STD_I32LE, H5T_NATIVE_INT, linkid_vals, dims[0], 0);
        HLM_NC_CHECK(ds >= 0);
        dt = make_coord(f_, "time", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, time_vals, dims[1], 1);
        HLM_NC_CHECK(dt >= 0);
        dv = make_coord(f_, "variable", H5T_STD_I32LE, H5T_NATIVE_INT, state_vals, dims[2], 2);
        HLM_NC_CHECK(dv >= 0);
        HLM_NC_CHECK(put_text_attr(ds, "long_name", "LinkID"));
        HLM_NC_CHECK(put_text_attr(dt, "long_name", "Time"));
        HLM_NC_CHECK(put_text_attr(dt, "units", "minutes since start of simulation"));
        HLM_NC_CHECK(put_text_attr(dv, "long_name", "state variable"));
        HLM_NC_CHECK(put_text_attr(dv, "units", "various units"));
        out_ = make_outputs(f_, 3, dims, compression_level, window_queries > 0 ? (hsize_t)window_queries : 0);
        HLM_NC_CHECK(out_ >= 0);
        HLM_NC_CHECK(H5DSattach_scale(out_, ds, 0) >= 0 && H5DSattach_scale(out_, dt, 1) >= 0 && H5DSattach_scale(out_, dv, 2) >= 0);
        HLM_NC_CHECK(put_int_attr(out_, "_Netcdf4Coordinates", coords, 3, false));
        ok_ = true;
    done:
        if (dv >= 0) H5Dclose(dv);
        if (dt >= 0) H5Dclose(dt);
        if (ds >= 0) H5Dclose(ds);
    }
    DenseNetcdfWriter(const DenseNetcdfWriter&) = delete;
    DenseNetcdfWriter& operator=(const DenseNetcdfWriter&) = delete;
    ~DenseNetcdfWriter() { close(); }

    // rows: [num_systems][q_end - q_begin][N_EQ], the layout a window sink receives
    bool write_window(int q_begin, int q_end, const double* rows)
    {
        if (!ok_ || q_begin < 0 || q_end <= q_begin || (hsize_t)q_end > nq_) return ok_ = false;
        const hsize_t start[3] = {0, (hsize_t)q_begin, 0}, count[3] = {ns_, (hsize_t)(q_end - q_begin), k_};
        const hid_t fs = H5Dget_space(out_), ms = H5Screate_simple(3, count, nullptr);
        const bool good = fs >= 0 && ms >= 0 && H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, nullptr, count, nullptr) >= 0 &&
                          H5Dwrite(out_, H5T_NATIVE_DOUBLE, ms, fs, H5P_DEFAULT, rows) >= 0;
        if (ms >= 0) H5Sclose(ms);
        if (fs >= 0) H5Sclose(fs);
        if (!good) std::cerr << "NetCDF error: writing the window [" << q_begin << ", " << q_end << ") failed" << std::endl;
        return ok_ = ok_ && good;
    }
    void close()
    {
        if (out_ >= 0) H5Dclose(out_);
        if (f_ >= 0) H5Fclose(f_);
        out_ = f_ = -1;
    }
    bool ok() const { return ok_; }

private:
    hsize_t ns_, nq_, k_;
    hid_t f_ = -1, out_ = -1;
    bool ok_ = false;
};

// outputs[system][variable] (output_series.cpp:77-123)
inline void write_final_netcdf(const std::string& filename, const double* h_y_final, const int* linkid_vals,
                               const int* state_vals, int num_systems, int N_EQ, int compression_level = 4)
{
    using namespace hlm_nc_detail;
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    hid_t f = -1, ds = -1, dv = -1, out = -1;
    const hsize_t dims[2] = {(hsize_t)num_systems, (hsize_t)N_EQ};
    const int coords[2] = {0, 1};
    f = H5Fcreate(filename.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    HLM_NC_CHECK(f >= 0);
    ds = make_coord(f, "system", H5T_STD_I32LE, H5T_NATIVE_INT, linkid_vals, dims[0], 0);
    HLM_NC_CHECK(ds >= 0);
    dv = make_coord(f, "variable", H5T_STD_I32LE, H5T_NATIVE_INT, state_vals, dims[1], 1);
    HLM_NC_CHECK(dv >= 0);
    HLM_NC_CHECK(put_text_attr(ds, "long_name", "LinkID"));
    HLM_NC_CHECK(put_text_attr(dv, "long_name", "state variable"));
    HLM_NC_CHECK(put_text_attr(dv, "units", "various units"));
    out = make_outputs(f, 2, dims, compression_level);
    HLM_NC_CHECK(out >= 0);
    HLM_NC_CHECK(H5Dwrite(out, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, h_y_final) >= 0);
    HLM_NC_CHECK(H5DSattach_scale(out, ds, 0) >= 0 && H5DSattach_scale(out, dv, 1) >= 0);
    HLM_NC_CHECK(put_int_attr(out, "_Netcdf4Coordinates", coords, 2, false));
done:
    if (out >= 0) H5Dclose(out);
    if (dv >= 0) H5Dclose(dv);
    if (ds >= 0) H5Dclose(ds);
    if (f >= 0) H5Fclose(f);
}

#undef HLM_NC_CHECK
