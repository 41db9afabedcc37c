// hlm/parameters_loader.hpp -- SpatialParams + the parameter-CSV reader.
// Drop-in for reference src/I_O/parameters_loader.hpp:19-46 / .cpp:8-107: the same 136-byte
// record (field order is ABI), columns found by header name, the same required set, unit
// conversions and std::runtime_error conditions.  Header-only.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
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

// One numeric field [b, e) of the file buffer (NUL-terminated as a whole; a field ends at ',' or the line end,
// where a number stops by itself).  strtod / strtol straight on the buffer -- what the reference's std::stod /
// std::stol (parameters_loader.cpp:72-101) call underneath, so the values and the accepted spellings (leading
// blanks, '+', inf / nan, hex floats, trailing text ignored) are the same; anything unusual is handed to
// std::stod / std::stol on a copy of the field for their exceptions.  (std::from_chars for double takes a global
// lock in this libstdc++: eight threads parse no faster than one.)
inline double field_to_double(const char* b, const char* e)
{
    if (b != e) {
        char* end = nullptr;
        errno = 0;
        const double v = std::strtod(b, &end);
        if (end != b && end <= e && errno == 0) return v;
    }
    return std::stod(std::string(b, e));
}
inline long field_to_long(const char* b, const char* e)
{
    if (b != e) {
        char* end = nullptr;
        errno = 0;
        const long v = std::strtol(b, &end, 10);
        if (end != b && end <= e && errno == 0) return v;
    }
    return std::stol(std::string(b, e));
}
// read-only mapping of a whole file whose last page has room for a terminating NUL
struct FileMapping {
    const char* data = nullptr;
    size_t length = 0;
    bool map(const std::string& path, size_t file_size)
    {
        const long page = sysconf(_SC_PAGESIZE);
        if (page <= 0 || file_size == 0 || file_size % (size_t)page == 0) return false;
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        void* p = ::mmap(nullptr, file_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) return false;
        (void)::madvise(p, file_size, MADV_WILLNEED);
        data = static_cast<const char*>(p);
        length = file_size;
        return true;
    }
    ~FileMapping() { if (data) ::munmap(const_cast<char*>(data), length); }
    FileMapping() = default;
    FileMapping(const FileMapping&) = delete;
    FileMapping& operator=(const FileMapping&) = delete;
};
}  // namespace hlm_detail

// The reader.  The file is read whole and its rows are parsed by several threads, each on a run of whole lines
// (a 1 M-row file, 231 MB, took 1.1 s of the driver's 1.6 s with getline + std::stod on one thread); rows come
// back in file order.
inline std::vector<SpatialParams> loadSpatialParams(const std::string& csv_path)
{
    std::ifstream in(csv_path, std::ios::binary);
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
    constexpr int NREQ = sizeof(required) / sizeof(required[0]);
    int at[NREQ];  // column of required[k]
    for (int k = 0; k < NREQ; ++k) {
        const auto it = col.find(required[k]);
        if (it == col.end()) throw std::runtime_error("Missing column '" + std::string(required[k]) + "' in " + csv_path);
        at[k] = it->second;
    }
    const size_t ncol = headers.size();

    // The rest of the file, NUL-terminated.  Mapped where possible (no copy of a quarter of a gigabyte out of the
    // page cache; the parsing threads fault their own parts in): the bytes after the end of a mapped file up to
    // the end of its last page read as zero, which is the terminator -- unless the file ends exactly on a page
    // boundary, then (or when mapping fails) it is read into a buffer.
    size_t body_size = 0;
    const char* body_ptr = nullptr;
    std::unique_ptr<char[]> body;
    hlm_detail::FileMapping mapping;
    {
        const std::streampos here = in.tellg();
        in.seekg(0, std::ios::end);
        const std::streampos end = in.tellg();
        in.seekg(here);
        body_size = (size_t)(end - here);
        if (mapping.map(csv_path, (size_t)end)) {
            body_ptr = mapping.data + (size_t)here;
        } else {
            body.reset(new char[body_size + 1]);
            in.read(body.get(), (std::streamsize)body_size);
            body[body_size] = '\0';
            body_ptr = body.get();
        }
    }

    constexpr double c_1 = 0.001 / 60.0;  // mm/hr -> m/min
    auto parse_run = [&](const char* p, const char* const stop, std::vector<SpatialParams>& rows) {
        std::vector<const char*> fb(ncol + 1), fe(ncol + 1);
        while (p < stop) {
            const char* eol = static_cast<const char*>(std::memchr(p, '\n', (size_t)(stop - p)));
            const char* const next = eol ? eol + 1 : stop;
            if (!eol) eol = stop;
            if (eol > p && eol[-1] == '\r') --eol;
            if (eol == p) { p = next; continue; }  // blank lines are skipped (parameters_loader.cpp:62)
            size_t nf = 0;  // fields of this row, the first ncol located
            const char* a = p;
            while (true) {
                const char* c = static_cast<const char*>(std::memchr(a, ',', (size_t)(eol - a)));
                if (nf < ncol) { fb[nf] = a; fe[nf] = c ? c : eol; }
                ++nf;
                if (!c) break;
                a = c + 1;
            }
            if (eol[-1] == ',') --nf;  // (no trailing empty field, as std::getline(ss, cell, ',') has it)
            if (nf < ncol) throw std::runtime_error("Bad row with too few fields in " + csv_path);
            auto num = [&](int k) { return hlm_detail::field_to_double(fb[at[k]], fe[at[k]]); };
            SpatialParams q{};
            q.stream = hlm_detail::field_to_long(fb[at[0]], fe[at[0]]);
            q.next_stream = hlm_detail::field_to_long(fb[at[1]], fe[at[1]]);
            q.c1 = c_1;
            q.infil = num(2) * c_1;
            q.perco = num(3) * c_1;
            q.Hu = num(4);
            q.lat = num(5);
            q.sw = num(6);
            q.ss = num(7);
            q.n_mann = num(8);
            q.slope = num(9);
            q.L = num(10);
            q.A_h = num(11);
            q.melt_f = num(12);
            q.temp_thr = num(13);
            q.alpha3 = num(14) * 24.0 * 60.0;  // days -> minutes
            q.alpha4 = num(15) * 24.0 * 60.0;
            rows.push_back(q);
            p = next;
        }
    };

    const char* const b0 = body_ptr;
    const char* const b1 = b0 + body_size;
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads > 16) nthreads = 16;
    const unsigned by_size = (unsigned)(body_size / ((size_t)4 << 20)) + 1;  // at least 4 MB of text per thread
    if (nthreads > by_size) nthreads = by_size;
    if (nthreads < 1) nthreads = 1;
    std::vector<const char*> cut(nthreads + 1, b1);
    cut[0] = b0;
    for (unsigned k = 1; k < nthreads; ++k) {  // runs of whole lines
        const char* c = b0 + body_size / nthreads * k;
        if (c < cut[k - 1]) c = cut[k - 1];
        const char* nl = c < b1 ? static_cast<const char*>(std::memchr(c, '\n', (size_t)(b1 - c))) : nullptr;
        cut[k] = nl ? nl + 1 : b1;
    }
    std::vector<std::vector<SpatialParams>> part(nthreads);
    std::vector<std::exception_ptr> err(nthreads);
    auto work = [&](unsigned k) {
        try {
            part[k].reserve((size_t)(cut[k + 1] - cut[k]) / 100 + 16);
            parse_run(cut[k], cut[k + 1], part[k]);
        } catch (...) {
            err[k] = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    for (unsigned k = 1; k < nthreads; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& t : th) t.join();
    for (unsigned k = 0; k < nthreads; ++k)
        if (err[k]) std::rethrow_exception(err[k]);  // the first bad row in file order
    size_t total = 0;
    for (const auto& v : part) total += v.size();
    std::vector<SpatialParams> rows;
    rows.reserve(total);
    for (const auto& v : part) rows.insert(rows.end(), v.begin(), v.end());
    return rows;
}
