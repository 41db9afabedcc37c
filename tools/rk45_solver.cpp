// rk45_solver -- host driver of the MI355X hillslope engine (drop-in for the reference's
// src/main.cpp:255-828 executable of the same name, minus MPI / NetCDF):
//   load SpatialParams CSV -> Stream<Model> records -> upload d_sp -> forcing block ->
//   setModelParameters<Model> -> run_rk45<Model> -> final.csv / dense.csv.
// Everything the reference hard-codes (paths main.cpp:272,371,495; y0 :376; t0/tf :610-611;
// tolerances :621; query grid :653-656) is a command-line option with the reference's value
// as default.  All GPU work goes through include/hlm/*.hpp -> libhlm_hip.so (C ABI).
//
//   rk45_solver --config run.yaml       the reference's data/config.yaml schema (include/hlm/config_yaml.hpp);
//                                       flags given as well override the file
//   rk45_solver                         DummyModel, 64 systems, t 0..5, nq 10000  (README.md:88-93)
//   rk45_solver --model 204 --params data/small_test.csv --tf 2880 --query-stride 60
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <mutex>
#include <string>
#include <future>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

#include "hlm/config_yaml.hpp"
#include "hlm/forcing.hpp"
#ifdef HLM_WITH_HDF5
#include "hlm/netcdf4_io.hpp"  // NetCDFLoader, write_final_netcdf, write_dense_netcdf
#endif
#include "hlm/model_registry.hpp"
#include "hlm/models/model_204.hpp"
#include "hlm/models/model_dummy.hpp"
#include "hlm/models/model_toy3.hpp"
#include "hlm/output_csv.hpp"
#include "hlm/rk45_api.hpp"
#include "hlm/stream.hpp"

namespace {

struct Options {
    std::string model = "dummy";
    std::string params;
    int num_systems = 64;
    double t0 = 0.0, tf = -1.0;
    int nq = -1;
    double query_stride = -1.0;
    double rain = 0.001, temp = 1.0;  // the reference's stub forcings, model_204.hpp:76-77
    double initial_step = -1.0, rtol = 1e-6, atol = 1e-9;
    double safety = -1.0, min_scale = -1.0, max_scale = -1.0;  // < 0: the model's Parameters defaults
    std::string config;           // YAML run configuration (schema of the reference's data/config.yaml)
    std::string init_file;        // "hot" start: CSV, one row of N_EQ states per hillslope (parameter-file order)
    std::vector<int> states;      // subset of states written to the NetCDF outputs (empty = all); with --out-format nc
                                  // the selection is made ON THE DEVICE (dense rows of K components: hlm_set_output_states)
    bool scipy_h0 = false;
    std::string math;  // "fast" | "strict"; empty = the library default (FAST where its atol range allows)
    std::string out_final = "final.csv", out_dense = "dense.csv";
    // NetCDF forcings "file:var:dt_hours" in forcing order (main.cpp:513-514: pr 1 h, t2m 24 h)
    std::vector<std::string> forcing_nc;
    std::string lookup;           // stream,lat_index,lon_index CSV (main.cpp:495)
    bool expand_forcings = false; // reference per-stream layout instead of the per-cell table
    double stream_forcings = 0.0; // > 0: keep the series on the host, stream them in chunks of this many minutes
    std::string out_format = "csv";  // csv | nc | both  (nc: main.cpp:786-823)
    int stream_dense = 0;         // > 0 (with --out-format nc, one GPU): the dense rows are never held whole -- windows of this
                                  // many queries go from the device through pinned staging straight into the NetCDF file
    int nc_compression = 0;
    int device = 0;
    int gpus = 1;               // shard the hillslopes over this many GPUs (one host thread each)
    bool radau = false;         // re-integrate flagged systems with the Radau-IIA fallback
    bool radau_continue = false; // ... continuing from the state RK45 reached instead of restarting at t0
    int radau_pieces = 0;       // time pieces of the Radau pass's work queue (0 = automatic; hlm_set_radau_pieces)
    int time_pieces = -1;        // launch shape of runs without dense output (-1 automatic, 0 one wave per tile)
    int run_pieces = 0, group_tiles = 0;   // what its queue hands out / tiles packed per item (hlm_set_queue_shape; 0 0 automatic)
    int hybrid = 0;              // hlm_set_queue_hybrid: 1 = waves holding saturated hillslopes run one block of general stages
    bool oversubscribe = false; // allow --gpus > visible devices (shards share GPUs round-robin)
    std::string gather = "threads"; // threads: one host thread per shard writes its rows (dense output too); rccl | host: ONE
                                    // host thread drives all devices (hlm_run_rk45_sharded), final states gathered to the first
                                    // device over RCCL / copied per device -- runs without dense output (--nq 0)
    bool quiet = false;
};

[[noreturn]] void usage(const char* msg)
{
    if (msg) std::fprintf(stderr, "rk45_solver: %s\n", msg);
    std::fprintf(stderr,
                 "usage: rk45_solver [--config run.yaml] [--init-file states.csv] [--model dummy|204|toy3] [--params file.csv] [--num-systems N] [--t0 T] [--tf T]\n"
                 "                   [--nq N | --query-stride MIN] [--rain M_PER_MIN] [--temp DEG_C]\n"
                 "                   [--initial-step H | --scipy-h0] [--rtol R] [--atol A] [--math strict|fast] [--radau | --radau-continue] [--radau-pieces N] [--time-pieces P] [--run-pieces N] [--group-tiles G] [--hybrid]\n"
                 "                   [--forcing-nc FILE:VAR:DT_HOURS]... [--lookup CSV] [--expand-forcings] [--stream-forcings MIN]\n"
                 "                   [--out-final F] [--out-dense F] [--out-format csv|nc|both] [--nc-compression L] [--states I,J,..] [--stream-dense QUERIES_PER_WINDOW]\n"
                 "                   [--device D] [--gpus N [--oversubscribe] [--gather threads|rccl|host]] [--quiet]\n");
    std::exit(msg ? 2 : 0);
}

// data/config.yaml schema -> Options (what the reference hard-codes in main.cpp)
void apply_config(Options& o, const std::string& path)
{
    const size_t slash = path.rfind('/');
    const hlm_config::RunConfig c = hlm_config::from_yaml(hlm_config::load_yaml_file(path), slash == std::string::npos ? std::string() : path.substr(0, slash));
    if (c.model_uid == 0) o.model = "dummy";
    else if (c.model_uid == 204) o.model = "204";
    else if (c.model_uid == 3) o.model = "toy3";
    else throw std::runtime_error("config: model.uid " + std::to_string(c.model_uid) + " is not registered (0 = DummyModel, 3 = ToyModel3, 204)");
    if (c.has_time) { o.t0 = 0.0; o.tf = c.tf_minutes; }  // t counts minutes from time.start (main.cpp:610-611)
    if (!c.params_file.empty()) o.params = c.params_file;
    if (c.initial_mode == "hot") o.init_file = c.initial_file;
    if (c.print_interval_min > 0) o.query_stride = c.print_interval_min;
    o.states = c.states;
    o.rtol = c.rtol; o.atol = c.atol; o.safety = c.safety; o.min_scale = c.min_scale; o.max_scale = c.max_scale;
    if (c.auto_initial_step) o.scipy_h0 = true;  // "null -> auto-scaled": the driver's SciPy-style guess (main.cpp:615-631)
    else o.initial_step = c.initial_step;
    if (c.forcing_type == "folder_nc") {
        // One file per variable, <path>/<variable>.nc.  A variable is placed by its ROLE, not by its
        // position in the file: the model reads F[0] as rainfall and F[1] as temperature
        // (model_204.hpp:82-83), sampled every 1 h and 24 h (main.cpp:513-514).  A model that reads
        // forcings needs every role; nothing is shifted into a missing role's slot.
        static const struct { const char* role; double dt_h; } kSlots[] = {{"precipitation", 1.0}, {"temperature", 24.0}};
        constexpr size_t nslots = sizeof(kSlots) / sizeof(kSlots[0]);
        std::string spec[nslots];
        for (const auto& rv : c.forcing_vars) {
            size_t k = 0;
            while (k < nslots && rv.first != kSlots[k].role) ++k;
            if (k == nslots) throw std::runtime_error("config: forcings.vars." + rv.first + ": unknown role (precipitation, temperature)");
            if (!spec[k].empty()) throw std::runtime_error("config: forcings.vars." + rv.first + " is given twice");
            spec[k] = c.forcing_path + "/" + rv.second + ".nc:" + rv.second + ":" + std::to_string(kSlots[k].dt_h);
        }
        for (size_t k = 0; k < nslots; ++k) {
            if (spec[k].empty())
                throw std::runtime_error(std::string("config: forcings.vars.") + kSlots[k].role + " is missing (Model 204 reads F[" +
                                         std::to_string(k) + "] = " + kSlots[k].role + ")");
            o.forcing_nc.push_back(spec[k]);
        }
        o.lookup = c.forcing_lookup;
    }
}

Options parse(int argc, char** argv)
{
    Options o;
    for (int i = 1; i + 1 < argc; ++i)
        if (std::string(argv[i]) == "--config") { o.config = argv[i + 1]; apply_config(o, o.config); }
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) usage(("missing value for " + a).c_str());
            return argv[++i];
        };
        if (a == "--config") (void)val();
        else if (a == "--init-file") o.init_file = val();
        else if (a == "--model") o.model = val();
        else if (a == "--params") o.params = val();
        else if (a == "--num-systems") o.num_systems = std::atoi(val());
        else if (a == "--t0") o.t0 = std::atof(val());
        else if (a == "--tf") o.tf = std::atof(val());
        else if (a == "--nq") o.nq = std::atoi(val());
        else if (a == "--query-stride") o.query_stride = std::atof(val());
        else if (a == "--rain") o.rain = std::atof(val());
        else if (a == "--temp") o.temp = std::atof(val());
        else if (a == "--initial-step") o.initial_step = std::atof(val());
        else if (a == "--scipy-h0") o.scipy_h0 = true;
        else if (a == "--rtol") o.rtol = std::atof(val());
        else if (a == "--atol") o.atol = std::atof(val());
        else if (a == "--math") o.math = val();
        else if (a == "--radau") o.radau = true;
        else if (a == "--radau-continue") o.radau = o.radau_continue = true;
        else if (a == "--radau-pieces") o.radau_pieces = std::atoi(val());
        else if (a == "--time-pieces") o.time_pieces = std::atoi(val());
        else if (a == "--run-pieces") o.run_pieces = std::atoi(val());
        else if (a == "--group-tiles") o.group_tiles = std::atoi(val());
        else if (a == "--hybrid") o.hybrid = 1;
        else if (a == "--out-final") o.out_final = val();
        else if (a == "--out-dense") o.out_dense = val();
        else if (a == "--forcing-nc") o.forcing_nc.push_back(val());
        else if (a == "--lookup") o.lookup = val();
        else if (a == "--expand-forcings") o.expand_forcings = true;
        else if (a == "--stream-forcings") o.stream_forcings = std::atof(val());
        else if (a == "--out-format") o.out_format = val();
        else if (a == "--nc-compression") o.nc_compression = std::atoi(val());
        else if (a == "--stream-dense") o.stream_dense = std::atoi(val());
        else if (a == "--states") {  // config output.states from the command line: "1,3"
            o.states.clear();
            for (const char* p = val(); *p;) {
                char* end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) usage("--states expects a comma-separated list of state indices");
                o.states.push_back((int)v);
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "--device") o.device = std::atoi(val());
        else if (a == "--gpus") o.gpus = std::atoi(val());
        else if (a == "--oversubscribe") o.oversubscribe = true;
        else if (a == "--gather") o.gather = val();
        else if (a == "--quiet") o.quiet = true;
        else if (a == "-h" || a == "--help") usage(nullptr);
        else usage(("unknown option " + a).c_str());
    }
    return o;
}

// SciPy-style first step of the reference driver (main.cpp:615-631).  The reference
// evaluates it at y = 0 for system 0, which makes d0 = 0 and the guess 1e-6 for every run;
// the same happens here when --scipy-h0 is given.
template <class Model>
double scipy_initial_step(const SpatialParams* sp, const float* F, int nForc, double rtol, double atol)
{
    constexpr int N = Model::N_EQ;
    double y[N] = {}, f[N];
    Model::rhs(0.0, y, f, N, 0, sp, F, nForc);
    double d0 = 0, d1 = 0;
    for (int i = 0; i < N; ++i) {
        const double sc = atol + rtol * std::fabs(y[i]);
        d0 += (y[i] / sc) * (y[i] / sc);
        d1 += (f[i] / sc) * (f[i] / sc);
    }
    return std::fmax(1e-6, 0.01 * std::sqrt(d0) / (std::sqrt(d1) + 1e-16));
}

template <class Model>
int run(const Options& o, hlm_io::CsvStyle style)
{
    constexpr int N_EQ = Model::N_EQ;
    const bool is204 = Model::UID == HLM_UID_204;
    const bool is_toy = Model::UID == HLM_UID_TOY3;
    // wall time of every host phase around the integration (the reference's only timer sits around its NetCDF
    // writer, main.cpp:809-823); printed in the result line as "phases"
    using clk = std::chrono::steady_clock;
    auto since = [](clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); };
    double t_params = 0, t_forcing_read = 0, t_write = 0;
    clk::time_point tp = clk::now();
    // The HIP runtime and the device context take 0.1-0.2 s to come up: let them do so while this thread parses the
    // parameter file (first device of the run; any error is reported by the shard that uses the device)
    std::thread device_warmup([&o] {
        if (hlm_set_device(o.device) == HLM_SUCCESS) {
            void* p = nullptr;
            if (hlm_device_malloc(&p, 256) == HLM_SUCCESS) (void)hlm_device_free(p);
        }
    });
    struct JoinOnExit {
        std::thread& t;
        ~JoinOnExit() { if (t.joinable()) t.join(); }
    } join_warmup{device_warmup};
    // ... and the lookup table (one row per stream, main.cpp:495-507) parses beside the parameter file
    std::unique_ptr<LookupMapper> lookup;
    std::future<bool> lookup_loaded;
    if (is204 && !o.forcing_nc.empty() && !o.lookup.empty()) {
        lookup.reset(new LookupMapper(o.lookup));
        lookup_loaded = std::async(std::launch::async, [&lookup] { return lookup->load(); });
    }

    // ---- streams ---------------------------------------------------------------------------
    std::vector<SpatialParams> spatial;
    if (!o.params.empty()) spatial = loadSpatialParams(o.params);
    else if (is204) usage("--model 204 needs --params <csv>");
    else spatial.assign((size_t)o.num_systems, SpatialParams{});
    std::array<double, N_EQ> y0_common;
    y0_common.fill(1.0);
    if constexpr (Model::UID == HLM_UID_204) y0_common = {0.01, 3.0, 0.0, 5.0, 0.2};  // main.cpp:376
    if constexpr (Model::UID == HLM_UID_TOY3) y0_common = {1.0, 0.5, 0.2};
    std::vector<Stream<Model>> streams;
    streams.reserve(spatial.size());
    for (const auto& sp : spatial) streams.emplace_back(sp, y0_common);
    const int ns = (int)streams.size();

    const double t0 = o.t0;
    const double tf = o.tf > 0 ? o.tf : (is204 ? 2 * 24.0 * 60.0 : 5.0);  // main.cpp:610-611 / README
    t_params = since(tp);
    tp = clk::now();

    // ---- forcings on the host: a table [sum nT][ncol] + one column index per stream ---------------
    // NetCDF (time, lat, lon) forcings go through the lookup table (main.cpp:495-549).  The
    // reference expands every forcing to one column per stream; this driver keeps ONE column per
    // distinct grid cell plus an int32 column index per stream (41 274 streams map to 88 cells
    // in the sample lookup) -- same values, 1/469 of the bytes.  --expand-forcings builds the
    // reference's per-stream layout instead.
    ForcingBlock fb;
    std::vector<int32_t> col_of_stream(ns, 0);
    if (is204 && !o.forcing_nc.empty()) {
#ifdef HLM_WITH_HDF5
        if (o.lookup.empty()) usage("--forcing-nc needs --lookup <csv>");
        if (!lookup_loaded.get()) throw std::runtime_error("Lookup load failed: " + o.lookup);
        LookupMapper& lm = *lookup;
        std::vector<size_t> cell_of_col;
        size_t lon_size = 0;
        const double days = std::ceil((tf - t0) / 1440.0);
        for (size_t fi = 0; fi < o.forcing_nc.size(); ++fi) {
            const std::string& spec = o.forcing_nc[fi];
            const size_t c2 = spec.rfind(':'), c1 = c2 == std::string::npos ? c2 : spec.rfind(':', c2 - 1);
            if (c1 == std::string::npos) usage("--forcing-nc expects FILE:VAR:DT_HOURS");
            const std::string path = spec.substr(0, c1), var = spec.substr(c1 + 1, c2 - c1 - 1);
            const double dt_h = std::atof(spec.substr(c2 + 1).c_str());
            if (!(dt_h > 0)) usage("--forcing-nc: DT_HOURS must be > 0");
            NetCDFLoader loader(path, var);
            if (fi == 0) {
                lon_size = loader.getLonSize();
                CellColumns cc = build_cell_columns(lm, (size_t)ns, [&](size_t s) { return (long long)streams[s].id; },
                                                    loader.getLatSize(), lon_size, o.lookup);
                col_of_stream = std::move(cc.col_of_stream);
                cell_of_col = std::move(cc.cell_of_col);
                fb.ld = (long long)cell_of_col.size();
            } else if (loader.getLonSize() != lon_size) {
                throw std::runtime_error("forcing grids differ: " + path);
            }
            size_t steps = (size_t)std::llround(days * 24.0 / dt_h);  // main.cpp:531-533
            steps = std::max<size_t>(1, std::min(loader.getTimeSize(), steps));
            const auto raw = loader.loadTimeChunk(0, steps);
            const size_t grid = loader.getLatSize() * loader.getLonSize();
            std::vector<float> table(steps * cell_of_col.size());
            for (size_t t = 0; t < steps; ++t)
                for (size_t c = 0; c < cell_of_col.size(); ++c) table[t * cell_of_col.size() + c] = raw[t * grid + cell_of_col[c]];
            fb.add(table, steps, dt_h);
        }
#else
        usage("--forcing-nc: this binary was built without HDF5 (HLM_WITH_HDF5)");
#endif
    } else if (is_toy) {
        // one constant inflow (--rain), hourly samples
        fb.ld = 1;
        const size_t n_hours = (size_t)std::ceil((tf - t0) / 60.0);
        fb.add_constant((float)o.rain, n_hours ? n_hours : 1, 1.0);
    } else if (is204) {
        // constant rain / temperature (the stub forcings of model_204.hpp:76-77) with the
        // reference's hourly and daily sampling (main.cpp:513-514): one shared column
        fb.ld = 1;
        const size_t n_hours = (size_t)std::ceil((tf - t0) / 60.0), n_days = (size_t)std::ceil((tf - t0) / 1440.0);
        fb.add_constant((float)o.rain, n_hours ? n_hours : 1, 1.0);
        fb.add_constant((float)o.temp, n_days ? n_days : 1, 24.0);
    }

    t_forcing_read = since(tp);

    // ---- solver parameters (main.cpp:615-641) ---------------------------------------------------------
    typename Model::Parameters hp;
    hp.rtol = o.rtol;
    hp.atol = o.atol;
    if (o.safety > 0) hp.safety = o.safety;
    if (o.min_scale > 0) hp.minScale = o.min_scale;
    if (o.max_scale > 0) hp.maxScale = o.max_scale;
    if (o.initial_step > 0) hp.initialStep = o.initial_step;
    else if (o.scipy_h0) {
        std::vector<float> f0;  // forcing samples of system 0 at t0
        size_t row = 0;
        for (size_t j = 0; j < fb.nT.size(); ++j) { f0.push_back(fb.data[row * (size_t)fb.ld + (size_t)col_of_stream[0]]); row += fb.nT[j]; }
        hp.initialStep = scipy_initial_step<Model>(spatial.data(), f0.data(), (int)f0.size(), o.rtol, o.atol);
    }
    rk45_api::setModelParameters<Model>(hp);
    if (!o.math.empty()) {
        if (o.math != "fast" && o.math != "strict") usage("--math must be fast or strict");
        rk45_api::detail::check(hlm_set_math_mode(o.math == "fast" ? HLM_MATH_FAST : HLM_MATH_STRICT), "hlm_set_math_mode");
    }
    rk45_api::set_radau_mode(o.radau_continue ? 2 : (o.radau ? 1 : 0));
    rk45_api::set_radau_pieces(o.radau_pieces);
    rk45_api::set_time_pieces(o.time_pieces);
    rk45_api::set_queue_shape(o.run_pieces, o.group_tiles);
    rk45_api::set_queue_hybrid(o.hybrid);

    // ---- initial state and query grid --------------------------------------------------------------------
    std::vector<double> h_y0((size_t)ns * N_EQ);
    for (int s = 0; s < ns; ++s)
        for (int i = 0; i < N_EQ; ++i) h_y0[(size_t)s * N_EQ + i] = streams[s].y0[i];
    if (!o.init_file.empty()) {
        // hot start (config initial.mode: hot).  The reference names a ".uini" file but has no reader for
        // it; here: CSV, N_EQ numbers per row, one row per hillslope in parameter-file order, an
        // optional header line.
        std::ifstream in(o.init_file);
        if (!in) throw std::runtime_error("cannot open initial-state file " + o.init_file);
        std::string line;
        int row = 0, lineno = 0;
        while (std::getline(in, line)) {
            ++lineno;
            if (line.empty() || line[0] == '#') continue;
            std::vector<double> vals;
            const char* p = line.c_str();
            bool numeric = true;
            while (*p) {
                char* end = nullptr;
                const double v = std::strtod(p, &end);
                if (end == p) { numeric = false; break; }
                vals.push_back(v);
                p = end;
                while (*p == ',' || *p == ' ' || *p == '\t' || *p == '\r') ++p;
            }
            if (!numeric && row == 0 && vals.empty()) continue;  // header
            if (!numeric || (int)vals.size() != N_EQ)
                throw std::runtime_error(o.init_file + ":" + std::to_string(lineno) + ": expected " + std::to_string(N_EQ) + " numbers");
            if (row >= ns) throw std::runtime_error(o.init_file + ": more rows than hillslopes (" + std::to_string(ns) + ")");
            for (int i = 0; i < N_EQ; ++i) h_y0[(size_t)row * N_EQ + i] = vals[i];
            ++row;
        }
        if (row != ns) throw std::runtime_error(o.init_file + ": " + std::to_string(row) + " rows for " + std::to_string(ns) + " hillslopes");
    }
    std::vector<double> h_query_times;
    if (o.query_stride > 0 || (is204 && o.nq < 0)) {
        const double stride = o.query_stride > 0 ? o.query_stride : 60.0;
        for (double t = t0; t <= tf; t += stride) h_query_times.push_back(t);  // main.cpp:653-656
    } else {
        const int nq = o.nq >= 0 ? o.nq : 10000;
        for (int q = 0; q < nq; ++q) h_query_times.push_back(t0 + (tf - t0) * (q + 1) / double(nq + 1));  // src/dense.csv grid
    }
    const size_t nq = h_query_times.size();

    // ---- shard over the GPUs of the node and integrate -------------------------------------------------------
    // The reference's rank 0 cuts the parameter rows into contiguous chunks, baseChunk + (r <=
    // remainder), one per worker rank / GPU (main.cpp:275-307).  Same rule here, one host thread per
    // GPU instead of one MPI rank; hillslopes are independent, so the shards never talk, and
    // the "gather" is each thread writing its rows of the host result arrays.
    if (device_warmup.joinable()) device_warmup.join();
    int ndev = 0;
    rk45_api::detail::check(hlm_device_count(&ndev), "hlm_device_count");
    const int G = std::max(1, o.gpus);
    if (G > ndev && !o.oversubscribe)
        throw std::runtime_error("--gpus " + std::to_string(G) + " but only " + std::to_string(ndev) + " device(s) visible");
    // config output.states: with NetCDF-only output the selection is made on the device -- the dense rows the
    // kernels store, the transposing kernel moves and PCIe carries hold the K listed components only.  The CSV
    // layouts (main.cpp:734-773) have every component, so csv / both keep whole rows and subset on the host.
    for (size_t k = 0; k < o.states.size(); ++k) {
        if (o.states[k] < 0 || o.states[k] >= N_EQ) throw std::runtime_error("output.states: " + std::to_string(o.states[k]) + " is not a state of this model");
        if (k > 0 && o.states[k] <= o.states[k - 1]) throw std::runtime_error("output.states must be ascending and distinct");
    }
    const bool device_select = !o.states.empty() && (int)o.states.size() < N_EQ && o.out_format == "nc";
    const int width = device_select ? (int)o.states.size() : N_EQ;  // doubles per dense row as it leaves the device
    rk45_api::set_output_states(device_select ? o.states : std::vector<int>{});
    // --stream-dense W: config 5's way -- the dense rows are never held whole on the host; each window of W queries is
    // written into dense.nc as one hyperslab while the next one integrates (rk45_api::run_rk45_streamed)
    const bool stream_dense = o.stream_dense > 0 && nq > 0;
#ifdef HLM_WITH_HDF5
    if (stream_dense && (o.out_format != "nc" || G != 1 || o.radau))
        usage("--stream-dense needs --out-format nc, one GPU and no --radau");
#else
    if (stream_dense) usage("--stream-dense: this binary was built without HDF5 (HLM_WITH_HDF5)");
#endif
    double t_sink = 0.0;
    std::vector<double> h_y_final((size_t)ns * N_EQ), h_dense(stream_dense ? 0 : (size_t)ns * nq * width);
    std::vector<double> t_h2d(G, 0.0), t_integrate(G, 0.0), t_d2h(G, 0.0);
    std::vector<int32_t> stiff(ns, 0);
    std::vector<uint32_t> n_attempt(ns, 0);
    std::vector<std::string> errors(G);
    // The host API keeps ONE forcing registration per device, like the reference's global symbols
    // (forcing_data.cu:5-10): from set_forcings to clear_forcings a shard owns its device.  With
    // --oversubscribe several shards share a device and take turns.
    std::vector<std::mutex> device_turn((size_t)std::max(ndev, 1));
    auto run_shard = [&](int r) {
        try {
            const int base = ns / G, rem = ns % G;
            const int start = r * base + std::min(r, rem), count = base + (r < rem ? 1 : 0);
            if (count == 0) return;
            const int dev = G > 1 ? (o.device + r) % std::max(ndev, 1) : o.device;
            std::lock_guard<std::mutex> turn(device_turn[(size_t)std::min(std::max(dev, 0), std::max(ndev, 1) - 1)]);
            rk45_api::detail::check(hlm_set_device(dev), "hlm_set_device");
            clk::time_point ts = clk::now();
            SpatialParams* d_sp = rk45_api::device_upload(spatial.data() + start, (size_t)count);
            float* d_forc = nullptr;
            int32_t* d_col = nullptr;
            if (!fb.nT.empty()) {
                if (o.expand_forcings) {  // the reference's [forcing][time][stream] block for this shard
                    size_t rows = 0;
                    for (size_t n : fb.nT) rows += n;
                    std::vector<float> block(rows * (size_t)count);
                    for (size_t t = 0; t < rows; ++t)
                        for (int s = 0; s < count; ++s) block[t * count + s] = fb.data[t * (size_t)fb.ld + (size_t)col_of_stream[start + s]];
                    d_forc = rk45_api::device_upload(block.data(), block.size());
                    rk45_api::set_forcings(d_forc, fb.dt_hours, fb.nT, count);
                } else if (o.stream_forcings > 0) {
                    // series stay in host memory; the library uploads the rows each time chunk needs
                    d_col = rk45_api::device_upload(col_of_stream.data() + start, (size_t)count);
                    std::vector<const float*> tables;
                    size_t row = 0;
                    for (size_t n : fb.nT) { tables.push_back(fb.data.data() + row * (size_t)fb.ld); row += n; }
                    rk45_api::set_forcings_host(tables, fb.dt_hours, fb.nT, fb.ld, o.stream_forcings, d_col);
                } else {
                    d_forc = rk45_api::device_upload(fb.data.data(), fb.data.size());
                    d_col = rk45_api::device_upload(col_of_stream.data() + start, (size_t)count);
                    rk45_api::set_forcings(d_forc, fb.dt_hours, fb.nT, fb.ld, d_col);
                }
            } else {
                rk45_api::detail::check(hlm_clear_forcings(), "hlm_clear_forcings");
            }
            const std::vector<double> y0_shard(h_y0.begin() + (size_t)start * N_EQ, h_y0.begin() + (size_t)(start + count) * N_EQ);
#ifdef HLM_WITH_HDF5
            if (stream_dense) {
                std::vector<int> linkid_vals(ns), state_vals(width);
                for (int s = 0; s < ns; ++s) linkid_vals[s] = s;
                for (int v = 0; v < width; ++v) state_vals[v] = device_select ? o.states[v] : v;
                auto nc_name = [](std::string p) { const size_t d = p.rfind('.'); return (d == std::string::npos ? p : p.substr(0, d)) + ".nc"; };
                DenseNetcdfWriter writer(nc_name(o.out_dense), h_query_times.data(), linkid_vals.data(), state_vals.data(), (int)nq, ns, width,
                                         o.nc_compression, o.stream_dense);
                if (!writer.ok()) throw std::runtime_error("cannot create " + nc_name(o.out_dense));
                t_h2d[r] = since(ts);
                ts = clk::now();
                const auto yf = rk45_api::run_rk45_streamed<Model>(y0_shard, t0, tf, h_query_times, d_sp, o.stream_dense,
                                                                   [&](int qb, int qe, const double* rows) {
                                                                       const clk::time_point tw = clk::now();
                                                                       if (!writer.write_window(qb, qe, rows)) throw std::runtime_error("dense.nc: window write failed");
                                                                       t_sink += since(tw);
                                                                   });
                t_integrate[r] = since(ts);  // (integration, D2H and the file writes of the windows overlap: one phase)
                writer.close();
                const rk45_api::RunInfo info = rk45_api::last_run_info();
                std::copy(yf.begin(), yf.end(), h_y_final.begin() + (size_t)start * N_EQ);
                std::copy(info.stiff.begin(), info.stiff.end(), stiff.begin() + start);
                if (!info.n_attempt.empty()) std::copy(info.n_attempt.begin(), info.n_attempt.end(), n_attempt.begin() + start);
                rk45_api::device_free(d_sp);
                hlm_clear_forcings();
                if (d_forc) rk45_api::device_free(d_forc);
                if (d_col) rk45_api::device_free(d_col);
                return;
            }
#endif
            // run_rk45<Model> (rk45_api.hpp:273-313) spelt out in its three phases so that each can be timed
            auto [d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, ns_b, nq_b] =
                rk45_api::setup_gpu_buffers<Model>(y0_shard, h_query_times);
            t_h2d[r] = since(ts);
            ts = clk::now();
            try {
                rk45_api::launch_rk45_kernel<Model>(d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, ns_b, nq_b, t0, tf, d_sp);
            } catch (...) {
                hlm_gpu_buffers b = rk45_api::detail::adopt(d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, ns_b, nq_b, true);
                hlm_retrieve_and_free(&b, nullptr, nullptr, nullptr, nullptr, nullptr);
                throw;
            }
            t_integrate[r] = since(ts);
            ts = clk::now();
            auto [yf, dn] = rk45_api::retrieve_and_free<Model>(d_y0_all, d_y_final_all, d_query_times, d_dense_all, d_stiff, ns_b, nq_b, t0, tf, d_sp);
            t_d2h[r] = since(ts);
            const rk45_api::RunInfo info = rk45_api::last_run_info();
            std::copy(yf.begin(), yf.end(), h_y_final.begin() + (size_t)start * N_EQ);
            std::copy(dn.begin(), dn.end(), h_dense.begin() + (size_t)start * nq * width);
            std::copy(info.stiff.begin(), info.stiff.end(), stiff.begin() + start);
            if (!info.n_attempt.empty()) std::copy(info.n_attempt.begin(), info.n_attempt.end(), n_attempt.begin() + start);
            rk45_api::device_free(d_sp);
            hlm_clear_forcings();
            if (d_forc) rk45_api::device_free(d_forc);
            if (d_col) rk45_api::device_free(d_col);
        } catch (const std::exception& e) {
            errors[r] = e.what();
        }
    };
    const auto w0 = std::chrono::steady_clock::now();
    std::string gather_note = "\"threads\"";
    if (o.gather == "rccl" || o.gather == "host") {
        // One host thread, all devices: the reference's rank-0 split (main.cpp:275-307) inside the library and the
        // end-of-run gather over RCCL (hlm_run_rk45_sharded) -- final states, flags and step counters only.
        // (dense output: --gather host -- every device streams the rows of its shards to the host through its own pinned
        // windows, hlm_run_rk45_sharded_streamed; nothing is gathered between the GPUs)
        if ((nq > 0 && o.gather == "rccl") || o.stream_forcings > 0 || stream_dense)
            usage("--gather rccl gathers final states only: needs --nq 0 (dense output: --gather host) and neither --stream-forcings nor --stream-dense");
        const int nd = std::min(G, std::max(ndev, 1));
        std::vector<int> devices((size_t)nd);
        for (int i = 0; i < nd; ++i) devices[(size_t)i] = (o.device + i) % std::max(ndev, 1);
        std::vector<const float*> tables;
        std::vector<float> expanded;   // --expand-forcings: the reference's per-stream block of the whole domain
        long long ld = fb.ld;
        const int32_t* col = col_of_stream.data();
        if (!fb.nT.empty()) {
            size_t rows = 0;
            for (size_t n : fb.nT) rows += n;
            const float* base = fb.data.data();
            if (o.expand_forcings) {
                expanded.resize(rows * (size_t)ns);
                for (size_t t = 0; t < rows; ++t)
                    for (int s = 0; s < ns; ++s) expanded[t * (size_t)ns + s] = fb.data[t * (size_t)fb.ld + (size_t)col_of_stream[s]];
                base = expanded.data();
                ld = ns;
                col = nullptr;
            }
            size_t row = 0;
            for (size_t n : fb.nT) { tables.push_back(base + row * (size_t)ld); row += n; }
        }
        rk45_api::ShardedStats st;
        if (nq > 0) {
            const int W = (int)std::min<size_t>(nq, 256);
            h_y_final = rk45_api::run_rk45_sharded_streamed<Model>(
                h_y0, t0, tf, h_query_times, W,
                [&](int, int s0, int cnt, int qb, int qe, const double* rows) {   // (disjoint hillslopes per shard: no lock needed)
                    for (int s = 0; s < cnt; ++s)
                        std::copy(rows + (size_t)s * (qe - qb) * width, rows + (size_t)(s + 1) * (qe - qb) * width,
                                  h_dense.begin() + ((size_t)(s0 + s) * nq + (size_t)qb) * width);
                },
                spatial.data(), tables, fb.dt_hours, fb.nT, ld, fb.nT.empty() ? nullptr : col, G, devices, &st);
        } else
        h_y_final = rk45_api::run_rk45_sharded<Model>(h_y0, t0, tf, spatial.data(), tables, fb.dt_hours, fb.nT, ld, fb.nT.empty() ? nullptr : col,
                                                      G, devices, o.gather == "rccl", &st);
        const rk45_api::RunInfo info = rk45_api::last_run_info();
        stiff = info.stiff;
        n_attempt = info.n_attempt;
        t_h2d[0] = st.upload_ms * 1e-3;
        t_integrate[0] = st.integrate_ms * 1e-3;
        t_d2h[0] = (st.gather_ms + st.download_ms) * 1e-3;
        char buf[384];
        std::snprintf(buf, sizeof(buf), "{\"mode\": \"%s\", \"devices\": %d, \"shards\": %d, \"gather_ms\": %.4f, \"gather_bytes\": %llu, "
                      "\"comm_init_ms\": %.3f, \"rccl_version\": %d, \"rccl_nranks\": %d, \"upload_mode\": %d, \"dense_bytes\": %llu}", o.gather.c_str(), st.num_devices, st.num_shards, st.gather_ms,
                      (unsigned long long)st.gather_bytes, st.comm_init_ms, st.rccl_version, st.rccl_nranks, st.upload_mode, (unsigned long long)st.dense_bytes);
        gather_note = buf;
    } else if (o.gather != "threads") {
        usage("--gather must be threads, rccl or host");
    } else if (G == 1) {
        run_shard(0);
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < G; ++r) th.emplace_back(run_shard, r);
        for (auto& t : th) t.join();
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    for (int r = 0; r < G; ++r)
        if (!errors[r].empty()) throw std::runtime_error("shard " + std::to_string(r) + ": " + errors[r]);

    tp = clk::now();
    if (o.out_format == "csv" || o.out_format == "both") {
        hlm_io::write_final_csv(o.out_final, h_y_final, ns, N_EQ, style);
        if (nq > 0) hlm_io::write_dense_csv(o.out_dense, h_dense, h_query_times, ns, N_EQ, style);  // (--nq 0: no dense file)
    }
    if (o.out_format == "nc" || o.out_format == "both") {
#ifdef HLM_WITH_HDF5
        // main.cpp:786-823: system = 0..ns-1 (not link ids), variable = 0..N_EQ-1
        std::vector<int> linkid_vals(ns), state_vals(N_EQ);
        for (int s = 0; s < ns; ++s) linkid_vals[s] = s;
        for (int v = 0; v < N_EQ; ++v) state_vals[v] = v;
        auto nc_name = [](std::string p) { const size_t d = p.rfind('.'); return (d == std::string::npos ? p : p.substr(0, d)) + ".nc"; };
        if (o.states.empty()) {
            write_final_netcdf(nc_name(o.out_final), h_y_final.data(), linkid_vals.data(), state_vals.data(), ns, N_EQ, o.nc_compression);
            if (nq > 0 && !stream_dense)
                write_dense_netcdf(nc_name(o.out_dense), h_dense.data(), h_query_times.data(), linkid_vals.data(), state_vals.data(),
                                   (int)nq, ns, N_EQ, o.nc_compression);
        } else {
            // config output.states: only these columns; the "variable" coordinate keeps their model indices.
            // Final states always come back whole (they are the next run's initial state); the dense rows
            // arrive already reduced when the device made the selection (device_select above).
            const int K = (int)o.states.size();
            std::vector<double> yf((size_t)ns * K), dn_host;
            for (size_t r = 0; r < (size_t)ns; ++r)
                for (int k = 0; k < K; ++k) yf[r * K + k] = h_y_final[r * N_EQ + o.states[k]];
            if (width != K && !stream_dense) {
                dn_host.resize((size_t)ns * nq * K);
                for (size_t r = 0; r < (size_t)ns * nq; ++r)
                    for (int k = 0; k < K; ++k) dn_host[r * K + k] = h_dense[r * N_EQ + o.states[k]];
            }
            const double* dn = width != K ? dn_host.data() : h_dense.data();
            write_final_netcdf(nc_name(o.out_final), yf.data(), linkid_vals.data(), o.states.data(), ns, K, o.nc_compression);
            if (nq > 0 && !stream_dense)
                write_dense_netcdf(nc_name(o.out_dense), dn, h_query_times.data(), linkid_vals.data(), o.states.data(), (int)nq, ns, K,
                                   o.nc_compression);
        }
#else
        usage("--out-format nc: this binary was built without HDF5 (HLM_WITH_HDF5)");
#endif
    }

    t_write = since(tp);

    unsigned long long attempts = 0;
    int flagged = 0, unsolved = 0;  // flagged by RK45; of those, not finished by the Radau fallback either
    for (int s = 0; s < ns; ++s) {
        attempts += n_attempt[s];
        flagged += stiff[s] != 0;
        unsolved += o.radau ? stiff[s] == HLM_SYS_CAPPED : stiff[s] != 0;
    }
    if (!o.quiet)
        std::printf("{\"model\": %d, \"num_systems\": %d, \"num_queries\": %zu, \"t0\": %g, \"tf\": %g, "
                    "\"initial_step\": %g, \"gpus\": %d, \"attempted_steps\": %llu, \"flagged\": %d, \"radau\": %s, \"unsolved\": %d, "
                    "\"run_rk45_seconds\": %.6f, \"dense_row_width\": %d, \"device_output_selection\": %s, "
                    "\"phases\": {\"params_csv_s\": %.6f, \"forcing_read_s\": %.6f, \"h2d_s\": %.6f, \"integrate_s\": %.6f, "
                    "\"d2h_s\": %.6f, \"write_s\": %.6f, \"window_sink_s\": %.6f}, \"streamed_dense_window\": %d, \"gather\": %s, \"final\": \"%s\", \"dense\": \"%s\"}\n",
                    (int)Model::UID, ns, nq, t0, tf, hp.initialStep, G, attempts, flagged, o.radau ? "true" : "false", unsolved, secs,
                    width, device_select ? "true" : "false", t_params, t_forcing_read,
                    *std::max_element(t_h2d.begin(), t_h2d.end()), *std::max_element(t_integrate.begin(), t_integrate.end()),
                    *std::max_element(t_d2h.begin(), t_d2h.end()), t_write, t_sink, stream_dense ? o.stream_dense : 0, gather_note.c_str(), o.out_final.c_str(),
                    o.out_dense.c_str());
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        const Options o = parse(argc, argv);
        if (o.model == "204") return run<Model204>(o, hlm_io::CsvStyle::Model204);
        if (o.model == "dummy") return run<DummyModel>(o, hlm_io::CsvStyle::Dummy);
        if (o.model == "toy3") return run<ToyModel3>(o, hlm_io::CsvStyle::Dummy);
        usage("--model must be dummy, 204 or toy3");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rk45_solver: %s\n", e.what());
        return 1;
    }
}
