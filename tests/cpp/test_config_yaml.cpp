// Unit test of include/hlm/config_yaml.hpp: the YAML subset parser and the schema mapping.
// Built and run by tests/test_host_logic.py (g++ -std=c++17, no GPU).
#include <cmath>
#include <cstdio>
#include <string>

#include "hlm/config_yaml.hpp"

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

template <class F>
static bool throws(F f, const char* needle)
{
    try { f(); } catch (const std::exception& e) { return std::string(e.what()).find(needle) != std::string::npos; }
    return false;
}

// "dump FILE": the parsed tree as one "path=value" line per scalar (sequence items as [i], null as ~),
// for the comparison with PyYAML in tests/test_host_logic.py
static void dump(const hlm_config::Node& n, const std::string& path)
{
    using hlm_config::Node;
    if (n.kind == Node::Map) {
        if (n.map.empty()) std::printf("%s={}\n", path.c_str());
        for (const auto& kv : n.map) dump(kv.second, path.empty() ? kv.first : path + "." + kv.first);
    } else if (n.kind == Node::Seq) {
        if (n.seq.empty()) std::printf("%s=[]\n", path.c_str());
        for (size_t i = 0; i < n.seq.size(); ++i) dump(n.seq[i], path + "[" + std::to_string(i) + "]");
    } else if (n.kind == Node::Null) {
        std::printf("%s=~\n", path.c_str());
    } else {
        std::printf("%s=%s\n", path.c_str(), n.scalar.c_str());
    }
}

int main(int argc, char** argv)
{
    using namespace hlm_config;
    if (argc > 2 && std::string(argv[1]) == "dump") {
        try {
            dump(load_yaml_file(argv[2]), "");
            (void)from_yaml(load_yaml_file(argv[2]), "");
        } catch (const std::exception& e) {
            std::printf("ERROR %s\n", e.what());
            return 1;
        }
        return 0;
    }
    // ---- parser ------------------------------------------------------------------------------
    const char* doc =
        "# header comment\n"
        "---\n"
        "model:\n"
        "  uid: 204          # trailing comment\n"
        "  name: \"Model # 204\"\n"
        "time:\n"
        "  start: \"2021-01-01T00:00:00\"\n"
        "  end:   '2021-10-01T00:00:00'\n"
        "global_params:\n"
        "  - name: foo\n"
        "    value: 0.5\n"
        "  - name: bar\n"
        "    value: -2\n"
        "output:\n"
        "  print_interval: \"15m\"\n"
        "  states:\n"
        "    - 0   # snow\n"
        "    - 4\n"
        "same_indent:\n"
        "- a\n"
        "- b\n"
        "flow: [1, 2.5, \"x, y\", ]\n"
        "nothing:\n"
        "tilde: ~\n"
        "solver:\n"
        "  tolerances:\n"
        "    rtol:      1e-6\n"
        "  initial_step: null\n";
    const Node r = parse_yaml(doc);
    CHECK(r.kind == Node::Map);
    CHECK(r.path("model.uid") && r.path("model.uid")->as_int() == 204);
    CHECK(r.path("model.name")->as_string() == "Model # 204");
    CHECK(r.path("time.end")->as_string() == "2021-10-01T00:00:00");
    CHECK(r.find("global_params")->kind == Node::Seq && r.find("global_params")->seq.size() == 2);
    CHECK(r.find("global_params")->seq[1].find("name")->as_string() == "bar");
    CHECK(r.find("global_params")->seq[1].find("value")->as_double() == -2.0);
    CHECK(r.path("output.states")->seq.size() == 2 && r.path("output.states")->seq[1].as_int() == 4);
    CHECK(r.find("same_indent")->kind == Node::Seq && r.find("same_indent")->seq[1].as_string() == "b");
    CHECK(r.find("flow")->seq.size() == 4 && r.find("flow")->seq[2].as_string() == "x, y" && r.find("flow")->seq[3].is_null());
    CHECK(r.find("nothing")->is_null() && r.find("tilde")->is_null());
    CHECK(r.path("solver.initial_step")->is_null());
    CHECK(r.path("solver.tolerances.rtol")->as_double() == 1e-6);
    CHECK(r.path("solver.missing.key") == nullptr);
    CHECK(parse_yaml("").is_null() && parse_yaml("# only a comment\n").is_null());
    CHECK(throws([] { parse_yaml("a: 1\n\tb: 2\n"); }, "tab"));
    CHECK(throws([] { parse_yaml("a: 1\na: 2\n"); }, "duplicate"));
    CHECK(throws([] { parse_yaml("a: &anchor 1\n"); }, "unsupported"));
    CHECK(throws([] { parse_yaml("a:\n    b: 1\n  c: 2\n"); }, "indentation"));
    CHECK(throws([] { parse_yaml("a: \"open\n"); }, "unterminated"));
    CHECK(throws([] { parse_yaml("just a scalar line\n"); }, "key: value"));
    CHECK(throws([&] { (void)r.path("model.name")->as_double(); }, "not a number"));

    // ---- helpers -------------------------------------------------------------------------------
    CHECK(interval_minutes("15m") == 15.0 && interval_minutes("1h") == 60.0 && interval_minutes("1d") == 1440.0);
    CHECK(interval_minutes("30s") == 0.5 && interval_minutes("7") == 7.0);
    CHECK(throws([] { interval_minutes("1w"); }, "unit"));
    CHECK(iso8601_minutes("1970-01-01T00:00:00") == 0.0);
    CHECK(iso8601_minutes("1970-01-02T01:30:00") == 1440.0 + 90.0);
    CHECK(iso8601_minutes("2021-10-01T00:00:00") - iso8601_minutes("2021-01-01T00:00:00") == 273.0 * 1440.0);
    CHECK(iso8601_minutes("2024-03-01 00:00") - iso8601_minutes("2024-02-28 00:00") == 2.0 * 1440.0);  // leap day
    CHECK(throws([] { iso8601_minutes("yesterday"); }, "ISO-8601"));

    // ---- schema mapping ------------------------------------------------------------------------
    CHECK(throws([] { parse_yaml("model: {uid: [1]}\n"); }, "nested"));
    {
        const Node fm = parse_yaml("columns: {stream_id: 0, name: \"a, b\", none: }\nempty: {}\nlist: []\n");
        CHECK(fm.path("columns.stream_id")->as_int() == 0 && fm.path("columns.name")->as_string() == "a, b");
        CHECK(fm.path("columns.none")->is_null() && fm.find("empty")->kind == Node::Map && fm.find("empty")->map.empty());
        CHECK(fm.find("list")->kind == Node::Seq && fm.find("list")->seq.empty());
    }
    const char* full =
        "model:\n  uid: 204\n"
        "time:\n  start: \"2021-06-01T00:00:00\"\n  end: \"2021-06-03T00:00:00\"\n"
        "initial:\n  mode: hot\n  file: inits/state.csv\n"
        "global_params:\n  - name: foo\n    value: 0.0\n"
        "local_params:\n  file: params/p.csv\n  columns:\n    stream_id: 0\n    next_stream_id: 1\n    params_start: 2\n    num_params: 15\n"
        "forcings:\n  type: folder_nc\n  path: /data/f\n  lookup: lut.csv\n  vars:\n    precipitation: PRCP\n    temperature: Tair\n"
        "output:\n  print_interval: 1h\n  states: [1, 3]\n"
        "solver:\n  method: RK45\n  tolerances:\n    rtol: 1e-5\n    atol: 1e-8\n    safety: 0.8\n    min_scale: 0.1\n    max_scale: 5.0\n  initial_step: 0.25\n"
        "mpi:\n  step_storage: 30\n"
        "flags:\n  uses_dam: false\n  convert_area: false\n";
    const RunConfig f = from_yaml(parse_yaml(full), "/cfg");
    CHECK(f.model_uid == 204 && f.has_time && f.tf_minutes == 2880.0);
    CHECK(f.initial_mode == "hot" && f.initial_file == "/cfg/inits/state.csv");
    CHECK(f.global_params.size() == 1 && f.global_params[0].first == "foo");
    CHECK(f.params_file == "/cfg/params/p.csv" && f.forcing_path == "/data/f" && f.forcing_lookup == "/cfg/lut.csv");
    CHECK(f.forcing_type == "folder_nc" && f.forcing_vars.size() == 2 && f.forcing_vars[0].first == "precipitation" &&
          f.forcing_vars[0].second == "PRCP" && f.forcing_vars[1].second == "Tair");
    CHECK(f.print_interval_min == 60.0 && f.states.size() == 2 && f.states[1] == 3);
    CHECK(f.rtol == 1e-5 && f.atol == 1e-8 && f.safety == 0.8 && f.min_scale == 0.1 && f.max_scale == 5.0);
    CHECK(!f.auto_initial_step && f.initial_step == 0.25);
    const RunConfig d = from_yaml(parse_yaml("model:\n  uid: 0\n"), "");
    CHECK(d.model_uid == 0 && !d.has_time && d.auto_initial_step && d.rtol == 1e-6 && d.atol == 1e-9 && d.states.empty());
    CHECK(throws([] { from_yaml(parse_yaml("modle:\n  uid: 0\n"), ""); }, "unknown section"));
    CHECK(throws([] { from_yaml(parse_yaml("solver:\n  method: Radau\n"), ""); }, "not supported"));
    CHECK(throws([] { from_yaml(parse_yaml("flags:\n  uses_dam: true\n"), ""); }, "uses_dam"));
    CHECK(throws([] { from_yaml(parse_yaml("local_params:\n  columns:\n    params_start: 3\n"), ""); }, "params_start"));
    CHECK(throws([] { from_yaml(parse_yaml("initial:\n  mode: hot\n"), ""); }, "initial.file"));
    CHECK(throws([] { from_yaml(parse_yaml("time:\n  start: \"2021-02-01T00:00:00\"\n  end: \"2021-01-01T00:00:00\"\n"), ""); }, "later"));

    // ---- the shipped example -------------------------------------------------------------------
    if (argc > 1) {
        const std::string path = argv[1];
        const RunConfig e = from_yaml(load_yaml_file(path), path.substr(0, path.rfind('/')));
        CHECK(e.model_uid == 204 && e.tf_minutes == 2880.0 && e.print_interval_min == 60.0 && e.states.size() == 5);
        CHECK(e.params_file.find("tests/golden/small_test.csv") != std::string::npos && e.forcing_type.empty());
        CHECK(e.auto_initial_step && e.global_params.empty());
    }
    if (failures == 0) std::printf("config_yaml: all checks passed\n");
    return failures ? 1 : 0;
}
