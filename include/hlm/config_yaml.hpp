// Run configuration from the YAML schema of the reference's data/config.yaml.
//
// The reference ships that file as a trial schema (sections model / time / initial /
// global_params / local_params / forcings / output / solver / mpi / flags) but has no reader for
// it: every value is hard-coded in src/main.cpp (paths :272,371,495; y0 :376; t0/tf :610-611;
// tolerances :621; query grid :653-656) and yaml-cpp is not a dependency.  This header gives the
// driver (tools/rk45_solver --config FILE) a reader for that schema:
//   * a parser for the YAML subset the schema uses -- block mappings and sequences nested by
//     indentation, "- key: value" items, one-level flow collections [a, b] / {k: v}, plain and
//     quoted scalars, comments, null / ~ / empty values.  No anchors, tags, nested flow
//     collections or multi-line scalars: those raise std::runtime_error with the line number;
//   * RunConfig, the schema mapped onto what the hot path needs, with the reference's hard-coded
//     values as defaults, and checks for settings this engine cannot honour.
// Header-only, no dependency beyond the C++17 standard library.
#pragma once

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace hlm_config {

struct Node {
    enum Kind { Null, Scalar, Map, Seq };
    Kind kind = Null;
    std::string scalar;
    bool quoted = false;  // a quoted scalar is never null / bool / number
    std::vector<std::pair<std::string, Node>> map;
    std::vector<Node> seq;
    int line = 0;

    const Node* find(const std::string& key) const
    {
        if (kind != Map) return nullptr;
        for (const auto& kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    // "a.b.c" through nested mappings; nullptr if any level is missing
    const Node* path(const std::string& dotted) const
    {
        const Node* n = this;
        size_t a = 0;
        while (n && a <= dotted.size()) {
            const size_t b = dotted.find('.', a);
            n = n->find(dotted.substr(a, b == std::string::npos ? std::string::npos : b - a));
            if (b == std::string::npos) break;
            a = b + 1;
        }
        return n;
    }
    bool is_null() const { return kind == Null; }
    std::string as_string() const
    {
        if (kind != Scalar) throw std::runtime_error("line " + std::to_string(line) + ": expected a scalar");
        return scalar;
    }
    double as_double() const
    {
        const std::string s = as_string();
        char* end = nullptr;
        const double v = std::strtod(s.c_str(), &end);
        if (quoted || end == s.c_str() || *end != '\0')
            throw std::runtime_error("line " + std::to_string(line) + ": '" + s + "' is not a number");
        return v;
    }
    long as_int() const
    {
        const double v = as_double();
        if (v != std::floor(v)) throw std::runtime_error("line " + std::to_string(line) + ": '" + scalar + "' is not an integer");
        return (long)v;
    }
    bool as_bool() const
    {
        const std::string s = as_string();
        if (!quoted && (s == "true" || s == "True" || s == "yes" || s == "on")) return true;
        if (!quoted && (s == "false" || s == "False" || s == "no" || s == "off")) return false;
        throw std::runtime_error("line " + std::to_string(line) + ": '" + s + "' is not a boolean");
    }
};

namespace detail {

struct Line {
    int indent;
    std::string text;  // without indentation, comment and trailing blanks
    int number;
};

[[noreturn]] inline void fail(int line, const std::string& what)
{
    throw std::runtime_error("yaml line " + std::to_string(line) + ": " + what);
}

inline std::string rtrim(std::string s)
{
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
    return s;
}

// strips a trailing comment (a '#' at the start or after a blank, outside quotes)
inline std::string strip_comment(const std::string& s)
{
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (q) {
            if (c == q) q = 0;
        } else if (c == '"' || c == '\'') {
            q = c;
        } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
            return s.substr(0, i);
        }
    }
    return s;
}

inline std::vector<Line> split_lines(const std::string& text)
{
    std::vector<Line> out;
    std::istringstream in(text);
    std::string raw;
    int number = 0;
    while (std::getline(in, raw)) {
        ++number;
        size_t ind = 0;
        while (ind < raw.size() && raw[ind] == ' ') ++ind;
        if (ind < raw.size() && raw[ind] == '\t') fail(number, "tab used for indentation");
        std::string body = rtrim(strip_comment(raw.substr(ind)));
        if (body.empty() || body == "---" || body == "...") continue;
        if (body[0] == '%') continue;  // directive
        out.push_back({(int)ind, body, number});
    }
    return out;
}

inline Node scalar_node(std::string s, int line)
{
    Node n;
    n.line = line;
    s = rtrim(s);
    size_t a = 0;
    while (a < s.size() && s[a] == ' ') ++a;
    s = s.substr(a);
    if (s.empty() || s == "~" || s == "null" || s == "Null" || s == "NULL") return n;  // Null
    if (s[0] == '&' || s[0] == '*' || s[0] == '!' || s[0] == '|' || s[0] == '>' || s[0] == '{' || s[0] == '[')
        fail(line, "unsupported YAML construct '" + s.substr(0, 1) + "' (anchors, tags, block scalars, nested flow collections)");
    n.kind = Node::Scalar;
    if (s.size() >= 2 && (s[0] == '"' || s[0] == '\'') && s.back() == s[0]) {
        n.quoted = true;
        n.scalar = s.substr(1, s.size() - 2);
    } else if (s[0] == '"' || s[0] == '\'') {
        fail(line, "unterminated quoted scalar");
    } else {
        n.scalar = s;
    }
    return n;
}

// [a, b, "c d"]
inline Node flow_seq(const std::string& s, int line)
{
    Node n;
    n.kind = Node::Seq;
    n.line = line;
    if (s.back() != ']') fail(line, "flow sequence must close on the same line");
    const std::string body = s.substr(1, s.size() - 2);
    std::string cur;
    char q = 0;
    bool any = false;
    for (char c : body) {
        if (q) {
            cur += c;
            if (c == q) q = 0;
        } else if (c == '"' || c == '\'') {
            q = c;
            cur += c;
        } else if (c == '[' || c == '{') {
            fail(line, "nested flow collections are not supported");
        } else if (c == ',') {
            n.seq.push_back(scalar_node(cur, line));
            cur.clear();
            any = true;
        } else {
            cur += c;
        }
    }
    if (!rtrim(cur).empty() || any) n.seq.push_back(scalar_node(cur, line));
    return n;
}

// position of the ':' that ends a mapping key (followed by a blank or the end), or npos
inline size_t key_colon(const std::string& s)
{
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (q) {
            if (c == q) q = 0;
        } else if (c == '"' || c == '\'') {
            q = c;
        } else if (c == ':' && (i + 1 == s.size() || s[i + 1] == ' ')) {
            return i;
        }
    }
    return std::string::npos;
}

// {a: 1, b: "x"} -- one level, scalar values
inline Node flow_map(const std::string& s, int line)
{
    Node n;
    n.kind = Node::Map;
    n.line = line;
    if (s.back() != '}') fail(line, "flow mapping must close on the same line");
    const std::string body = s.substr(1, s.size() - 2);
    std::vector<std::string> items;
    std::string cur;
    char q = 0;
    for (char c : body) {
        if (q) {
            cur += c;
            if (c == q) q = 0;
        } else if (c == '"' || c == '\'') {
            q = c;
            cur += c;
        } else if (c == '[' || c == '{') {
            fail(line, "nested flow collections are not supported");
        } else if (c == ',') {
            items.push_back(cur);
            cur.clear();
        } else {
            cur += c;
        }
    }
    if (!rtrim(cur).empty()) items.push_back(cur);
    for (std::string it : items) {
        size_t a = 0;
        while (a < it.size() && it[a] == ' ') ++a;
        it = rtrim(it.substr(a));
        const size_t c = key_colon(it);
        if (c == std::string::npos) fail(line, "expected 'key: value' inside { }");
        const std::string key = rtrim(it.substr(0, c));
        if (n.find(key)) fail(line, "duplicate key '" + key + "'");
        n.map.emplace_back(key, scalar_node(it.substr(c + 1), line));
    }
    return n;
}

inline Node value_node(const std::string& s, int line)
{
    size_t a = 0;
    while (a < s.size() && s[a] == ' ') ++a;
    if (a < s.size() && s[a] == '[') return flow_seq(rtrim(s.substr(a)), line);
    if (a < s.size() && s[a] == '{') return flow_map(rtrim(s.substr(a)), line);
    return scalar_node(s, line);
}

inline bool is_seq_item(const std::string& s) { return s[0] == '-' && (s.size() == 1 || s[1] == ' '); }

Node parse_block(std::vector<Line>& L, size_t& pos, int indent);

inline Node parse_map(std::vector<Line>& L, size_t& pos, int indent)
{
    Node n;
    n.kind = Node::Map;
    n.line = L[pos].number;
    while (pos < L.size() && L[pos].indent == indent && !is_seq_item(L[pos].text)) {
        const Line ln = L[pos];
        const size_t c = key_colon(ln.text);
        if (c == std::string::npos) fail(ln.number, "expected 'key: value'");
        std::string key = rtrim(ln.text.substr(0, c));
        if (key.size() >= 2 && (key[0] == '"' || key[0] == '\'') && key.back() == key[0]) key = key.substr(1, key.size() - 2);
        if (n.find(key)) fail(ln.number, "duplicate key '" + key + "'");
        const std::string rest = ln.text.substr(c + 1);
        ++pos;
        Node v;
        if (!rtrim(rest).empty()) {
            v = value_node(rest, ln.number);
        } else if (pos < L.size() && L[pos].indent > indent) {
            v = parse_block(L, pos, L[pos].indent);
        } else if (pos < L.size() && L[pos].indent == indent && is_seq_item(L[pos].text)) {
            v = parse_block(L, pos, indent);  // a sequence may sit at its key's indentation
        } else {
            v.line = ln.number;  // empty value: null
        }
        n.map.emplace_back(std::move(key), std::move(v));
    }
    if (pos < L.size() && L[pos].indent > indent) fail(L[pos].number, "unexpected indentation");
    return n;
}

inline Node parse_seq(std::vector<Line>& L, size_t& pos, int indent)
{
    Node n;
    n.kind = Node::Seq;
    n.line = L[pos].number;
    while (pos < L.size() && L[pos].indent == indent && is_seq_item(L[pos].text)) {
        const Line ln = L[pos];
        std::string rest = ln.text.size() > 1 ? ln.text.substr(2) : std::string();
        size_t lead = 0;
        while (lead < rest.size() && rest[lead] == ' ') ++lead;
        rest = rest.substr(lead);
        if (rest.empty()) {
            ++pos;
            if (pos < L.size() && L[pos].indent > indent) n.seq.push_back(parse_block(L, pos, L[pos].indent));
            else { Node nul; nul.line = ln.number; n.seq.push_back(nul); }
        } else if (rest[0] != '[' && key_colon(rest) != std::string::npos) {
            // "- key: value": a mapping whose first entry shares the dash's line
            const int inner = indent + 2 + (int)lead;
            L[pos].indent = inner;
            L[pos].text = rest;
            n.seq.push_back(parse_map(L, pos, inner));
        } else {
            n.seq.push_back(value_node(rest, ln.number));
            ++pos;
        }
    }
    return n;
}

inline Node parse_block(std::vector<Line>& L, size_t& pos, int indent)
{
    return is_seq_item(L[pos].text) ? parse_seq(L, pos, indent) : parse_map(L, pos, indent);
}

}  // namespace detail

inline Node parse_yaml(const std::string& text)
{
    std::vector<detail::Line> L = detail::split_lines(text);
    Node root;
    if (L.empty()) return root;
    size_t pos = 0;
    root = detail::parse_block(L, pos, L[0].indent);
    if (pos != L.size()) detail::fail(L[pos].number, "inconsistent indentation");
    return root;
}

inline Node load_yaml_file(const std::string& path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open config " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_yaml(ss.str());
}

// "15m", "1h", "1d", "30s", or a bare number of minutes -> minutes
inline double interval_minutes(const std::string& s)
{
    char* end = nullptr;
    const double v = std::strtod(s.c_str(), &end);
    if (end == s.c_str() || !(v > 0)) throw std::runtime_error("bad interval '" + s + "'");
    std::string unit(end);
    while (!unit.empty() && unit[0] == ' ') unit.erase(0, 1);
    if (unit.empty() || unit == "m" || unit == "min") return v;
    if (unit == "h") return v * 60.0;
    if (unit == "d") return v * 1440.0;
    if (unit == "s") return v / 60.0;
    throw std::runtime_error("bad interval unit in '" + s + "' (want s, m, h or d)");
}

// "YYYY-MM-DDTHH:MM[:SS]" (or with a blank instead of 'T') -> minutes since 1970-01-01T00:00
inline double iso8601_minutes(const std::string& s)
{
    int Y = 0, M = 0, D = 0, h = 0, m = 0;
    double sec = 0.0;
    char sep = 0;
    const int got = std::sscanf(s.c_str(), "%d-%d-%d%c%d:%d:%lf", &Y, &M, &D, &sep, &h, &m, &sec);
    if (got < 3 || (got > 3 && sep != 'T' && sep != ' ') || M < 1 || M > 12 || D < 1 || D > 31 || h < 0 || h > 24 || m < 0 || m > 59)
        throw std::runtime_error("bad ISO-8601 time '" + s + "'");
    // days from civil (proleptic Gregorian)
    const int y = M <= 2 ? Y - 1 : Y;
    const long era = (y >= 0 ? y : y - 399) / 400;
    const long yoe = y - era * 400;
    const long doy = (153 * (M + (M > 2 ? -3 : 9)) + 2) / 5 + D - 1;
    const long doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    const long days = era * 146097 + doe - 719468;
    return (double)days * 1440.0 + h * 60.0 + m + sec / 60.0;
}

// The schema mapped onto the driver.  Unset optional values keep the reference's hard-coded ones.
struct RunConfig {
    int model_uid = 204;
    std::string model_name;
    bool has_time = false;
    std::string time_start, time_end;
    double tf_minutes = -1.0;          // end - start; t0 = 0 at time.start (main.cpp:610)
    std::string initial_mode = "cold"; // cold: the model's constant y0 (main.cpp:376); hot: initial_file
    std::string initial_file;          // CSV, one row of N_EQ states per hillslope in parameter-file order
    std::vector<std::pair<std::string, double>> global_params;  // parsed, unused by Model 204 / Dummy
    std::string params_file;
    std::string forcing_type;          // "folder_nc" | "" (stub forcings)
    std::string forcing_path, forcing_lookup;
    std::vector<std::pair<std::string, std::string>> forcing_vars;  // (role, NetCDF variable) in file order
    double print_interval_min = -1.0;
    std::vector<int> states;           // empty = all
    double rtol = 1e-6, atol = 1e-9, safety = 0.9, min_scale = 0.2, max_scale = 10.0;  // model_204.hpp:22-30
    bool auto_initial_step = true;     // initial_step: null
    double initial_step = 0.01;
};

// `base_dir`: relative paths in the file are resolved against it (normally the config's directory)
inline RunConfig from_yaml(const Node& root, const std::string& base_dir)
{
    if (root.kind != Node::Map) throw std::runtime_error("config: top level must be a mapping");
    auto resolve = [&](const std::string& p) {
        if (p.empty() || p[0] == '/' || base_dir.empty()) return p;
        return base_dir + "/" + p;
    };
    static const char* known[] = {"model", "time", "initial", "global_params", "local_params",
                                  "forcings", "output", "solver", "mpi", "flags"};
    for (const auto& kv : root.map) {
        bool ok = false;
        for (const char* k : known) ok = ok || kv.first == k;
        if (!ok) throw std::runtime_error("config line " + std::to_string(kv.second.line) + ": unknown section '" + kv.first + "'");
    }
    RunConfig c;
    if (const Node* n = root.path("model.uid")) c.model_uid = (int)n->as_int();
    if (const Node* n = root.path("model.name")) if (!n->is_null()) c.model_name = n->as_string();

    const Node* ts = root.path("time.start");
    const Node* te = root.path("time.end");
    if (ts && te && !ts->is_null() && !te->is_null()) {
        c.has_time = true;
        c.time_start = ts->as_string();
        c.time_end = te->as_string();
        c.tf_minutes = iso8601_minutes(c.time_end) - iso8601_minutes(c.time_start);
        if (!(c.tf_minutes > 0)) throw std::runtime_error("config: time.end must be later than time.start");
    }

    if (const Node* n = root.path("initial.mode")) c.initial_mode = n->as_string();
    if (c.initial_mode != "cold" && c.initial_mode != "hot")
        throw std::runtime_error("config: initial.mode must be cold or hot, not '" + c.initial_mode + "'");
    if (c.initial_mode == "hot") {
        const Node* f = root.path("initial.file");
        if (!f || f->is_null()) throw std::runtime_error("config: initial.mode hot needs initial.file");
        c.initial_file = resolve(f->as_string());
    }

    if (const Node* g = root.find("global_params"))
        if (g->kind == Node::Seq)
            for (const Node& it : g->seq) {
                const Node* nm = it.find("name");
                const Node* v = it.find("value");
                if (!nm || !v) throw std::runtime_error("config line " + std::to_string(it.line) + ": global_params items need name and value");
                c.global_params.emplace_back(nm->as_string(), v->as_double());
            }

    if (const Node* f = root.path("local_params.file")) if (!f->is_null()) c.params_file = resolve(f->as_string());
    if (const Node* cols = root.path("local_params.columns")) {
        // the loader's layout is fixed (parameters_loader.cpp:52-93): id, next id, then 15 parameters
        const struct { const char* key; long want; } fixed[] = {{"stream_id", 0}, {"next_stream_id", 1}, {"params_start", 2}, {"num_params", 15}};
        for (const auto& f : fixed)
            if (const Node* v = cols->find(f.key))
                if (v->as_int() != f.want)
                    throw std::runtime_error(std::string("config: local_params.columns.") + f.key + " must be " + std::to_string(f.want) +
                                             " (the SpatialParams CSV layout is fixed)");
    }

    if (const Node* fo = root.find("forcings")) {
        if (fo->kind == Node::Map) {
            if (const Node* t = fo->find("type")) if (!t->is_null()) c.forcing_type = t->as_string();
            if (!c.forcing_type.empty() && c.forcing_type != "folder_nc" && c.forcing_type != "none")
                throw std::runtime_error("config: forcings.type '" + c.forcing_type + "' is not supported (folder_nc or none)");
            if (c.forcing_type == "none") c.forcing_type.clear();
            if (const Node* p = fo->find("path")) if (!p->is_null()) c.forcing_path = resolve(p->as_string());
            if (const Node* l = fo->find("lookup")) if (!l->is_null()) c.forcing_lookup = resolve(l->as_string());
            if (const Node* v = fo->find("vars"))
                if (v->kind == Node::Map)
                    for (const auto& kv : v->map) c.forcing_vars.emplace_back(kv.first, kv.second.as_string());
        }
    }

    if (const Node* p = root.path("output.print_interval")) if (!p->is_null()) c.print_interval_min = interval_minutes(p->as_string());
    if (const Node* st = root.path("output.states"))
        if (st->kind == Node::Seq)
            for (const Node& it : st->seq) c.states.push_back((int)it.as_int());

    if (const Node* m = root.path("solver.method"))
        if (!m->is_null() && m->as_string() != "RK45")
            throw std::runtime_error("config: solver.method '" + m->as_string() + "' is not supported (RK45)");
    const struct { const char* key; double* dst; } tol[] = {{"rtol", &c.rtol}, {"atol", &c.atol}, {"safety", &c.safety},
                                                            {"min_scale", &c.min_scale}, {"max_scale", &c.max_scale}};
    for (const auto& t : tol)
        if (const Node* v = root.path(std::string("solver.tolerances.") + t.key)) if (!v->is_null()) *t.dst = v->as_double();
    if (const Node* h = root.path("solver.initial_step")) {
        if (!h->is_null()) {
            c.auto_initial_step = false;
            c.initial_step = h->as_double();
            if (!(c.initial_step > 0)) throw std::runtime_error("config: solver.initial_step must be > 0 or null");
        }
    }

    // mpi.*: buffers of the reference's MPI transport; this engine shards over GPUs instead -> ignored
    for (const char* flag : {"uses_dam", "convert_area"})
        if (const Node* v = root.path(std::string("flags.") + flag))
            if (!v->is_null() && v->as_bool())
                throw std::runtime_error(std::string("config: flags.") + flag + " = true is not supported (the reference has no such code path either)");
    return c;
}

}  // namespace hlm_config
