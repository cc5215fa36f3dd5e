// magnetite_gpu -- compiled caller with the stage order of the reference's entry() (main.rs:54-76):
//     mesher (input JSON + boundary rules; the mesh itself comes pre-generated as a Gmsh MSH-4 ASCII file, because
//     the reference's `gmsh` subprocess, mesher.rs:501-506, is outside the hot path) -> solver::run -> csv_output.
// Everything numerical happens in libmagnetite_hip.so through include/magnetite_solver.hpp; this file is glue:
//   * input JSON            mesher.rs:713-808   (tiny order-preserving JSON reader below; no third-party library)
//   * boundary rules        mesher.rs:815-930   strict region test, every matching rule overwrites, later rules win
//   * MSH-4 ASCII           mesher.rs:536-704   + check_ccw with its `< 1.0` quirk (mesher.rs:522-526)
//   * nodes.csv/elements.csv post_processor.rs:18-83, floats as Rust's `{}` prints them
// Usage: magnetite_gpu <input.json> <mesh.msh> [--nodes nodes.csv] [--elements elements.csv] [--dry-run] [--rel TOL]
//   --dry-run  stop before the solver and print what was parsed (no GPU needed)
//   --rel TOL  stop CG on relative residual TOL instead of the reference's absolute 1e-4
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "magnetite_solver.hpp"

using namespace magnetite;

namespace {

[[noreturn]] void die(const MagnetiteError &e)
{
    std::fprintf(stderr, "Received error: %s\n", e.display().c_str());  // main.rs:44-50
    std::exit(1);
}

// ---- order-preserving JSON (objects keep the file's key order: boundary rules apply in that order) ----
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json *get(const std::string &k) const
    {
        for (const auto &kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool has(const std::string &k) const { return get(k) != nullptr; }
    std::optional<double> as_f64() const { return kind == Num ? std::optional<double>(num) : std::nullopt; }
};

struct JsonParser {
    const std::string &s;
    size_t i = 0;
    explicit JsonParser(const std::string &text) : s(text) {}
    void ws()
    {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i;
    }
    [[noreturn]] void bad(const char *what) { die({MagnetiteError::Input, std::string("Error in input file json: ") + what}); }
    Json value()
    {
        ws();
        if (i >= s.size()) bad("unexpected end");
        Json v;
        const char c = s[i];
        if (c == '{') {
            v.kind = Json::Obj;
            ++i;
            ws();
            if (i < s.size() && s[i] == '}') { ++i; return v; }
            for (;;) {
                ws();
                Json key = value();
                if (key.kind != Json::Str) bad("object key is not a string");
                ws();
                if (i >= s.size() || s[i] != ':') bad("expected ':'");
                ++i;
                v.obj.emplace_back(key.str, value());
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; break; }
                bad("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = Json::Arr;
            ++i;
            ws();
            if (i < s.size() && s[i] == ']') { ++i; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; break; }
                bad("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = Json::Str;
            ++i;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) ++i;
                v.str.push_back(s[i++]);
            }
            if (i >= s.size()) bad("unterminated string");
            ++i;
        } else if (!s.compare(i, 4, "null")) {
            i += 4;
        } else if (!s.compare(i, 4, "true")) {
            v.kind = Json::Bool;
            v.b = true;
            i += 4;
        } else if (!s.compare(i, 5, "false")) {
            v.kind = Json::Bool;
            i += 5;
        } else {
            char *end = nullptr;
            v.num = std::strtod(s.c_str() + i, &end);
            if (end == s.c_str() + i) bad("unexpected character");
            v.kind = Json::Num;
            i = (size_t)(end - s.c_str());
        }
        return v;
    }
};

std::string slurp(const std::string &path, MagnetiteError::Kind kind, const std::string &msg)
{
    std::ifstream f(path);
    if (!f) die({kind, msg});
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// ---- datatypes.rs:31-52 ----
struct BoundaryRule {
    std::string name;
    double x_min = std::numeric_limits<double>::lowest(), x_max = std::numeric_limits<double>::max();
    double y_min = std::numeric_limits<double>::lowest(), y_max = std::numeric_limits<double>::max();
    std::optional<double> ux, uy, fx, fy;
};

// mesher.rs:769-808
ModelMetadata parse_input_metadata(const Json &doc)
{
    const Json *md = doc.get("metadata");
    auto f = [&](const char *k) { return md && md->get(k) ? md->get(k)->as_f64() : std::nullopt; };
    const auto e = f("material_elasticity"), t = f("part_thickness"), nu = f("poisson_ratio");
    const auto cmin = f("characteristic_length_min"), cmax = f("characteristic_length_max");
    if (!e) die({MagnetiteError::Input, "Input json missing material elasticity"});
    if (!nu) die({MagnetiteError::Input, "Input json missing poisson ratio"});
    if (!cmin) die({MagnetiteError::Input, "Input json missing minimum characteristic length"});
    if (!cmax) die({MagnetiteError::Input, "Input json missing maximum characteristic length"});
    if (!t) die({MagnetiteError::Input, "Input json missing part thickness"});  // the reference unwrap()-panics
    return ModelMetadata{*e, *nu, *t, (float)*cmin, (float)*cmax};
}

// mesher.rs:822-904
std::vector<BoundaryRule> parse_boundary_rules(const Json &doc)
{
    std::vector<BoundaryRule> rules;
    const Json *bc = doc.get("boundary_conditions");
    if (!bc) return rules;
    for (const auto &kv : bc->obj) {
        const std::string &name = kv.first;
        const Json &rj = kv.second;
        if (!rj.has("region")) die({MagnetiteError::Input, "Boundary rule " + name + " is missing region field"});
        if (!rj.has("targets")) die({MagnetiteError::Input, "Boundary rule " + name + " is missing target field"});
        BoundaryRule r;
        r.name = name;
        const Json &reg = *rj.get("region"), &tg = *rj.get("targets");
        auto bound = [&](const char *k, double &dst) {
            if (!reg.has(k)) return;
            const auto v = reg.get(k)->as_f64();
            if (!v) die({MagnetiteError::Input, std::string("Bad value for ") + k + " in " + name});
            dst = *v;
        };
        bound("x_target_min", r.x_min);
        bound("x_target_max", r.x_max);
        bound("y_target_min", r.y_min);
        bound("y_target_max", r.y_max);
        auto target = [&](const char *k) { return tg.get(k) ? tg.get(k)->as_f64() : std::nullopt; };
        r.ux = target("ux");
        r.uy = target("uy");
        r.fx = target("fx");
        r.fy = target("fy");
        const std::string b = "Boundary '" + name + "' ";
        if (r.x_min > r.x_max) die({MagnetiteError::Input, b + "has x_target_min greater than x_target_max"});
        if (r.y_min > r.y_max) die({MagnetiteError::Input, b + "has y_target_min greater than y_target_max"});
        if (!r.fx && !r.ux) die({MagnetiteError::Input, b + "is under-constrained in x-axis"});
        if (!r.fy && !r.uy) die({MagnetiteError::Input, b + "is under-constrained in y-axis"});
        if (r.fx && r.ux) die({MagnetiteError::Input, b + "is over-constrained in x-axis"});
        if (r.fy && r.uy) die({MagnetiteError::Input, b + "is over-constrained in y-axis"});
        rules.push_back(r);
    }
    std::printf("info: loaded %zu boundary rules from input file\n", rules.size());
    return rules;
}

// mesher.rs:913-927
void apply_boundary_conditions(const std::vector<BoundaryRule> &rules, std::vector<Node> &nodes)
{
    for (Node &n : nodes)
        for (const BoundaryRule &r : rules) {
            const bool candidate = n.vertex.x > r.x_min && n.vertex.x < r.x_max && n.vertex.y > r.y_min && n.vertex.y < r.y_max;
            if (candidate) {
                n.ux = r.ux;
                n.uy = r.uy;
                n.fx = r.fx;
                n.fy = r.fy;
            }
        }
}

// mesher.rs:536-704 (+ check_ccw, :522-526)
void parse_mesh(const std::string &path, std::vector<Node> &nodes, std::vector<Element> &elements)
{
    std::ifstream f(path);
    if (!f) die({MagnetiteError::Mesher, "Unable to open auto-generated mesh file: " + path});
    enum { Limbo, Entities, Nodes, Elements } state = Limbo;
    bool skipped_meta = false;
    std::vector<std::pair<size_t, Node>> unordered;
    std::string line;
    auto ints = [&](const std::string &l) {
        std::vector<long long> v;
        std::stringstream ss(l);
        long long x;
        while (ss >> x) v.push_back(x);
        if (v.empty()) die({MagnetiteError::Mesher, "Unexpected non-int in mesh data " + l});
        return v;
    };
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line.rfind("$End", 0) == 0) state = Limbo;
        if (state == Limbo) {
            skipped_meta = false;
            if (line.rfind("$Entities", 0) == 0) state = Entities;
            else if (line.rfind("$Node", 0) == 0) state = Nodes;
            else if (line.rfind("$Elements", 0) == 0) state = Elements;
            continue;
        }
        if (state == Entities) continue;
        if (!skipped_meta) { skipped_meta = true; continue; }
        const auto head = ints(line);
        if (head.size() < 4) die({MagnetiteError::Mesher, "short block header in mesh data"});
        if (state == Nodes) {
            const size_t n_local = (size_t)head[3];
            std::vector<size_t> tags(n_local);
            for (size_t k = 0; k < n_local; ++k) {
                if (!std::getline(f, line)) die({MagnetiteError::Mesher, "truncated node block"});
                tags[k] = (size_t)std::stoll(line);
            }
            for (size_t k = 0; k < n_local; ++k) {
                if (!std::getline(f, line)) die({MagnetiteError::Mesher, "truncated node block"});
                std::stringstream ss(line);
                double x = 0, y = 0;
                ss >> x >> y;
                unordered.push_back({tags[k] - 1, Node{{x, y}, std::nullopt, std::nullopt, 0.0, 0.0}});  // mesher.rs:615-624
            }
        } else {
            const long long entity_dim = head[0];
            const size_t n_local = (size_t)head[3];
            for (size_t k = 0; k < n_local; ++k) {
                if (!std::getline(f, line)) die({MagnetiteError::Mesher, "truncated element block"});
                if (entity_dim != 2) continue;
                const auto md = ints(line);
                if (md.size() < 4) die({MagnetiteError::Mesher, "element with fewer than 3 nodes"});
                elements.push_back({{(size_t)md[1] - 1, (size_t)md[2] - 1, (size_t)md[3] - 1}, std::nullopt});
            }
        }
    }
    nodes.assign(unordered.size(), Node{{0, 0}, std::nullopt, std::nullopt, 0.0, 0.0});
    for (auto &kv : unordered) {
        if (kv.first >= nodes.size()) die({MagnetiteError::Mesher, "node tags are not a permutation of 1..N"});
        nodes[kv.first] = kv.second;
    }
    for (Element &e : elements) {
        for (size_t n : e.nodes)
            if (n >= nodes.size()) die({MagnetiteError::Mesher, "element references a node outside the mesh"});
        if (solver::compute_element_area(e, nodes) < 1.0) std::swap(e.nodes[0], e.nodes[2]);  // check_ccw: reverse()
    }
    std::printf("info: loaded %zu nodes and %zu elements\n", nodes.size(), elements.size());
}

// Rust's `{}` for f64: shortest representation that round-trips, never in exponent form, no trailing ".0"
std::string rust_display(double v)
{
    if (v != v) return "NaN";
    if (v == std::numeric_limits<double>::infinity()) return "inf";
    if (v == -std::numeric_limits<double>::infinity()) return "-inf";
    char buf[512];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

// post_processor.rs:18-83
void csv_output(const std::vector<Element> &elements, const std::vector<Node> &nodes, const std::string &nodes_output,
                const std::string &elements_output)
{
    std::FILE *nf = std::fopen(nodes_output.c_str(), "w");
    if (!nf) die({MagnetiteError::Solver, "Failed to create nodes.csv: " + nodes_output});
    std::FILE *ef = std::fopen(elements_output.c_str(), "w");
    if (!ef) die({MagnetiteError::Solver, "Failed to create elements.csv: " + elements_output});
    std::fputs("x,y,ux,uy\n", nf);
    for (const Node &n : nodes)
        std::fprintf(nf, "%s,%s,%s,%s\n", rust_display(n.vertex.x).c_str(), rust_display(n.vertex.y).c_str(),
                     rust_display(n.ux.value()).c_str(), rust_display(n.uy.value()).c_str());
    std::fputs("n0,n1,n2,stress\n", ef);
    for (const Element &e : elements)
        std::fprintf(ef, "%zu,%zu,%zu,%s\n", e.nodes[0], e.nodes[1], e.nodes[2], rust_display(e.stress.value()).c_str());
    std::fclose(nf);
    std::fclose(ef);
    std::printf("info: wrote output to %s and %s\n", nodes_output.c_str(), elements_output.c_str());
}

}  // namespace

int main(int argc, char **argv)
{
    std::string input, mesh, nodes_out = "nodes.csv", elements_out = "elements.csv";
    bool dry = false;
    double rel = 0.0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--dry-run") dry = true;
        else if (a == "--nodes" && i + 1 < argc) nodes_out = argv[++i];
        else if (a == "--elements" && i + 1 < argc) elements_out = argv[++i];
        else if (a == "--rel" && i + 1 < argc) rel = std::atof(argv[++i]);
        else if (input.empty()) input = a;
        else if (mesh.empty()) mesh = a;
        else die({MagnetiteError::Input, "Unrecognized argument " + a});
    }
    if (input.empty() || mesh.empty()) {
        std::fprintf(stderr, "usage: magnetite_gpu <input.json> <mesh.msh> [--nodes F] [--elements F] [--dry-run] [--rel TOL]\n");
        return 2;
    }
    // mesher::run (mesher.rs:939-974) minus geometry parsing and the gmsh subprocess
    const std::string text = slurp(input, MagnetiteError::Input, "Unable to open input file " + input);
    JsonParser jp(text);
    const Json doc = jp.value();
    // load_input_file's checks in its order, messages verbatim (mesher.rs:733-755; "in metadata section" for
    // boundary_conditions is the reference's own wording)
    if (!doc.has("metadata")) die({MagnetiteError::Input, "Input json missing metadata field"});
    if (!doc.has("boundary_conditions"))
        die({MagnetiteError::Input, "Input json missing boundary_conditions field in metadata section"});
    for (const char *k : {"part_thickness", "material_elasticity", "poisson_ratio"})
        if (!doc.get("metadata")->has(k))
            die({MagnetiteError::Input, std::string("Input json missing ") + k + " field in metadata section"});
    const ModelMetadata meta = parse_input_metadata(doc);
    std::vector<Node> nodes;
    std::vector<Element> elements;
    parse_mesh(mesh, nodes, elements);
    apply_boundary_conditions(parse_boundary_rules(doc), nodes);
    if (dry) {
        size_t ku = 0, kf = 0;
        double su = 0.0, sf = 0.0;
        for (const Node &n : nodes) {
            ku += n.ux.has_value() + n.uy.has_value();
            kf += n.fx.has_value() + n.fy.has_value();
            su += n.ux.value_or(0.0) + n.uy.value_or(0.0);
            sf += n.fx.value_or(0.0) + n.fy.value_or(0.0);
        }
        std::printf("dry-run: nodes %zu elements %zu prescribed_u %zu prescribed_f %zu sum_u %s sum_f %s E %s nu %s t %s first %zu,%zu,%zu\n",
                    nodes.size(), elements.size(), ku, kf, rust_display(su).c_str(), rust_display(sf).c_str(),
                    rust_display(meta.youngs_modulus).c_str(), rust_display(meta.poisson_ratio).c_str(),
                    rust_display(meta.part_thickness).c_str(), elements.empty() ? 0 : elements[0].nodes[0],
                    elements.empty() ? 0 : elements[0].nodes[1], elements.empty() ? 0 : elements[0].nodes[2]);
        return 0;
    }
    // solver::run (main.rs:64)
    mag_options opt;
    mag_default_options(&opt);
    opt.verbose = 1;
    if (rel > 0.0) {
        opt.stop_mode = MAG_STOP_REL;
        opt.tol = rel;
    }
    if (Result err = solver::run(nodes, elements, meta, &opt)) die(*err);
    // post_processor::csv_output (main.rs:69); the matplotlib plot (main.rs:72) is not part of this tool
    csv_output(elements, nodes, nodes_out, elements_out);
    return 0;
}
