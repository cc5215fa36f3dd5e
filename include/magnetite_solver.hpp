// magnetite_solver.hpp -- C++ host-side mirror of Magnetite's solver interface over the C ABI
// (include/magnetite_hip.h).  Header-only; link with -lmagnetite_hip.
//
// The reference is compiled Rust and this image has no Rust toolchain, so the host side above the C ABI is
// written in C++ with the reference's own names, argument meaning and error behaviour:
//   datatypes.rs:1-29   Vertex, Node, Element, ModelMetadata        (Option<f64> -> std::optional<double>)
//   error.rs:3-22       MagnetiteError{Input,Mesher,Solver,PostProcessor}, Display "<Kind> error: <msg>"
//   solver.rs:17-19     DOF, MAX_CG_ITER, TARGET_CG_COST
//   solver.rs:187-193   compute_element_area (pub; mesher.rs:9,523 imports it)
//   solver.rs:543-547   run(nodes, elements, model_metadata) -> Result<(), MagnetiteError>, mutating in place;
//                       afterwards every node.ux/uy/fx/fy and element.stress holds a value (solver.rs:476-482,532-533)
// The Rust shim a Magnetite maintainer would add instead is shown in INTEGRATION.md.
#pragma once
#include <array>
#include <cstdint>
#include <optional>
#include <string>
#include <variant>
#include <vector>

#include "magnetite_hip.h"

namespace magnetite {

constexpr std::size_t DOF = MAG_DOF;                    // solver.rs:17
constexpr std::uint64_t MAX_CG_ITER = MAG_MAX_CG_ITER;  // solver.rs:18
constexpr double TARGET_CG_COST = MAG_TARGET_CG_COST;   // solver.rs:19

struct Vertex {  // datatypes.rs:1-5
    double x, y;
};
struct Node {  // datatypes.rs:7-14
    Vertex vertex;
    std::optional<double> ux, uy, fx, fy;
};
struct Element {  // datatypes.rs:16-20
    std::array<std::size_t, 3> nodes;
    std::optional<double> stress;
};
struct ModelMetadata {  // datatypes.rs:22-29
    double youngs_modulus, poisson_ratio, part_thickness;
    float characteristic_length_min = 0.f, characteristic_length_max = 0.f;
};

struct MagnetiteError {  // error.rs:3-22
    enum Kind { Input, Mesher, Solver, PostProcessor } kind;
    std::string message;
    std::string display() const
    {
        static const char *names[] = {"Input", "Mesher", "Solver", "Post Processor"};
        return std::string(names[kind]) + " error: " + message;
    }
};

// Result<(), MagnetiteError>
using Result = std::optional<MagnetiteError>;  // nullopt == Ok(())

namespace solver {

// solver.rs:187-193
inline double compute_element_area(const Element &element, const std::vector<Node> &nodes)
{
    const double xy[6] = {nodes[element.nodes[0]].vertex.x, nodes[element.nodes[0]].vertex.y,
                          nodes[element.nodes[1]].vertex.x, nodes[element.nodes[1]].vertex.y,
                          nodes[element.nodes[2]].vertex.x, nodes[element.nodes[2]].vertex.y};
    const std::int32_t tri[3] = {0, 1, 2};
    return mag_compute_element_area(xy, tri);
}

// solver.rs:543-586.  `options` == nullptr keeps the reference's constants (absolute cost 1e-4, 1e7 iterations).
inline Result run(std::vector<Node> &nodes, std::vector<Element> &elements, const ModelMetadata &model_metadata,
                  const mag_options *options = nullptr, mag_stats *stats_out = nullptr)
{
    auto err = [](std::string m) { return Result(MagnetiteError{MagnetiteError::Solver, std::move(m)}); };
    const std::size_t N = nodes.size(), E = elements.size();
    std::vector<double> xy(2 * N), u_in(2 * N, 0.0), f_in(2 * N, 0.0), u(2 * N), f(2 * N), stress(E);
    std::vector<std::uint8_t> u_known(2 * N, 0);
    std::vector<std::int32_t> conn(3 * E);
    for (std::size_t i = 0; i < N; ++i) {
        xy[2 * i] = nodes[i].vertex.x;
        xy[2 * i + 1] = nodes[i].vertex.y;
        const std::optional<double> *uu[2] = {&nodes[i].ux, &nodes[i].uy}, *ff[2] = {&nodes[i].fx, &nodes[i].fy};
        for (int a = 0; a < 2; ++a) {
            // exactly one of (u, f) per DOF; the reference panics otherwise (solver.rs:431,453,472)
            if (uu[a]->has_value() == ff[a]->has_value())
                return err("node " + std::to_string(i) + ": exactly one of displacement/force must be prescribed per axis");
            if (uu[a]->has_value()) {
                u_known[2 * i + a] = 1;
                u_in[2 * i + a] = **uu[a];
            } else {
                f_in[2 * i + a] = **ff[a];
            }
        }
    }
    for (std::size_t e = 0; e < E; ++e)
        for (int c = 0; c < 3; ++c) {
            if (elements[e].nodes[c] >= N || elements[e].nodes[c] > 0x7fffffffu)
                return err("element " + std::to_string(e) + " references a node outside the mesh");
            conn[3 * e + c] = (std::int32_t)elements[e].nodes[c];
        }
    mag_ctx *ctx = mag_create(options);
    if (!ctx) return err("mag_create failed");
    mag_problem p{};
    p.num_nodes = (std::int64_t)N;
    p.num_elements = (std::int64_t)E;
    p.xy = xy.data();
    p.conn = conn.data();
    p.u_known = u_known.data();
    p.u_in = u_in.data();
    p.f_in = f_in.data();
    p.youngs_modulus = model_metadata.youngs_modulus;
    p.poisson_ratio = model_metadata.poisson_ratio;
    p.part_thickness = model_metadata.part_thickness;
    p.memory = MAG_MEM_HOST;
    mag_result r{};
    r.u_out = u.data();
    r.f_out = f.data();
    r.stress_out = stress.data();
    r.memory = MAG_MEM_HOST;
    const int rc = mag_solve(ctx, &p, &r);
    if (stats_out) mag_get_stats(ctx, stats_out);
    if (rc != MAG_OK) {
        Result e = err(mag_last_error(ctx));
        mag_destroy(ctx);
        return e;
    }
    mag_destroy(ctx);
    for (std::size_t i = 0; i < N; ++i) {  // solver.rs:476-482
        nodes[i].ux = u[2 * i];
        nodes[i].uy = u[2 * i + 1];
        nodes[i].fx = f[2 * i];
        nodes[i].fy = f[2 * i + 1];
    }
    for (std::size_t e = 0; e < E; ++e) elements[e].stress = stress[e];  // solver.rs:532-533
    return std::nullopt;
}

}  // namespace solver
}  // namespace magnetite
