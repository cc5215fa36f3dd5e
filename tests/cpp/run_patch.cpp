// Compiled-caller check of include/magnetite_solver.hpp: the reference's run(nodes, elements, metadata)
// signature (solver.rs:543-547) on a uniform-tension patch test with a known exact answer.
#include <cmath>
#include <cstdio>

#include "magnetite_solver.hpp"

using namespace magnetite;

int main()
{
    const int nx = 12, ny = 6;
    const double L = 2.0, H = 1.0, delta = 1e-3;
    std::vector<Node> nodes;
    for (int j = 0; j <= ny; ++j)
        for (int i = 0; i <= nx; ++i) {
            Node n{{L * i / nx, H * j / ny}, std::nullopt, std::nullopt, 0.0, 0.0};  // mesher.rs:615-624 defaults
            if (i == 0) { n.ux = 0.0; n.fx = std::nullopt; if (j == 0) { n.uy = 0.0; n.fy = std::nullopt; } }
            if (i == nx) { n.ux = delta; n.fx = std::nullopt; }
            nodes.push_back(n);
        }
    std::vector<Element> elements;
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            const std::size_t a = j * (nx + 1) + i, b = a + 1, c = a + nx + 1, d = c + 1;
            elements.push_back({{a, b, d}, std::nullopt});
            elements.push_back({{a, d, c}, std::nullopt});
        }
    const ModelMetadata meta{69e9, 0.33, 0.5};
    mag_stats st{};
    if (Result e = solver::run(nodes, elements, meta, nullptr, &st)) {
        std::printf("FAIL %s\n", e->display().c_str());
        return 2;
    }
    double worst = 0.0;
    for (const Node &n : nodes) {
        worst = std::fmax(worst, std::fabs(*n.ux - delta * n.vertex.x / L));
        worst = std::fmax(worst, std::fabs(*n.uy + meta.poisson_ratio * delta * n.vertex.y / L));
    }
    double smin = 1e300, smax = -1e300;
    for (const Element &e : elements) { smin = std::fmin(smin, *e.stress); smax = std::fmax(smax, *e.stress); }
    const double sigma = meta.youngs_modulus * delta / L;
    std::printf("iterations %lld worst |u - exact| %.3e stress [%.6e, %.6e] expected %.6e area0 %.4f\n",
                (long long)st.iterations, worst, smin, smax, sigma, solver::compute_element_area(elements[0], nodes));
    // an over-constrained DOF must come back as a Solver error, not a crash (the reference panics, solver.rs:431)
    nodes[5].fx = 1.0;
    nodes[5].ux = 0.0;
    Result e2 = solver::run(nodes, elements, meta);
    const bool ok = worst <= 1e-9 * delta && std::fabs(smin - sigma) <= 1e-8 * sigma && std::fabs(smax - sigma) <= 1e-8 * sigma &&
                    e2.has_value() && e2->display().rfind("Solver error:", 0) == 0;
    std::printf("%s\n", ok ? "PASS" : "FAIL");
    return ok ? 0 : 1;
}
