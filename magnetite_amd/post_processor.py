"""the reference's CSV result writer (post_processor.rs:18-83), so results of the HIP path are
consumable by the reference's own scripts/plot.py (plot.py:47-83 reads exactly these two files).

  nodes.csv     header "x,y,ux,uy", one line per node in node order          (post_processor.rs:42-56)
  elements.csv  header "n0,n1,n2,stress", one line per element               (post_processor.rs:58-75)

Rust's `{}` Display for f64 prints the shortest string that round-trips; Python's repr() does the same, except that
Rust never uses exponent notation and prints integral values without a fraction ("3", not "3.0").
"""
from decimal import Decimal

from .solver import MagnetiteError


def _fmt(v):
    """f64 as Rust's Display writes it."""
    v = float(v)
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    s = format(Decimal(repr(v)), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return "-0" if s in ("-0", "-") or (v == 0 and str(v).startswith("-")) else (s or "0")


def csv_output(elements, nodes, nodes_output, elements_output):
    """post_processor.rs:18-83.  nodes / elements: magnetite_amd.solver.Node / Element after run()."""
    try:
        nf = open(nodes_output, "w")
    except OSError as err:
        raise MagnetiteError("Solver", f"Failed to create nodes.csv: {err}")  # the reference reports Solver here too
    try:
        ef = open(elements_output, "w")
    except OSError as err:
        nf.close()
        raise MagnetiteError("Solver", f"Failed to create elements.csv: {err}")
    with nf, ef:
        nf.write("x,y,ux,uy\n")
        for n in nodes:
            if n.ux is None or n.uy is None:  # the reference unwrap()s (post_processor.rs:50-51)
                raise MagnetiteError("PostProcessor", "node without displacement: run the solver first")
            nf.write(f"{_fmt(n.vertex.x)},{_fmt(n.vertex.y)},{_fmt(n.ux)},{_fmt(n.uy)}\n")
        ef.write("n0,n1,n2,stress\n")
        for e in elements:
            if e.stress is None:
                raise MagnetiteError("PostProcessor", "element without stress: run the solver first")
            ef.write(f"{e.nodes[0]},{e.nodes[1]},{e.nodes[2]},{_fmt(e.stress)}\n")
    print(f"info: wrote output to {nodes_output} and {elements_output}")


def csv_output_arrays(xy, conn, u, stress, nodes_output, elements_output):
    """Same files straight from the flat arrays of the C ABI (no per-node Python objects)."""
    with open(nodes_output, "w") as nf:
        nf.write("x,y,ux,uy\n")
        for i in range(len(xy) // 2):
            nf.write(f"{_fmt(xy[2 * i])},{_fmt(xy[2 * i + 1])},{_fmt(u[2 * i])},{_fmt(u[2 * i + 1])}\n")
    with open(elements_output, "w") as ef:
        ef.write("n0,n1,n2,stress\n")
        for e in range(len(conn) // 3):
            ef.write(f"{conn[3 * e]},{conn[3 * e + 1]},{conn[3 * e + 2]},{_fmt(stress[e])}\n")
