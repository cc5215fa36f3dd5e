"""CPU: SURVEY 8f ranks 1-2 (callers either side of the path): CSV writer and input-JSON reader."""
import json
import os

import numpy as np
import pytest

from magnetite_amd import Element, MagnetiteError, Node, Vertex, meshgen
from magnetite_amd.inputs import load_input_file, parse_boundary_rules, parse_input_metadata, problem_from_input
from magnetite_amd.post_processor import _fmt, csv_output, csv_output_arrays

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# examples/tensile-example/input.json (input data of the reference's plumbing example), reproduced as a fixture value
TENSILE_JSON = {
    "metadata": {"part_thickness": 0.5, "material_elasticity": 69e9, "poisson_ratio": 0.33,
                 "characteristic_length_min": 0, "characteristic_length_max": 0.3},
    "boundary_conditions": {
        "restraint": {"region": {"x_target_min": -12, "x_target_max": -10},
                      "targets": {"ux": 0, "uy": 0, "fx": None, "fy": None}},
        "load": {"region": {"x_target_min": 10, "x_target_max": 12},
                 "targets": {"ux": 3, "uy": None, "fx": None, "fy": 0}},
    },
}


def test_rust_display_formatting():
    assert _fmt(3.0) == "3" and _fmt(0.5) == "0.5" and _fmt(-11.0) == "-11" and _fmt(0.0) == "0"
    assert _fmt(1e21) == "1000000000000000000000" and _fmt(1.5e-7) == "0.00000015"
    assert _fmt(0.1 + 0.2) == "0.30000000000000004" and float(_fmt(6.9e10)) == 6.9e10
    for v in np.random.default_rng(3).standard_normal(200) * 10.0 ** np.random.default_rng(4).integers(-12, 12, 200):
        assert float(_fmt(v)) == v  # shortest round-trip representation


def test_csv_output_layout(tmp_path):
    nodes = [Node(Vertex(0.0, 0.0), ux=0.0, uy=0.0, fx=1.0, fy=2.0), Node(Vertex(1.5, 0.0), ux=3.0, uy=-0.25, fx=0.0, fy=0.0)]
    els = [Element([0, 1, 1], stress=-12.5)]
    n, e = str(tmp_path / "nodes.csv"), str(tmp_path / "elements.csv")
    csv_output(els, nodes, n, e)
    assert open(n).read() == "x,y,ux,uy\n0,0,0,0\n1.5,0,3,-0.25\n"           # post_processor.rs:42-56
    assert open(e).read() == "n0,n1,n2,stress\n0,1,1,-12.5\n"                 # post_processor.rs:58-75
    els[0].stress = None
    with pytest.raises(MagnetiteError, match="Post Processor error|PostProcessor error"):
        csv_output(els, nodes, n, e)
    csv_output_arrays(np.array([0, 0, 1.5, 0.0]), np.array([0, 1, 1]), np.array([0, 0, 3, -0.25]), np.array([-12.5]), n, e)
    assert open(n).read() == "x,y,ux,uy\n0,0,0,0\n1.5,0,3,-0.25\n"
    # plot.py:47-83 reads them back with csv + float()
    rows = [l.split(",") for l in open(n).read().splitlines()[1:]]
    assert [float(v) for v in rows[1]] == [1.5, 0.0, 3.0, -0.25]


def test_input_json_tensile_example(tmp_path):
    path = tmp_path / "input.json"
    path.write_text(json.dumps(TENSILE_JSON))
    doc = load_input_file(str(path))
    md = parse_input_metadata(doc)
    assert (md.youngs_modulus, md.poisson_ratio, md.part_thickness) == (69e9, 0.33, 0.5)
    assert md.characteristic_length_max == pytest.approx(0.3)
    rules = parse_boundary_rules(doc)
    assert [r.name for r in rules] == ["restraint", "load"]                 # file order kept: later rules win
    assert rules[0].y_min == -np.finfo(np.float64).max and rules[0].y_max == np.finfo(np.float64).max
    assert (rules[1].ux, rules[1].uy, rules[1].fx, rules[1].fy) == (3.0, None, None, 0.0)
    # on the committed tensile mesh this reproduces the golden fixture's boundary arrays exactly
    g = np.load(os.path.join(GOLD, "tensile.npz"))
    p = problem_from_input(meshgen.Mesh(g["xy"], g["conn"]), str(path))
    assert np.array_equal(p.u_known, g["u_known"]) and np.array_equal(p.u_in, g["u_in"])
    assert np.array_equal(p.f_in, g["f_in"])


@pytest.mark.parametrize("mutate,msg", [
    (lambda d: d["metadata"].pop("material_elasticity"), "Input json missing material elasticity"),
    (lambda d: d["metadata"].pop("poisson_ratio"), "Input json missing poisson ratio"),
    (lambda d: d["boundary_conditions"]["load"].pop("region"), "Boundary rule load is missing region field"),
    (lambda d: d["boundary_conditions"]["load"].pop("targets"), "Boundary rule load is missing target field"),
    (lambda d: d["boundary_conditions"]["load"]["targets"].update(fx=1.0), "over-constrained in x-axis"),
    (lambda d: d["boundary_conditions"]["load"]["targets"].update(fy=None), "under-constrained in y-axis"),
    (lambda d: d["boundary_conditions"]["load"]["region"].update(x_target_min=20), "x_target_min greater than x_target_max"),
])
def test_input_json_errors(tmp_path, mutate, msg):
    d = json.loads(json.dumps(TENSILE_JSON))
    mutate(d)
    path = tmp_path / "input.json"
    path.write_text(json.dumps(d))
    with pytest.raises(MagnetiteError, match="Input error: .*" + msg):
        doc = load_input_file(str(path))
        parse_input_metadata(doc)
        parse_boundary_rules(doc)
    with pytest.raises(MagnetiteError, match="Unable to open input file"):
        load_input_file(str(tmp_path / "missing.json"))
