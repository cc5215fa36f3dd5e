"""tools/magnetite_gpu.cpp: the compiled caller with main.rs's stage order.  CPU: it builds and its parsers
(input JSON, boundary rules, MSH-4 + check_ccw) agree with the Python mirrors through --dry-run.  GPU: the full
pipeline writes the reference's CSV files with the same numbers as the Python path."""
import json
import os
import subprocess

import numpy as np
import pytest

from magnetite_amd import meshgen
from magnetite_amd.inputs import problem_from_input
from magnetite_amd.msh import parse_mesh, write_msh
from magnetite_amd.post_processor import _fmt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
from test_io_rows import TENSILE_JSON  # noqa: E402


@pytest.fixture(scope="module")
def exe(built, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("tool") / "magnetite_gpu")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "magnetite_gpu.cpp"), "-o", out,
                           "-L", os.path.join(ROOT, "magnetite_amd"), "-lmagnetite_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "magnetite_amd"), "-Wl,-rpath,/opt/rocm/lib",
                           "-L/opt/rocm/lib", "-lamdhip64"])
    return out


@pytest.fixture(scope="module")
def tensile_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("tensile")
    g = np.load(os.path.join(GOLD, "tensile.npz"))
    # the fixture mesh already went through check_ccw (all clockwise); undo it so that the tool's own check_ccw acts
    mesh = meshgen.Mesh(g["xy"], np.ascontiguousarray(g["conn"][:, ::-1]))
    write_msh(mesh, str(d / "geom.msh"))
    (d / "input.json").write_text(json.dumps(TENSILE_JSON))
    return d, g


def test_dry_run_matches_python_mirrors(exe, tensile_files):
    d, g = tensile_files
    r = subprocess.run([exe, str(d / "input.json"), str(d / "geom.msh"), "--dry-run"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("dry-run:")][0].split()
    kv = dict(zip(line[1::2], line[2::2]))
    mesh = parse_mesh(str(d / "geom.msh"))
    p = problem_from_input(mesh, str(d / "input.json"))
    assert int(kv["nodes"]) == mesh.num_nodes and int(kv["elements"]) == mesh.num_elements
    assert int(kv["prescribed_u"]) == int(p.u_known.sum()) and int(kv["prescribed_f"]) == int((p.u_known == 0).sum())
    assert float(kv["sum_u"]) == float(p.u_in.sum()) and float(kv["sum_f"]) == float(p.f_in.sum())
    assert (kv["E"], kv["nu"], kv["t"]) == (_fmt(69e9), "0.33", "0.5")
    assert kv["first"] == ",".join(str(int(v)) for v in mesh.conn[0])  # check_ccw applied identically
    assert np.array_equal(p.u_known, g["u_known"]) and np.array_equal(mesh.conn, g["conn"])
    assert "info: loaded 2 boundary rules from input file" in r.stdout


def test_input_errors_exit_code_and_message(exe, tensile_files, tmp_path):
    d, _ = tensile_files
    bad = json.loads(json.dumps(TENSILE_JSON))
    bad["boundary_conditions"]["load"]["targets"]["fx"] = 1.0
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    r = subprocess.run([exe, str(tmp_path / "bad.json"), str(d / "geom.msh"), "--dry-run"], capture_output=True, text=True)
    assert r.returncode == 1  # main.rs:44-50
    assert "Received error: Input error: Boundary 'load' is over-constrained in x-axis" in r.stderr
    r = subprocess.run([exe, str(tmp_path / "nope.json"), str(d / "geom.msh"), "--dry-run"], capture_output=True, text=True)
    assert r.returncode == 1 and "Input error: Unable to open input file" in r.stderr
    # load_input_file's key checks, messages verbatim and in the reference's order (mesher.rs:733-755)
    for mutate, msg in ((lambda b: b.pop("metadata"), "Input json missing metadata field"),
                        (lambda b: b.pop("boundary_conditions"),
                         "Input json missing boundary_conditions field in metadata section"),
                        (lambda b: b["metadata"].pop("part_thickness"),
                         "Input json missing part_thickness field in metadata section"),
                        (lambda b: b["metadata"].pop("material_elasticity"),
                         "Input json missing material_elasticity field in metadata section"),
                        (lambda b: b["metadata"].pop("poisson_ratio"),
                         "Input json missing poisson_ratio field in metadata section")):
        bad = json.loads(json.dumps(TENSILE_JSON))
        mutate(bad)
        (tmp_path / "bad.json").write_text(json.dumps(bad))
        r = subprocess.run([exe, str(tmp_path / "bad.json"), str(d / "geom.msh"), "--dry-run"], capture_output=True,
                           text=True)
        assert r.returncode == 1 and r.stderr.strip().endswith("Received error: Input error: " + msg), r.stderr
    (tmp_path / "bad.json").write_text('{"metadata": ')
    r = subprocess.run([exe, str(tmp_path / "bad.json"), str(d / "geom.msh"), "--dry-run"], capture_output=True, text=True)
    assert r.returncode == 1 and "Received error: Input error: Error in input file json: " in r.stderr


@pytest.mark.gpu
def test_full_pipeline_writes_reference_csv(exe, tensile_files, tmp_path):
    d, g = tensile_files
    n, e = str(tmp_path / "nodes.csv"), str(tmp_path / "elements.csv")
    r = subprocess.run([exe, str(d / "input.json"), str(d / "geom.msh"), "--nodes", n, "--elements", e],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "info: finished conjugate gradient approximation in" in r.stdout
    nodes = np.loadtxt(n, delimiter=",", skiprows=1)
    els = np.loadtxt(e, delimiter=",", skiprows=1)
    assert open(n).readline() == "x,y,ux,uy\n" and open(e).readline() == "n0,n1,n2,stress\n"
    assert np.array_equal(nodes[:, :2], g["xy"]) and np.array_equal(els[:, :3].astype(int), g["conn"])
    u = nodes[:, 2:].reshape(-1)
    assert np.linalg.norm(u - g["u"]) / np.linalg.norm(g["u"]) <= 1e-8
    stable = np.abs(g["stress"]) > 1e-3 * np.abs(g["stress"]).max()
    assert np.allclose(els[:, 3][stable], g["stress"][stable], rtol=1e-6)
