"""CPU: the C-ABI shared library loads, exports every symbol include/magnetite_hip.h declares, the ctypes
binding matches the header's struct layouts, and -- without a GPU -- compute entry points fail loudly (there is
no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle
from magnetite_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "magnetite_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mag_[a-z_0-9]+)\s*\(", src)) - {"mag_allreduce_fn"})


def test_header_binding_and_library_agree_on_symbols(built):
    names = header_functions()
    assert names == sorted(_lib.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.SO_PATH], text=True)
    exported = set(re.findall(r" T (mag_[a-z_0-9]+)", out))
    assert set(names) <= exported, set(names) - exported
    L = _lib.lib()
    for n in names:
        getattr(L, n)


def test_struct_layouts_match_header(built, tmp_path):
    structs = {"mag_options": _lib.Options, "mag_problem": _lib.Problem, "mag_result": _lib.Result,
               "mag_stats": _lib.Stats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "magnetite_hip.h"', "int main(void){"]
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in cls._fields_:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"


def test_version_defaults_and_constants(built):
    L = _lib.lib()
    assert L.mag_version() == 4
    o = _lib.Options()
    L.mag_default_options(C.byref(o))
    # solver.rs:17-19
    assert o.tol == 1e-4 and o.max_iter == int(1e7) and o.stop_mode == _lib.MAG_STOP_RNORM
    assert o.cg_operator == _lib.MAG_OP_MATRIX_FREE and o.assemble_csr == 1 and o.tile_nodes == 0
    hdr = open(HEADER).read()
    assert "#define MAG_DOF 2" in hdr and "#define MAG_MAX_CG_ITER 10000000LL" in hdr
    assert "#define MAG_TARGET_CG_COST 1e-4" in hdr


def test_compute_element_area_is_the_reference_formula(built):
    """solver.rs:187-193 stays a host-side pub function (the mesher imports it, mesher.rs:9,523)."""
    L = _lib.lib()
    rng = np.random.default_rng(7)
    for _ in range(50):
        xy = rng.uniform(-10, 10, size=8)
        tri = rng.permutation(4)[:3].astype(np.int32)
        a = L.mag_compute_element_area(xy.ctypes.data_as(C.POINTER(C.c_double)), tri.ctypes.data_as(C.POINTER(C.c_int32)))
        assert a == oracle.element_area(xy, tri)


def test_pub_matrix_functions_match_the_oracle_bitwise(built):
    """solver.rs:204-230 and :240-250 are pub in the reference; the library keeps host-side twins."""
    L = _lib.lib()
    rng = np.random.default_rng(11)
    for _ in range(50):
        xy = rng.uniform(-10, 10, size=8)
        tri = rng.permutation(4)[:3].astype(np.int32)
        area = oracle.element_area(xy, tri)
        B = np.empty(18)
        L.mag_compute_strain_displacement_matrix(xy.ctypes.data_as(C.POINTER(C.c_double)),
                                                 tri.ctypes.data_as(C.POINTER(C.c_int32)), area,
                                                 B.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(B.reshape(3, 6), oracle.strain_displacement(xy, tri, area))
        nu, E = rng.uniform(0.0, 0.49), rng.uniform(1.0, 1e11)
        D = np.empty(9)
        L.mag_compute_stress_strain_matrix(nu, E, D.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(D.reshape(3, 3), oracle.stress_strain(nu, E))


def test_no_cpu_fallback(built):
    """On a box without a HIP device the product path must fail loudly, never compute on the CPU."""
    probe = subprocess.run([sys.executable, "-c", "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 3)"],
                           capture_output=True)
    if probe.returncode == 0:
        pytest.skip("a GPU is present")
    from magnetite_amd import Context, MagnetiteError, meshgen
    p = meshgen.config_fixed_left_pull_right(meshgen.plate(4))
    with Context(device=0) as ctx:
        with pytest.raises(MagnetiteError) as ei:
            ctx.upload_problem(p)
        assert ei.value.code == _lib.MAG_ERR_HIP
        assert "no CPU path" in str(ei.value)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under magnetite_amd/ or include/ may reference it."""
    bad = []
    for base in ("magnetite_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(import|from)\s+oracle\b|liboracle|orc_[a-z_]+\(", txt, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
