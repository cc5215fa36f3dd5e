"""include/magnetite_solver.hpp (the C++ twin of solver::run for compiled callers): it must compile against the
C ABI on any box, and -- on the GPU box -- pass an exact patch test through the reference's own call shape."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "run_patch.cpp")


def compile_to(path):
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", path,
           "-L", os.path.join(ROOT, "magnetite_amd"), "-lmagnetite_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "magnetite_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles_and_links(built, tmp_path):
    compile_to(str(tmp_path / "run_patch"))


@pytest.mark.gpu
def test_cpp_mirror_patch_test_on_gpu(built, tmp_path):
    exe = str(tmp_path / "run_patch")
    compile_to(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
