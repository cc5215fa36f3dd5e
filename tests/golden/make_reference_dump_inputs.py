"""Writes the inputs of tests/golden/dump_reference.rs -- the tensile mesh and boundary values of tests/golden/tensile.npz as
two CSV files a Rust test can read without extra crates (empty field = None, i.e. the other of (u, f) is prescribed).
Floats are written with repr(): they parse back to the same bits.

    python tests/golden/make_reference_dump_inputs.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
g = np.load(os.path.join(HERE, "tensile.npz"))
xy, conn, known, u_in, f_in = g["xy"], g["conn"], g["u_known"], g["u_in"], g["f_in"]
with open(os.path.join(HERE, "tensile_nodes.csv"), "w") as f:
    f.write("x,y,ux,uy,fx,fy\n")
    for i in range(xy.shape[0]):
        c = lambda d, arr, want: repr(float(arr[2 * i + d])) if known[2 * i + d] == want else ""
        f.write(",".join([repr(float(xy[i, 0])), repr(float(xy[i, 1])), c(0, u_in, 1), c(1, u_in, 1), c(0, f_in, 0),
                          c(1, f_in, 0)]) + "\n")
with open(os.path.join(HERE, "tensile_elements.csv"), "w") as f:
    f.write("n0,n1,n2\n")
    for e in conn:
        f.write(",".join(str(int(v)) for v in e) + "\n")
print("wrote tensile_nodes.csv, tensile_elements.csv:", xy.shape[0], "nodes,", conn.shape[0], "elements")
