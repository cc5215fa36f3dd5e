"""Host-side mirror of Magnetite's solver interface over the C ABI.

Same names, argument meaning and error behaviour as the reference:
  datatypes.rs:1-29   Vertex, Node, Element, ModelMetadata
  error.rs:3-22       MagnetiteError, displayed as "<Kind> error: <msg>"
  solver.rs:543-547   run(nodes, elements, model_metadata) mutates in place; afterwards every
                      node.ux/uy/fx/fy and element.stress is set (solver.rs:476-482,532-533)
  solver.rs:17-19     DOF, MAX_CG_ITER, TARGET_CG_COST
  solver.rs:187-193   compute_element_area (pub; the mesher imports it)

The arithmetic happens in libmagnetite_hip.so (hand-written HIP for gfx950); this module only
flattens the AoS-with-Options model into the SoA arrays of include/magnetite_hip.h and back.
The C++ twin for compiled callers is include/magnetite_solver.hpp; the Rust shim is in INTEGRATION.md.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import (MAG_ERR_NOT_CONVERGED, MAG_MEM_HOST, MAG_OK, MAG_OP_CSR, MAG_OP_MATRIX_FREE, MAG_STOP_REL,
                   MAG_STOP_RNORM, MAG_STOP_RNORM_SQ)

DOF = 2
MAX_CG_ITER = int(1e7)
TARGET_CG_COST = 1e-4


class MagnetiteError(Exception):
    """error.rs:3-22"""

    def __init__(self, kind, message, code=None):
        super().__init__(f"{kind} error: {message}")
        self.kind, self.message, self.code = kind, message, code


@dataclass
class Vertex:
    x: float
    y: float


@dataclass
class Node:
    vertex: Vertex
    ux: Optional[float] = None
    uy: Optional[float] = None
    fx: Optional[float] = 0.0  # mesher.rs:615-624 defaults
    fy: Optional[float] = 0.0


@dataclass
class Element:
    nodes: List[int]
    stress: Optional[float] = None


@dataclass
class ModelMetadata:
    youngs_modulus: float
    poisson_ratio: float
    part_thickness: float
    characteristic_length_min: float = 0.0
    characteristic_length_max: float = 0.0


def compute_element_area(element, nodes):
    """solver.rs:187-193 (signed)."""
    xy = np.array([[nodes[i].vertex.x, nodes[i].vertex.y] for i in element.nodes], dtype=np.float64).reshape(-1)
    tri = np.arange(3, dtype=np.int32)
    return _lib.lib().mag_compute_element_area(xy.ctypes.data_as(C.POINTER(C.c_double)),
                                               tri.ctypes.data_as(C.POINTER(C.c_int32)))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Context:
    """Owns one mag_ctx (one GPU, one stream)."""

    def __init__(self, **opts):
        L = _lib.lib()
        o = _lib.Options()
        L.mag_default_options(C.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k}")
            setattr(o, k, v)
        self.options = o
        self._L = L
        self._h = L.mag_create(C.byref(o))
        if not self._h:
            raise MagnetiteError("Solver", "mag_create returned NULL")
        self._keep = None
        self.N = self.E = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.mag_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, allow=()):
        if rc != MAG_OK and rc not in allow:
            raise MagnetiteError("Solver", self._L.mag_last_error(self._h).decode(), rc)
        return rc

    # -- multi-GPU: one process per GPU ------------------------------------------
    def init_rccl_from_torch(self, dist, rank, world):
        """RCCL communicator for the library's own stream; torch.distributed only carries the 128-byte id."""
        import torch
        ident = (C.c_uint8 * _lib.MAG_UNIQUE_ID_BYTES)()
        if rank == 0:
            rc = self._L.mag_comm_get_unique_id(C.cast(ident, C.c_void_p))
            if rc != MAG_OK:
                raise MagnetiteError("Solver", "mag_comm_get_unique_id failed (librccl not loadable?)", rc)
        t = torch.tensor(list(bytes(ident)), dtype=torch.uint8, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        buf = (C.c_uint8 * _lib.MAG_UNIQUE_ID_BYTES).from_buffer_copy(raw)
        self._check(self._L.mag_comm_init_rccl(self._h, C.cast(buf, C.c_void_p), world, rank))

    def comm_info(self):
        """dict(ranks, rank, transport, rccl_ranks) of the context's communicator (mag_comm_query)."""
        info = (C.c_int32 * 4)()
        self._check(self._L.mag_comm_query(self._h, info))
        return dict(ranks=info[0], rank=info[1], transport={0: "none", 1: "rccl", 2: "callback"}[info[2]],
                    rccl_ranks=info[3])

    def set_window(self, shm):
        """Multi-GPU on-chip CG: `shm` is a multiprocessing.shared_memory.SharedMemory (or None to remove the window)
        that EVERY rank of the node has opened under the same name; see mag_comm_set_window in the header."""
        if shm is None:
            self._check(self._L.mag_comm_set_window(self._h, None, 0))
            self._window = None
            return
        addr = C.addressof(C.c_uint8.from_buffer(shm.buf))  # the temporary export ends here: shm.close() stays possible
        self._check(self._L.mag_comm_set_window(self._h, C.c_void_p(addr), shm.size))
        self._window = shm  # keep the mapping alive as long as the library uses it

    def create_inbox(self, nbytes=8 << 20):
        """Multi-GPU on-chip CG, peer-memory form: allocates this rank's inbox in device memory and returns its HIP IPC
        handle (bytes); exchange the handles of all ranks and pass them, in rank order, to open_inboxes()."""
        h = (C.c_uint8 * _lib.MAG_IPC_HANDLE_BYTES)()
        self._check(self._L.mag_comm_inbox_create(self._h, nbytes, C.cast(h, C.c_void_p)))
        return bytes(h)

    def open_inboxes(self, handles):
        raw = b"".join(handles)
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        self._check(self._L.mag_comm_inbox_open(self._h, C.cast(buf, C.c_void_p)))

    def close_inboxes(self):
        self._check(self._L.mag_comm_inbox_create(self._h, 0, None))

    def init_callback(self, fn, rank, world):
        """Test transport: fn(numpy_view) must sum the array over ranks in place (e.g. gloo all_reduce)."""
        import numpy as _np

        def _cb(user, ptr, count):
            try:
                fn(_np.ctypeslib.as_array(ptr, shape=(count,)))
                return 0
            except Exception as exc:  # pragma: no cover - surfaced as MAG_ERR_RCCL
                print("allreduce callback failed:", exc, flush=True)
                return 1

        self._cb = _lib.ALLREDUCE_FN(_cb)  # keep alive
        self._check(self._L.mag_comm_init_callback(self._h, world, rank, self._cb, None))

    # -- problem ---------------------------------------------------------------
    def upload(self, xy, conn, u_known, u_in, f_in, youngs_modulus, poisson_ratio, part_thickness):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
        conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1)
        u_known = np.ascontiguousarray(u_known, dtype=np.uint8).reshape(-1)
        u_in = np.ascontiguousarray(u_in, dtype=np.float64).reshape(-1)
        f_in = np.ascontiguousarray(f_in, dtype=np.float64).reshape(-1)
        N, E = xy.size // 2, conn.size // 3
        if not (u_known.size == u_in.size == f_in.size == 2 * N) or xy.size != 2 * N or conn.size != 3 * E:
            raise MagnetiteError("Solver", "array sizes do not match num_nodes/num_elements")
        p = _lib.Problem(N, E, xy.ctypes.data, conn.ctypes.data, u_known.ctypes.data, u_in.ctypes.data,
                         f_in.ctypes.data, float(youngs_modulus), float(poisson_ratio), float(part_thickness),
                         MAG_MEM_HOST, 0)
        self._check(self._L.mag_upload(self._h, C.byref(p)))
        self.N, self.E = N, E

    def upload_problem(self, prob):
        """prob: magnetite_amd.meshgen.Problem"""
        self.upload(prob.xy_flat, prob.conn_flat, prob.u_known, prob.u_in, prob.f_in, prob.youngs_modulus,
                    prob.poisson_ratio, prob.part_thickness)

    def run(self, allow_not_converged=False):
        """mag_run.  Stopping at the iteration cap is a normal termination, as in the reference (solver.rs:149-176
        returns Ok(best_param)): no error, stats()["converged"] == 0, the best iterate is returned.
        allow_not_converged only tolerates a numerical breakdown (MAG_ERR_NOT_CONVERGED: non-finite residual)."""
        allow = (MAG_ERR_NOT_CONVERGED,) if allow_not_converged else ()
        return self._check(self._L.mag_run(self._h), allow)

    def download(self):
        u, f, s = np.empty(2 * self.N), np.empty(2 * self.N), np.empty(self.E)
        r = _lib.Result(u.ctypes.data, f.ctypes.data, s.ctypes.data, MAG_MEM_HOST, 0)
        self._check(self._L.mag_download(self._h, C.byref(r)))
        return u, f, s

    def stats(self):
        st = _lib.Stats()
        self._L.mag_get_stats(self._h, C.byref(st))
        return st.as_dict()

    def history(self, n):
        h = np.empty(max(n, 1))
        self._check(self._L.mag_get_history(self._h, _p(h, C.c_double), n))
        return h[:n]

    def solve(self, prob, allow_not_converged=False):
        """mag_solve on a meshgen.Problem: dict(u, f, stress, **stats)."""
        self.upload_problem(prob)
        self.run(allow_not_converged)
        u, f, s = self.download()
        out = dict(u=u, f=f, stress=s)
        out.update(self.stats())
        return out

    # -- pieces, for parity tests ----------------------------------------------
    def element_stiffness(self):
        ke = np.empty(36 * self.E)
        self._check(self._L.mag_element_stiffness(self._h, _p(ke, C.c_double)))
        return ke.reshape(self.E, 6, 6)

    def assemble_csr(self):
        nnz = C.c_int64(0)
        self._check(self._L.mag_assemble_csr(self._h, C.byref(nnz), None, None, None))
        rowptr = np.empty(2 * self.N + 1, dtype=np.int32)
        col = np.empty(nnz.value, dtype=np.int32)
        val = np.empty(nnz.value)
        self._check(self._L.mag_assemble_csr(self._h, C.byref(nnz), _p(rowptr, C.c_int32), _p(col, C.c_int32),
                                             _p(val, C.c_double)))
        return rowptr, col, val

    def reduce_system(self):
        nf, nz = C.c_int64(0), C.c_int64(0)
        self._check(self._L.mag_reduce_system(self._h, C.byref(nf), C.byref(nz), None, None, None, None))
        rowptr = np.empty(nf.value + 1, dtype=np.int32)
        col = np.empty(max(nz.value, 1), dtype=np.int32)
        val = np.empty(max(nz.value, 1))
        b = np.empty(nf.value)
        self._check(self._L.mag_reduce_system(self._h, C.byref(nf), C.byref(nz), _p(rowptr, C.c_int32),
                                              _p(col, C.c_int32), _p(val, C.c_double), _p(b, C.c_double)))
        return rowptr, col[:nz.value], val[:nz.value], b

    def apply_operator(self, x, masked=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(2 * self.N)
        self._check(self._L.mag_apply_operator(self._h, _p(x, C.c_double), _p(y, C.c_double), 1 if masked else 0))
        return y

    def time_operator(self, reps=200):
        ms = C.c_double(0.0)
        self._check(self._L.mag_time_operator(self._h, reps, C.byref(ms)))
        return ms.value

    def time_spmv(self, reps=200):
        ms = C.c_double(0.0)
        self._check(self._L.mag_time_spmv(self._h, reps, C.byref(ms)))
        return ms.value


def flatten(nodes, elements):
    """Vec<Node>/Vec<Element> -> SoA (what the Rust shim does before the extern "C" call).

    A DOF with both or neither of (u, f) set cannot be expressed across the ABI; the reference
    panics on it (solver.rs:431), here it is a Solver error.
    """
    N, E = len(nodes), len(elements)
    xy = np.empty(2 * N)
    u_known = np.zeros(2 * N, dtype=np.uint8)
    u_in = np.zeros(2 * N)
    f_in = np.zeros(2 * N)
    for i, nd in enumerate(nodes):
        xy[2 * i], xy[2 * i + 1] = nd.vertex.x, nd.vertex.y
        for a, (u, f) in enumerate(((nd.ux, nd.fx), (nd.uy, nd.fy))):
            if (u is None) == (f is None):
                raise MagnetiteError("Solver", f"node {i} axis {'xy'[a]}: exactly one of displacement/force "
                                               "must be prescribed")
            if u is not None:
                u_known[2 * i + a], u_in[2 * i + a] = 1, u
            else:
                f_in[2 * i + a] = f
    conn = np.empty(3 * E, dtype=np.int32)
    for e, el in enumerate(elements):
        if len(el.nodes) != 3:
            raise MagnetiteError("Solver", f"element {e} does not have 3 nodes")
        conn[3 * e:3 * e + 3] = el.nodes
    return xy, conn, u_known, u_in, f_in


def run(nodes, elements, model_metadata, **options):
    """solver.rs:543-586: updates values on the nodes and elements lists in place."""
    xy, conn, u_known, u_in, f_in = flatten(nodes, elements)
    with Context(**options) as ctx:
        ctx.upload(xy, conn, u_known, u_in, f_in, model_metadata.youngs_modulus, model_metadata.poisson_ratio,
                   model_metadata.part_thickness)
        ctx.run()
        u, f, s = ctx.download()
    for i, nd in enumerate(nodes):
        nd.ux, nd.uy = float(u[2 * i]), float(u[2 * i + 1])
        nd.fx, nd.fy = float(f[2 * i]), float(f[2 * i + 1])
    for e, el in enumerate(elements):
        el.stress = float(s[e])
    return None
