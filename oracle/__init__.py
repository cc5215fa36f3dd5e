"""ctypes front-end of the CPU oracle (oracle/magnetite_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under magnetite_amd/ may import this.
PARITY UNPINNED (see the C file's header): the reference ships no tests or
fixtures and cannot be built here (Rust crate, no cargo/rustc).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liboracle.so")

STOP_RNORM, STOP_RNORM_SQ, STOP_REL = 0, 1, 2
# solver.rs:17-19
DOF = 2
MAX_CG_ITER = int(1e7)
TARGET_CG_COST = 1e-4


def build(force=False):
    src = os.path.join(_DIR, "magnetite_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("final_cost", C.c_double),
                ("n_free", C.c_int64), ("nnz_ff", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.orc_element_area.restype = C.c_double
        L.orc_element_area.argtypes = [dp, ip]
        L.orc_strain_displacement.argtypes = [dp, ip, C.c_double, dp]
        L.orc_stress_strain.argtypes = [C.c_double, C.c_double, dp]
        L.orc_element_stiffness.argtypes = [dp, ip, C.c_double, C.c_double, C.c_double, dp]
        L.orc_element_stiffness_all.argtypes = [C.c_int64, dp, ip, C.c_double, C.c_double, C.c_double, dp]
        L.orc_assemble_dense.argtypes = [C.c_int64, C.c_int64, ip, dp, dp]
        L.orc_partition_dense.argtypes = [C.c_int64, dp, bp, dp, dp, dp, dp]
        L.orc_sparsify_dense.restype = C.c_void_p
        L.orc_sparsify_dense.argtypes = [C.c_int64, dp]
        L.orc_csr_free.argtypes = [C.c_void_p]
        L.orc_csr_n.restype = C.c_int64
        L.orc_csr_n.argtypes = [C.c_void_p]
        L.orc_csr_nnz.restype = C.c_int64
        L.orc_csr_nnz.argtypes = [C.c_void_p]
        L.orc_csr_copy.argtypes = [C.c_void_p, C.POINTER(C.c_int64), ip, dp]
        L.orc_spmv.argtypes = [C.c_void_p, dp, dp]
        L.orc_cg.restype = C.c_int64
        L.orc_cg.argtypes = [C.c_void_p, dp, C.c_int, C.c_double, C.c_int64, dp, dp, dp, C.c_int64]
        L.orc_cg_parallel.restype = C.c_int64
        L.orc_cg_parallel.argtypes = [C.c_void_p, dp, C.c_int, C.c_double, C.c_int64, dp, dp]
        L.orc_stress.argtypes = [C.c_int64, dp, ip, dp, C.c_double, C.c_double, dp]
        L.orc_assemble_sparse.restype = C.c_void_p
        L.orc_assemble_sparse.argtypes = [C.c_int64, C.c_int64, dp, ip, C.c_double, C.c_double, C.c_double]
        L.orc_reduce_system.restype = C.c_void_p
        L.orc_reduce_system.argtypes = [C.c_void_p, bp, dp, dp, dp]
        L.orc_assemble_sparse_omp.restype = C.c_void_p
        L.orc_assemble_sparse_omp.argtypes = L.orc_assemble_sparse.argtypes
        L.orc_reduce_system_omp.restype = C.c_void_p
        L.orc_reduce_system_omp.argtypes = L.orc_reduce_system.argtypes
        run_args = [C.c_int64, C.c_int64, dp, ip, bp, dp, dp, C.c_double, C.c_double, C.c_double,
                    C.c_int, C.c_double, C.c_int64, dp, dp, dp, C.POINTER(Stats), dp, C.c_int64]
        L.orc_run_dense.argtypes = run_args
        L.orc_run_sparse.argtypes = run_args
        L.orc_run_sparse_pcg.argtypes = run_args + [C.c_int]
        L.orc_block_jacobi.argtypes = [C.c_void_p, C.c_int64, bp, C.c_int, C.POINTER(C.c_float)]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _b(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _prep(xy, conn):
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1)
    return xy, conn


def element_area(xy, tri):
    xy, tri = _prep(xy, tri)
    return lib().orc_element_area(_d(xy), _i(tri))


def strain_displacement(xy, tri, area):
    xy, tri = _prep(xy, tri)
    B = np.empty(18)
    lib().orc_strain_displacement(_d(xy), _i(tri), float(area), _d(B))
    return B.reshape(3, 6)


def stress_strain(nu, youngs):
    D = np.empty(9)
    lib().orc_stress_strain(float(nu), float(youngs), _d(D))
    return D.reshape(3, 3)


def element_stiffness_all(xy, conn, nu, youngs, thickness):
    xy, conn = _prep(xy, conn)
    E = conn.size // 3
    Ke = np.empty(36 * max(E, 1))
    lib().orc_element_stiffness_all(E, _d(xy), _i(conn), float(nu), float(youngs), float(thickness), _d(Ke))
    return Ke[:36 * E].reshape(E, 6, 6)


class Csr:
    """Owned copy of an orc_csr (rowptr int64, col int32 ascending per row, val f64)."""

    def __init__(self, handle):
        L = lib()
        self.n = L.orc_csr_n(handle)
        self.nnz = L.orc_csr_nnz(handle)
        self.rowptr = np.empty(self.n + 1, dtype=np.int64)
        self.col = np.empty(max(self.nnz, 1), dtype=np.int32)
        self.val = np.empty(max(self.nnz, 1), dtype=np.float64)
        L.orc_csr_copy(handle, self.rowptr.ctypes.data_as(C.POINTER(C.c_int64)), _i(self.col), _d(self.val))
        self.col = self.col[:self.nnz]
        self.val = self.val[:self.nnz]
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_csr_free(self._h)
            self._h = None

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(max(self.n, 1))
        lib().orc_spmv(self._h, _d(x), _d(y))
        return y[:self.n]

    def toarray(self):
        A = np.zeros((self.n, self.n))
        for r in range(self.n):
            s, e = self.rowptr[r], self.rowptr[r + 1]
            A[r, self.col[s:e]] = self.val[s:e]
        return A


def assemble_dense(N, conn, Ke):
    _, conn = _prep(np.zeros(1), conn)
    Ke = np.ascontiguousarray(Ke, dtype=np.float64).reshape(-1)
    K = np.empty((2 * N, 2 * N))
    lib().orc_assemble_dense(N, conn.size // 3, _i(conn), _d(Ke), _d(K))
    return K


def partition_dense(K, u_known, u_in, f_in):
    n = K.shape[0]
    u_known = np.ascontiguousarray(u_known, dtype=np.uint8)
    nf = int(n - u_known.sum())
    Kff = np.zeros((nf, nf))
    b = np.zeros(max(nf, 1))
    K = np.ascontiguousarray(K)
    lib().orc_partition_dense(n, _d(K), _b(u_known), _d(np.ascontiguousarray(u_in, dtype=np.float64)),
                              _d(np.ascontiguousarray(f_in, dtype=np.float64)), _d(Kff), _d(b))
    return Kff, b[:nf]


def sparsify_dense(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    return Csr(lib().orc_sparsify_dense(A.shape[0], _d(A)))


def set_threads(threads):
    """OpenMP thread count of the all-cores variants (None keeps OMP_NUM_THREADS / the runtime default)."""
    if threads:
        os.environ["OMP_NUM_THREADS"] = str(int(threads))
        try:
            C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
        except OSError:
            pass


def assemble_sparse(xy, conn, nu, youngs, thickness, threads=None):
    """threads=None: the serial restatement; threads=k: the OpenMP variant (rows are independent), same bits."""
    xy, conn = _prep(xy, conn)
    set_threads(threads)
    fn = lib().orc_assemble_sparse_omp if threads else lib().orc_assemble_sparse
    return Csr(fn(xy.size // 2, conn.size // 3, _d(xy), _i(conn), float(nu), float(youngs), float(thickness)))


def reduce_system(K, u_known, u_in, f_in, threads=None):
    u_known = np.ascontiguousarray(u_known, dtype=np.uint8)
    nf = int(K.n - u_known.sum())
    b = np.zeros(max(nf, 1))
    set_threads(threads)
    fn = lib().orc_reduce_system_omp if threads else lib().orc_reduce_system
    h = fn(K._h, _b(u_known), _d(np.ascontiguousarray(u_in, dtype=np.float64)),
           _d(np.ascontiguousarray(f_in, dtype=np.float64)), _d(b))
    return Csr(h), b[:nf]


def cg(A, b, stop_mode=STOP_RNORM, tol=TARGET_CG_COST, max_iter=MAX_CG_ITER, hist_len=0):
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(max(A.n, 1))
    cost = C.c_double(0.0)
    hist = np.zeros(max(hist_len, 1))
    it = lib().orc_cg(A._h, _d(b), stop_mode, float(tol), int(max_iter), _d(x), C.byref(cost), _d(hist), hist_len)
    return x[:A.n], int(it), cost.value, hist[:min(hist_len, it)]


def cg_parallel(A, b, stop_mode=STOP_RNORM, tol=TARGET_CG_COST, max_iter=MAX_CG_ITER, threads=None):
    """All-cores variant (OpenMP).  threads=None keeps OMP_NUM_THREADS / the runtime default."""
    set_threads(threads)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(max(A.n, 1))
    cost = C.c_double(0.0)
    it = lib().orc_cg_parallel(A._h, _d(b), stop_mode, float(tol), int(max_iter), _d(x), C.byref(cost))
    return x[:A.n], int(it), cost.value


def stress(xy, conn, u, nu, youngs):
    xy, conn = _prep(xy, conn)
    E = conn.size // 3
    s = np.empty(max(E, 1))
    lib().orc_stress(E, _d(xy), _i(conn), _d(np.ascontiguousarray(u, dtype=np.float64)), float(nu), float(youngs), _d(s))
    return s[:E]


def block_jacobi(K, N, u_known, kind):
    """(i00, i01, i11) per node, fp32: the opt-in preconditioner's inverse node blocks (no reference counterpart)."""
    out = np.zeros(3 * N + 1, dtype=np.float32)
    lib().orc_block_jacobi(K._h, N, _b(np.ascontiguousarray(u_known, dtype=np.uint8)), int(kind),
                           out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:3 * N].reshape(N, 3)


def run(xy, conn, u_known, u_in, f_in, youngs, nu, thickness, path="sparse", stop_mode=STOP_RNORM,
        tol=TARGET_CG_COST, max_iter=MAX_CG_ITER, hist_len=0, precond=0):
    """solver.rs:543-586 run(): returns dict(u, f, stress, iterations, final_cost, n_free, nnz_ff, history).
    precond != 0 (sparse path only): Jacobi (1) / block-Jacobi (2) preconditioned CG -- an addition, not the reference."""
    xy, conn = _prep(xy, conn)
    N, E = xy.size // 2, conn.size // 3
    u_known = np.ascontiguousarray(u_known, dtype=np.uint8).reshape(-1)
    u_in = np.ascontiguousarray(u_in, dtype=np.float64).reshape(-1)
    f_in = np.ascontiguousarray(f_in, dtype=np.float64).reshape(-1)
    assert u_known.size == 2 * N and u_in.size == 2 * N and f_in.size == 2 * N
    u = np.empty(2 * N)
    f = np.empty(2 * N)
    s = np.empty(max(E, 1))
    st = Stats()
    hist = np.zeros(max(hist_len, 1))
    args = (N, E, _d(xy), _i(conn), _b(u_known), _d(u_in), _d(f_in), float(youngs), float(nu),
            float(thickness), stop_mode, float(tol), int(max_iter), _d(u), _d(f), _d(s), C.byref(st),
            _d(hist), hist_len)
    if precond:
        assert path == "sparse"
        rc = lib().orc_run_sparse_pcg(*args, int(precond))
    else:
        rc = (lib().orc_run_dense if path == "dense" else lib().orc_run_sparse)(*args)
    if rc != 0:
        raise RuntimeError(f"oracle run failed rc={rc}")
    return dict(u=u, f=f, stress=s[:E], iterations=int(st.iterations), final_cost=st.final_cost,
                n_free=int(st.n_free), nnz_ff=int(st.nnz_ff), history=hist[:min(hist_len, int(st.iterations))])
