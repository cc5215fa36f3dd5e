"""ctypes binding of include/magnetite_hip.h (libmagnetite_hip.so).

Plumbing only: structs and prototypes mirror the header one to one.  There is
no fallback -- if the shared library is missing or no HIP device is usable the
calls raise MagnetiteError with the library's message.
"""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
# MAG_LIB_PATH: load another build of the same library (A/B of compiler flags or kernel variants in one GPU session)
SO_PATH = os.environ.get("MAG_LIB_PATH") or os.path.join(_DIR, "libmagnetite_hip.so")
CSRC = os.path.join(_DIR, "csrc")

MAG_OK, MAG_ERR_BAD_ARGS, MAG_ERR_BC_MISMATCH, MAG_ERR_NOT_CONVERGED = 0, 1, 2, 3
MAG_ERR_HIP, MAG_ERR_RCCL, MAG_ERR_TOO_LARGE, MAG_ERR_STATE = 4, 5, 6, 7
MAG_STOP_RNORM, MAG_STOP_RNORM_SQ, MAG_STOP_REL = 0, 1, 2
MAG_OP_MATRIX_FREE, MAG_OP_CSR = 0, 1
MAG_TERM_NONE, MAG_TERM_TARGET_COST, MAG_TERM_MAX_ITERS, MAG_TERM_BREAKDOWN = 0, 1, 2, 3
MAG_MEM_HOST, MAG_MEM_DEVICE = 0, 1
MAG_UNIQUE_ID_BYTES = 128
MAG_IPC_HANDLE_BYTES = 64

# every symbol include/magnetite_hip.h declares (tests/test_abi.py checks the .so exports them all)
SYMBOLS = [
    "mag_version", "mag_default_options", "mag_create", "mag_destroy", "mag_last_error", "mag_solve",
    "mag_upload", "mag_run", "mag_download", "mag_get_stats", "mag_get_history", "mag_compute_element_area",
    "mag_compute_strain_displacement_matrix", "mag_compute_stress_strain_matrix",
    "mag_element_stiffness", "mag_assemble_csr", "mag_reduce_system", "mag_apply_operator", "mag_time_operator", "mag_time_spmv",
    "mag_comm_get_unique_id", "mag_comm_init_rccl", "mag_comm_query", "mag_comm_init_callback", "mag_comm_set_window", "mag_comm_inbox_create", "mag_comm_inbox_open",
]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("stop_mode", C.c_int32), ("tol", C.c_double), ("max_iter", C.c_int64),
                ("cg_operator", C.c_int32), ("assemble_csr", C.c_int32), ("check_every", C.c_int32),
                ("use_graph", C.c_int32), ("tile_nodes", C.c_int32), ("history_len", C.c_int32),
                ("verbose", C.c_int32), ("op_variant", C.c_int32), ("cg_variant", C.c_int32), ("precision", C.c_int32),
                ("preconditioner", C.c_int32), ("reserved", C.c_int32)]


class Problem(C.Structure):
    _fields_ = [("num_nodes", C.c_int64), ("num_elements", C.c_int64), ("xy", C.c_void_p), ("conn", C.c_void_p),
                ("u_known", C.c_void_p), ("u_in", C.c_void_p), ("f_in", C.c_void_p),
                ("youngs_modulus", C.c_double), ("poisson_ratio", C.c_double), ("part_thickness", C.c_double),
                ("memory", C.c_int32), ("reserved", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("u_out", C.c_void_p), ("f_out", C.c_void_p), ("stress_out", C.c_void_p),
                ("memory", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("final_cost", C.c_double), ("rhs_norm", C.c_double),
                ("converged", C.c_int32), ("breakdown", C.c_int32), ("n_free", C.c_int64), ("nnz", C.c_int64),
                ("num_tiles", C.c_int64), ("ell_entries", C.c_int64), ("halo_nodes", C.c_int64),
                ("max_tile_halo", C.c_int32), ("lds_operator", C.c_int32), ("cg_kernel", C.c_int32),
                ("exchange", C.c_int32), ("ms_order", C.c_double),
                ("ms_csr_symbolic", C.c_double), ("ms_element", C.c_double), ("ms_assemble", C.c_double),
                ("ms_bc", C.c_double), ("ms_cg", C.c_double), ("ms_post", C.c_double), ("ms_total", C.c_double),
                ("best_iteration", C.c_int64), ("termination", C.c_int32), ("persist_timeout", C.c_int32),
                ("exchange_timeout", C.c_int32), ("best_param_mismatch", C.c_int32),
                ("edge_blocks", C.c_int32), ("tiles_per_workgroup", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


HASHED_SOURCES = ("persist.hip", "cg.hip", "cg_device.h", "exact.hip", "symbolic.hip", "kernels.h")


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the kernel sources as they are in the tree now -- the digest csrc/Makefile
    writes next to the library when it links it."""
    import hashlib
    h = hashlib.sha256()
    for name in HASHED_SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_source_hash():
    """The digest recorded when the loaded library was linked (None for a library built by other means)."""
    path = os.path.splitext(SO_PATH)[0] + ".srchash"
    try:
        with open(path) as f:
            return f.read().strip() or None
    except OSError:
        return None


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the shared library (cross-compiles without a GPU), and of its diagnostic twin with
    the in-kernel phase stamps (libmagnetite_hip_stamps.so: scripts/persist_phases*.py load it; the product never does)."""
    cmd = ["make", "-C", CSRC, "-j6", "all", "stamps"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise OSError(f"{SO_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)")
    L = C.CDLL(SO_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.mag_version.restype = C.c_int
    L.mag_default_options.argtypes = [C.POINTER(Options)]
    L.mag_default_options.restype = None
    L.mag_create.argtypes = [C.POINTER(Options)]
    L.mag_create.restype = vp
    L.mag_destroy.argtypes = [vp]
    L.mag_destroy.restype = None
    L.mag_last_error.argtypes = [vp]
    L.mag_last_error.restype = C.c_char_p
    L.mag_solve.argtypes = [vp, C.POINTER(Problem), C.POINTER(Result)]
    L.mag_upload.argtypes = [vp, C.POINTER(Problem)]
    L.mag_run.argtypes = [vp]
    L.mag_download.argtypes = [vp, C.POINTER(Result)]
    L.mag_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.mag_get_history.argtypes = [vp, dp, C.c_int64]
    L.mag_compute_element_area.argtypes = [dp, ip]
    L.mag_compute_element_area.restype = C.c_double
    L.mag_compute_strain_displacement_matrix.argtypes = [dp, ip, C.c_double, dp]
    L.mag_compute_strain_displacement_matrix.restype = None
    L.mag_compute_stress_strain_matrix.argtypes = [C.c_double, C.c_double, dp]
    L.mag_compute_stress_strain_matrix.restype = None
    L.mag_element_stiffness.argtypes = [vp, dp]
    L.mag_assemble_csr.argtypes = [vp, C.POINTER(C.c_int64), ip, ip, dp]
    L.mag_reduce_system.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), ip, ip, dp, dp]
    L.mag_apply_operator.argtypes = [vp, dp, dp, C.c_int32]
    L.mag_time_operator.argtypes = [vp, C.c_int32, dp]
    L.mag_time_spmv.argtypes = [vp, C.c_int32, dp]
    L.mag_comm_get_unique_id.argtypes = [vp]
    L.mag_comm_init_rccl.argtypes = [vp, vp, C.c_int32, C.c_int32]
    L.mag_comm_query.argtypes = [vp, ip]
    L.mag_comm_init_callback.argtypes = [vp, C.c_int32, C.c_int32, ALLREDUCE_FN, vp]
    L.mag_comm_set_window.argtypes = [vp, vp, C.c_uint64]
    L.mag_comm_inbox_create.argtypes = [vp, C.c_uint64, vp]
    L.mag_comm_inbox_open.argtypes = [vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("mag_version",):
            fn.restype = C.c_int
    _lib = L
    return L
