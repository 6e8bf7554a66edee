"""ctypes bindings of libgcge_host.so / libgcge_hip.so (C ABI: include/*.h).

There is NO CPU fallback here: HipBackend() raises if the HIP library or a GPU is
missing.  The CPU oracle lives under oracle/ and is imported by tests only.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIBDIR = os.path.join(_HERE, "lib")


class CSR(C.Structure):
    """GCGE_CSR (include/gcge_problems.h)."""
    _fields_ = [("nrows", C.c_int), ("ncols", C.c_int), ("row_begin", C.c_int),
                ("nnz", C.c_int64), ("rowptr", C.POINTER(C.c_int)),
                ("colidx", C.POINTER(C.c_int)), ("val", C.POINTER(C.c_double))]


class Timing(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("initX", "checkconv", "compP", "compRR", "rr_matW",
                                          "dsyevx", "compRV", "compW", "linsol", "compX", "total")]


class RunResult(C.Structure):
    _fields_ = [("nevConv", C.c_int), ("numIter", C.c_int), ("nevMax", C.c_int),
                ("block_size", C.c_int), ("nevInit", C.c_int), ("seconds", C.c_double),
                ("timing", Timing)]


def build_libs(hip=True, verbose=False):
    """make -C gcge_amd/csrc [host|all]  (hipcc cross-compiles gfx950 without a GPU)."""
    target = "all" if hip else "host"
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), target],
                       capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building gcge libraries failed:\n" + (r.stdout or "") + (r.stderr or ""))


_host = None
_hip = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(_LIBDIR, "libgcge_host.so")
        if not os.path.exists(path):
            build_libs(hip=False)
        _host = C.CDLL(path, mode=C.RTLD_GLOBAL)
        _host.gcge_uniform.restype = C.c_double
        _host.gcge_uniform.argtypes = [C.c_uint64, C.c_uint64]
        _host.GCGE_RunGCG.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.POINTER(C.c_char_p), C.c_void_p,
                                      C.POINTER(C.c_double), C.c_void_p, C.POINTER(RunResult)]
    return _host


def hip_lib():
    """Load the HIP back-end; fails loudly when it is missing (no fallback)."""
    global _hip
    if _hip is None:
        host_lib()
        path = os.path.join(_LIBDIR, "libgcge_hip.so")
        if not os.path.exists(path):
            raise RuntimeError("libgcge_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _hip = C.CDLL(path, mode=C.RTLD_GLOBAL)
    return _hip


def load_petsc_binary(path, row_begin=0, row_end=-1):
    """PETSc binary Mat file (what the reference's SLEPc driver reads: test/test_app_slepc.c:416-445) -> host CSR;
    rows [row_begin, row_end) with global columns (row_end < 0: to the end)."""
    h = host_lib()
    h.gcge_load_petsc_binary.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(CSR)]
    A = CSR()
    rc = h.gcge_load_petsc_binary(os.fsencode(path), row_begin, row_end, C.byref(A))
    if rc != 0:
        raise RuntimeError("gcge_load_petsc_binary(%r) failed: %d" % (path, rc))
    return A


def load_matrix_market(path):
    """Matrix Market coordinate file (the form SuiteSparse ships the reference's SiO2 / Ga41As41H72 ... in) -> host CSR, full
    matrix (symmetric files are expanded), ascending columns."""
    h = host_lib()
    h.gcge_load_matrix_market.argtypes = [C.c_char_p, C.POINTER(CSR)]
    A = CSR()
    rc = h.gcge_load_matrix_market(os.fsencode(path), C.byref(A))
    if rc != 0:
        raise RuntimeError("gcge_load_matrix_market(%r) failed: %d" % (path, rc))
    return A


def make_problem(kind, size, row_begin=0, row_end=-1, **kw):
    """Returns (A, B) CSR structs (B is None for standard problems)."""
    h = host_lib()
    A, B = CSR(), CSR()
    if kind == "lap3d":
        rc = h.gcge_problem_lap3d(C.c_int(size), C.c_int64(row_begin), C.c_int64(row_end), C.byref(A)); B = None
    elif kind == "fe1d":
        rc = h.gcge_problem_fe1d(C.c_int(size), C.byref(A), C.byref(B))
    elif kind == "fe3d":
        rc = h.gcge_problem_fe3d(C.c_int(size), C.c_int64(row_begin), C.c_int64(row_end), C.byref(A), C.byref(B))
    elif kind == "sio2":
        rc = h.gcge_problem_sio2_like(C.c_int(size), C.c_int(kw.get("K", 8)), C.c_double(kw.get("R0", 1.5)),
                                      C.c_double(kw.get("R1", 3.0)), C.c_uint64(kw.get("seed", 12345)),
                                      C.c_int64(row_begin), C.c_int64(row_end), C.byref(A)); B = None
    elif kind == "sio2ball":
        rc = h.gcge_problem_sio2_ball(C.c_int(size), C.c_int(kw.get("K", 8)), C.c_double(kw.get("R0", 1.5)),
                                      C.c_double(kw.get("R1", 3.0)), C.c_uint64(kw.get("seed", 12345)),
                                      C.c_int64(row_begin), C.c_int64(row_end), C.byref(A), None); B = None
    else:
        raise ValueError(kind)
    if rc != 0:
        raise RuntimeError("problem generator failed rc=%d" % rc)
    return A, B


def ball_geometry(size, row_begin=0, row_end=-1):
    """box index x + G (y + G z) of every row of the ball matrix make_problem("sio2ball", size) (numpy int32 array)."""
    import numpy as np
    h = host_lib()
    A = CSR()
    bp = C.POINTER(C.c_int)()
    rc = h.gcge_problem_sio2_ball(C.c_int(size), C.c_int(0), C.c_double(1.0), C.c_double(0.0), C.c_uint64(1),
                                  C.c_int64(row_begin), C.c_int64(row_end), C.byref(A), C.byref(bp))
    if rc != 0:
        raise RuntimeError("gcge_problem_sio2_ball failed rc=%d" % rc)
    out = np.ctypeslib.as_array(bp, shape=(A.nrows,)).astype(np.int32).copy()
    h.gcge_free_ints(bp)
    h.gcge_csr_free(C.byref(A))
    return out


def make_argv(args):
    arr = (C.c_char_p * (len(args) + 1))()
    for i, a in enumerate(args):
        arr[i] = str(a).encode()
    return len(args), arr


def run_gcg(ops, matA, matB, args, flag=0, quiet=True, keep_evec=False, given=None):
    """GCGE_RunGCG through the operator table `ops` (a void* OPS handle).
    keep_evec: also return the eigenvector multivector handle (nevMax columns; the caller destroys it).
    given = (multivector handle with nevMax columns, nevGiven): warm start from its first nevGiven columns
    (GCGE_RunGCGGiven); the eigenvectors come back in the same block, which stays the caller's."""
    import numpy as np
    h = host_lib()
    args = ["gcge"] + [str(a) for a in args]
    if quiet and "-gcge_print_usage" not in args:
        args += ["-gcge_print_usage", "0"]
    argc, argv = make_argv(args)
    nev = 30
    nev_max = None
    for i, a in enumerate(args):
        if a == "-nevConv":
            nev = int(args[i + 1])
        if a == "-nevMax":
            nev_max = int(args[i + 1])
    nev_max = nev_max or 2 * nev
    ev = np.zeros(nev_max)
    res = RunResult()
    evec = C.c_void_p()
    if given is not None:
        h.GCGE_RunGCGGiven.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p]
        rc = h.GCGE_RunGCGGiven(matA, matB, flag, argc, C.cast(argv, C.c_void_p), ops,
                                ev.ctypes.data_as(C.c_void_p), given[0], int(given[1]), C.cast(C.byref(res), C.c_void_p))
        if rc != 0:
            raise RuntimeError("GCGE_RunGCGGiven rc=%d" % rc)
        return ev, res
    rc = h.GCGE_RunGCG(matA, matB, flag, argc, argv, ops,
                       ev.ctypes.data_as(C.POINTER(C.c_double)), C.byref(evec) if keep_evec else None, C.byref(res))
    if rc != 0:
        raise RuntimeError("GCGE_RunGCG rc=%d" % rc)
    if keep_evec:
        return ev, res, evec
    return ev, res


class HipBackend:
    """Placeholder filled in by gcge_amd/hip_backend.py once the HIP library is loaded."""
    def __init__(self, *a, **k):
        from .hip_backend import HipBackendImpl
        self.__class__ = HipBackendImpl
        HipBackendImpl.__init__(self, *a, **k)
