"""TEST INFRASTRUCTURE — ctypes access to the CPU oracle (oracle/liboracle.so) and, where it
was built, to the real reference (oracle/_ref/libgcge_ref.so).  Imported by tests/, by
__graft_entry__.smoke() and by the cpu_baseline leg of bench.py only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

os.environ.setdefault("MKL_THREADING_LAYER", "GNU")


class OVec(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_double)), ("nrows", C.c_int), ("ncols", C.c_int), ("ldd", C.c_int)]


class OCcs(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_double)), ("i_row", C.POINTER(C.c_int)),
                ("j_col", C.POINTER(C.c_int)), ("nrows", C.c_int), ("ncols", C.c_int)]


_oracle = None
_ref = None


def build():
    subprocess.run(["make", "-C", _HERE, "oracle"], check=True, capture_output=True)
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=False, capture_output=True)


def oracle_lib():
    global _oracle
    if _oracle is None:
        from gcge_amd.lib import host_lib
        host_lib()
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _oracle = C.CDLL(path, mode=C.RTLD_GLOBAL)
    return _oracle


_ref_omp = None


def ref_lib(omp=False):
    """The real reference; None when it has not been built (e.g. no /root/reference).
    omp=True: the OPS_USE_OMP build (oracle/Makefile target ref_omp) — app_ccs.c:117-131 threaded over the block
    columns with OMP_NUM_THREADS threads (read at run time); needs MKL_THREADING_LAYER=GNU, set above."""
    global _ref, _ref_omp
    if omp:
        if _ref_omp is None:
            path = os.path.join(_HERE, "_ref", "libgcge_ref_omp.so")
            if not os.path.exists(path):
                return None
            try:
                _ref_omp = C.CDLL(path, mode=C.RTLD_LOCAL)
            except OSError:
                return None
        return _ref_omp
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libgcge_ref.so")
        if not os.path.exists(path):
            return None
        try:
            _ref = C.CDLL(path, mode=C.RTLD_LOCAL)
        except OSError:
            return None
    return _ref


def ccs_from_csr(A):
    """ORACLE_CCS view of a gcge_amd.lib.CSR (symmetric: CSR arrays == CCS arrays)."""
    m = OCcs()
    m.data, m.i_row, m.j_col, m.nrows, m.ncols = A.val, A.colidx, A.rowptr, A.nrows, A.ncols
    return m


def make_ops(quiet=True):
    from gcge_amd.lib import host_lib
    h = host_lib()
    o = oracle_lib()
    ops = C.c_void_p()
    h.OPS_Create(C.byref(ops))
    o.OPS_ORACLE_Set(ops)
    h.OPS_Setup(ops)
    h.GCGE_SetQuiet(ops, 1 if quiet else 0)
    return ops


def ref_gcg(A, B, nev, nev_max=0, block=0, nev_init=0, abs_tol=1e-1, rel_tol=1e-8, max_iter=500,
            extra=(), given=None, omp=False):
    """given: (n, nevGiven) array of start vectors (the nevGiven argument of ops->EigenSolver)."""
    r = ref_lib(omp)
    nm = nev_max if nev_max > 0 else 2 * nev
    ev = np.zeros(nm)
    conv, it, sec = C.c_int(), C.c_int(), C.c_double()
    argv = (C.c_char_p * max(1, len(extra)))()
    for i, a in enumerate(extra):
        argv[i] = str(a).encode()
    ng, gp = 0, None
    if given is not None:
        gcm = np.asfortranarray(given, dtype=np.float64)
        ng, gp = gcm.shape[1], gcm.ctypes.data_as(C.POINTER(C.c_double))
    r.ref_gcg_solve_given(C.c_int(A.nrows), A.rowptr, A.colidx, A.val,
                          B.rowptr if B is not None else None, B.colidx if B is not None else None,
                          B.val if B is not None else None,
                          nev, nev_max, block, nev_init, C.c_double(abs_tol), C.c_double(rel_tol), max_iter, 0,
                          len(extra), argv, ev.ctypes.data_as(C.POINTER(C.c_double)), None,
                          C.byref(conv), C.byref(it), C.byref(sec), ng, gp)
    return ev, conv.value, it.value, sec.value
