"""ctypes calls of the solver-stack Setup functions (include/gcge_solver.h) used by the tests."""
import ctypes as C

import numpy as np

from gcge_amd.lib import host_lib


def orth_setup(ops_handle):
    h = host_lib()
    keep = {}

    def setup(method, block, reorth, zero_tol, mv_ws):
        dbl = np.zeros(200000)
        keep["dbl"] = dbl
        fn = h.MultiVecOrthSetup_BinaryGramSchmidt if method == 1 else h.MultiVecOrthSetup_ModifiedGramSchmidt
        fn.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
        fn(block, reorth, zero_tol, mv_ws, dbl.ctypes.data_as(C.POINTER(C.c_double)), ops_handle)
    return setup


class _BPCG(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("rate", C.c_double), ("tol", C.c_double), ("tol_type", C.c_char * 8),
                ("mv_ws", C.c_void_p * 3), ("dbl_ws", C.c_void_p), ("int_ws", C.c_void_p), ("pc", C.c_void_p),
                ("MatDotMultiVec", C.c_void_p), ("niter", C.c_int), ("residual", C.c_double)]


def bpcg_setup(ops_handle):
    h = host_lib()
    from gcge_amd.ops_struct import OPS

    def setup(max_iter, rate, tol, ws, solve):
        dbl = np.zeros(64); iw = np.zeros(64, dtype=np.int32)
        arr = (C.c_void_p * 3)(*[w.value if isinstance(w, C.c_void_p) else w for w in ws])
        h.MultiLinearSolverSetup_BlockPCG.argtypes = [C.c_int, C.c_double, C.c_double, C.c_char_p, C.c_void_p,
                                                      C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p,
                                                      C.c_void_p, C.c_void_p]
        h.MultiLinearSolverSetup_BlockPCG(max_iter, rate, tol, b"abs", arr, dbl.ctypes.data_as(C.POINTER(C.c_double)),
                                          iw.ctypes.data_as(C.POINTER(C.c_int)), None, None, ops_handle)
        solve()
        st = C.cast(C.cast(ops_handle, C.POINTER(OPS)).contents.multi_linear_solver_workspace, C.POINTER(_BPCG)).contents
        return st.niter
    return setup
