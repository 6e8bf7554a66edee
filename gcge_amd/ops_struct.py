"""ctypes mirror of `struct OPS_` (include/gcge_ops.h == reference src/ops.h:43-152) and a thin
caller for the slots the GCG hot path uses.  Works with ANY back-end that filled the table
(HIP, the CPU oracle, the host dense table)."""
import ctypes as C

import numpy as np

OPS_FIELDS = [
    "Printf", "GetWtime", "GetOptionFromCommandLine", "MatView", "MatAxpby",
    "VecCreateByMat", "VecCreateByVec", "VecDestroy", "VecView", "VecInnerProd",
    "VecLocalInnerProd", "VecSetRandomValue", "VecAxpby", "MatDotVec", "MatTransDotVec",
    "MultiVecCreateByMat", "MultiVecCreateByVec", "MultiVecCreateByMultiVec", "MultiVecDestroy",
    "GetVecFromMultiVec", "RestoreVecForMultiVec", "MultiVecView", "MultiVecLocalInnerProd",
    "MultiVecInnerProd", "MultiVecSetRandomValue", "MultiVecAxpby", "MultiVecLinearComb",
    "MatDotMultiVec", "MatTransDotMultiVec", "MultiVecQtAP", "lapack_ops", "DenseMatQtAP",
    "DenseMatOrth", "LinearSolver", "linear_solver_workspace", "MultiLinearSolver",
    "multi_linear_solver_workspace", "MultiVecOrth", "orth_workspace", "MultiGridCreate",
    "MultiGridDestroy", "VecFromItoJ", "MultiVecFromItoJ", "EigenSolver",
    "eigen_solver_workspace", "app_ops",
]


class OPS(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in OPS_FIELDS]


_vp, _i, _d, _c = C.c_void_p, C.c_int, C.c_double, C.c_char
_ip, _dp = C.POINTER(C.c_int), C.POINTER(C.c_double)

SIGNATURES = {
    "MultiVecCreateByMat": (None, [C.POINTER(_vp), _i, _vp, _vp]),
    "MultiVecCreateByMultiVec": (None, [C.POINTER(_vp), _i, _vp, _vp]),
    "MultiVecDestroy": (None, [C.POINTER(_vp), _i, _vp]),
    "MultiVecLocalInnerProd": (None, [_c, _vp, _vp, _i, _ip, _ip, _dp, _i, _vp]),
    "MultiVecInnerProd": (None, [_c, _vp, _vp, _i, _ip, _ip, _dp, _i, _vp]),
    "MultiVecSetRandomValue": (None, [_vp, _i, _i, _vp]),
    "MultiVecAxpby": (None, [_d, _vp, _d, _vp, _ip, _ip, _vp]),
    "MultiVecLinearComb": (None, [_vp, _vp, _i, _ip, _ip, _dp, _i, _dp, _i, _vp]),
    "MatDotMultiVec": (None, [_vp, _vp, _vp, _ip, _ip, _vp]),
    "MatTransDotMultiVec": (None, [_vp, _vp, _vp, _ip, _ip, _vp]),
    "MultiVecQtAP": (None, [_c, _c, _vp, _vp, _vp, _i, _ip, _ip, _dp, _i, _vp, _vp]),
    "MultiVecOrth": (None, [_vp, _i, _ip, _vp, _vp]),
    "MultiLinearSolver": (None, [_vp, _vp, _vp, _ip, _ip, _vp]),
}


def _pair(a, b):
    return (C.c_int * 2)(a, b)


class OpsTable:
    """Call slots of an OPS table created in C (handle = void* to struct OPS_)."""

    def __init__(self, handle):
        self.handle = handle if isinstance(handle, C.c_void_p) else C.c_void_p(handle)
        self.struct = C.cast(self.handle, C.POINTER(OPS)).contents

    def fn(self, name):
        ptr = getattr(self.struct, name)
        if not ptr:
            raise RuntimeError("slot %s is NULL" % name)
        res, args = SIGNATURES[name]
        return C.CFUNCTYPE(res, *args)(ptr)

    # -- multivectors
    def mv_create(self, ncols, mat):
        mv = C.c_void_p()
        self.fn("MultiVecCreateByMat")(C.byref(mv), ncols, mat, self.handle)
        return mv

    def mv_destroy(self, mv, ncols=0):
        self.fn("MultiVecDestroy")(C.byref(mv), ncols, self.handle)

    def set_random(self, mv, start, end):
        self.fn("MultiVecSetRandomValue")(mv, start, end, self.handle)

    def axpby(self, alpha, x, beta, y, s, e):
        self.fn("MultiVecAxpby")(alpha, x, beta, y, _pair(*s), _pair(*e), self.handle)

    def lincomb(self, x, y, s, e, coef, ldc, beta=None, incb=0):
        cp = coef.ctypes.data_as(_dp) if coef is not None else None
        bp = beta.ctypes.data_as(_dp) if beta is not None else None
        self.fn("MultiVecLinearComb")(x, y, 0, _pair(*s), _pair(*e), cp, ldc, bp, incb, self.handle)

    def inner_prod(self, nsd, x, y, s, e, ld=None, local=False):
        k, m = e[0] - s[0], e[1] - s[1]
        if nsd == "D":
            ld = ld or 1
            out = np.zeros(max(1, ld * m))
        else:
            ld = ld or k
            out = np.zeros((m, ld)).ravel()
        name = "MultiVecLocalInnerProd" if local else "MultiVecInnerProd"
        self.fn(name)(nsd.encode(), x, y, 0, _pair(*s), _pair(*e), out.ctypes.data_as(_dp), ld, self.handle)
        if nsd == "D":
            return out[::ld][:m].copy()
        return out.reshape(m, ld).T[:k, :].copy()

    def spmm(self, mat, x, y, s, e):
        self.fn("MatDotMultiVec")(mat, x, y, _pair(*s), _pair(*e), self.handle)

    def qtap(self, ntsA, ntsd, q, mat, p, s, e, ws, ld=None):
        k, m = e[0] - s[0], e[1] - s[1]
        if ntsd == "D":
            ld = ld or 1
            out = np.zeros(max(1, ld * m))
        elif ntsd == "T":
            ld = ld or m
            out = np.zeros((k, ld)).ravel()
        else:
            ld = ld or k
            out = np.zeros((m, ld)).ravel()
        self.fn("MultiVecQtAP")(ntsA.encode(), ntsd.encode(), q, mat, p, 0, _pair(*s), _pair(*e),
                                out.ctypes.data_as(_dp), ld, ws, self.handle)
        if ntsd == "D":
            return out[::ld][:m].copy()
        if ntsd == "T":
            return out.reshape(k, ld).T[:m, :].copy()      # m x k (the transpose that was asked for)
        return out.reshape(m, ld).T[:k, :].copy()

    def orth(self, x, start, end, matB):
        e = C.c_int(end)
        self.fn("MultiVecOrth")(x, start, C.byref(e), matB, self.handle)
        return e.value

    def multi_linear_solver(self, mat, b, x, s, e):
        self.fn("MultiLinearSolver")(mat, b, x, _pair(*s), _pair(*e), self.handle)
