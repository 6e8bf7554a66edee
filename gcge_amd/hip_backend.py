"""Python handle on the HIP back-end (libgcge_hip.so).  Plumbing only."""
import ctypes as C

import numpy as np

from .lib import CSR, host_lib, hip_lib
from .ops_struct import OpsTable


class HipBackendImpl:
    def __init__(self, device=0, quiet=True):
        self.h = host_lib()
        self.g = hip_lib()
        self.g.gcge_hip_mat_create_csr.restype = C.c_void_p
        self.g.gcge_hip_mat_create_csr.argtypes = [C.POINTER(CSR)]
        self.g.gcge_hip_mat_destroy.argtypes = [C.c_void_p]
        self.g.gcge_hip_mv_to_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_long]
        self.g.gcge_hip_mv_from_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_long]
        self.g.gcge_hip_set_random_mode.argtypes = [C.c_int, C.c_ulonglong]
        if self.g.gcge_hip_init(device) != 0:
            raise RuntimeError("HIP back-end: no GPU visible (there is no CPU fallback)")
        self.ops_handle = C.c_void_p()
        self.h.OPS_Create(C.byref(self.ops_handle))
        self.g.OPS_HIP_Set(self.ops_handle)
        self.h.OPS_Setup(self.ops_handle)
        self.h.GCGE_SetQuiet(self.ops_handle, 1 if quiet else 0)
        self.ops = OpsTable(self.ops_handle)

    def matrix(self, csr):
        m = self.g.gcge_hip_mat_create_csr(C.byref(csr))
        if not m:
            raise RuntimeError("gcge_hip_mat_create_csr failed")
        return C.c_void_p(m)

    def matrix_grid(self, csr, dims, box_of_row):
        """A matrix on a masked grid: box_of_row[r] = x + nx (y + ny z) (int32 array, ascending); see gcge_hip_mat_create_grid."""
        g = self.g
        g.gcge_hip_mat_create_grid.restype = C.c_void_p
        g.gcge_hip_mat_create_grid.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                               C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        box = np.ascontiguousarray(box_of_row, dtype=np.int32)
        m = g.gcge_hip_mat_create_grid(csr.nrows, csr.rowptr, csr.colidx, csr.val, int(dims[0]), int(dims[1]), int(dims[2]),
                                       box.ctypes.data_as(C.POINTER(C.c_int)))
        if not m:
            raise RuntimeError("gcge_hip_mat_create_grid failed")
        return C.c_void_p(m)

    def free_matrix(self, m):
        self.g.gcge_hip_mat_destroy(m)

    def matrix_rect(self, csr):
        """A rectangular matrix (a prolongation of a multigrid hierarchy): MatDotMultiVec applies it, MatTransDotMultiVec its
        transpose (gcge_hip_mat_create_rect_csr, csrc/hip/multigrid.hip)."""
        self.g.gcge_hip_mat_create_rect_csr.restype = C.c_void_p
        self.g.gcge_hip_mat_create_rect_csr.argtypes = [C.POINTER(CSR)]
        m = self.g.gcge_hip_mat_create_rect_csr(C.byref(csr))
        if not m:
            raise RuntimeError("gcge_hip_mat_create_rect_csr failed")
        return C.c_void_p(m)

    def free_matrix_rect(self, m):
        self.g.gcge_hip_mat_destroy(m)

    def mv_from_numpy(self, mat, arr):
        """arr: (n, ncols) array -> device multivector with the same columns."""
        a = np.asfortranarray(arr, dtype=np.float64)
        mv = self.ops.mv_create(a.shape[1], mat)
        self.g.gcge_hip_mv_from_host(mv, 0, a.shape[1], a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0])
        return mv

    def mv_to_numpy(self, mv, n, c0, c1):
        out = np.zeros((n, c1 - c0), order="F")
        self.g.gcge_hip_mv_to_host(mv, c0, c1, out.ctypes.data_as(C.POINTER(C.c_double)), n)
        return out

    def set_random_mode(self, mode, seed=12345):
        self.g.gcge_hip_set_random_mode(mode, seed)

    def sync(self):
        self.g.gcge_hip_sync()
