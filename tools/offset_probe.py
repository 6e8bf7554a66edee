"""Time MatDotMultiVec on the 7-point Laplacian with row-dependent coefficients (patterns by offsets, values streamed per
row) against the pad-8 kernel and against the constant-coefficient matrix.   python tools/offset_probe.py [N] [m]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend, make_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_mat_spmm_form.restype = C.c_char_p
g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
A, _ = make_problem("lap3d", N)
n, nnz = A.nrows, int(A.nnz)
# perturb the values in place: a_ij = -(1 + 0.2 w_i w_j), a_ii = row sum of |a_ij| + 0.1 (symmetric, SPD)
rp = np.ctypeslib.as_array(A.rowptr, shape=(n + 1,)); ci = np.ctypeslib.as_array(A.colidx, shape=(nnz,)); va = np.ctypeslib.as_array(A.val, shape=(nnz,))
rng = np.random.default_rng(1)
w = rng.random(n)
rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
off = rows != ci
va[off] = -(1.0 + 0.2 * w[rows[off]] * w[ci[off]])
va[~off] = 0.0
d = np.bincount(rows, weights=np.abs(va), minlength=n) + 0.1
va[~off] = d
hip.set_random_mode(1, 7)
ops = hip.ops
for label, offs in (("offset patterns + streamed values", 1), ("generic kernels", 0)):
    g.gcge_hip_set_offset_patterns(offs)
    hip.set_random_mode(1, 7)          # the same start block for both forms
    mA = hip.matrix(A)
    V = ops.mv_create(2 * m, mA); ops.set_random(V, 0, 2 * m)
    Wv = ops.mv_create(m, mA)
    ops.spmm(mA, V, Wv, (m, 0), (2 * m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(10):
        ops.spmm(mA, V, Wv, (m, 0), (2 * m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    t = ms.value / cnt
    print("%-36s %-30s m=%d: %.3f ms  %.1f GB/s on CSR bytes (%.1f%% of 8 TB/s)" % (label, g.gcge_hip_mat_spmm_form(mA).decode(), m, t, by.value / cnt / t * 1e-6, by.value / cnt / t * 1e-6 / 80), flush=True)
    if offs == 1:
        ref = hip.mv_to_numpy(Wv, n, 0, 2)
    else:
        print("max rel diff between the two forms: %.2e" % (np.max(np.abs(hip.mv_to_numpy(Wv, n, 0, 2) - ref)) / np.max(np.abs(ref))))
    ops.mv_destroy(V, 2 * m); ops.mv_destroy(Wv, m)
    hip.free_matrix(mA)
