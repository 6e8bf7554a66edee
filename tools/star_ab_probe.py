"""A/B of the two forms of the plane sweep (spmm_star.hip) on the SiO2-like matrix: whole product and with column sums.
    python tools/star_ab_probe.py G K [m]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
K = int(sys.argv[2]) if len(sys.argv) > 2 else 354
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend(); g = hip.g
if os.environ.get("GCGE_STAR_XCD") is not None:      # 0: workgroups in launch order (before round 4's XCD-aware order)
    g.gcge_hip_spmm_star_xcd.argtypes = [C.c_int]; g.gcge_hip_spmm_star_xcd(int(os.environ["GCGE_STAR_XCD"]))
g.gcge_hip_spmm_star_lanes.argtypes = [C.c_int]
g.gcge_hip_spmm_star_form.argtypes = [C.c_int]
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_mat_spmm_form.restype = C.c_char_p
g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
g.gcge_hip_spmm_dot2_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
A, B = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
mA = hip.matrix(A)
print("n", A.nrows, "nnz", A.nnz, "form", g.gcge_hip_mat_spmm_form(mA).decode(), flush=True)
g.gcge_hip_mat_form_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
st = (C.c_double * 12)()
if g.gcge_hip_mat_form_stats(mA, st):
    print("beyond the star: %d dense blocks hold %d non-zeros (%d stored entries), %d non-zeros in the listed rows" % (st[0], st[2], st[3], st[4]), flush=True)
hip.set_random_mode(1, 7)
ops = hip.ops
V = ops.mv_create(m, mA); ops.set_random(V, 0, m)
W = {2: ops.mv_create(m, mA), 3: ops.mv_create(m, mA), 4: ops.mv_create(m, mA)}
for which in (2, 3, 4, 2, 3, 4):
    g.gcge_hip_spmm_star_lanes(4 if which == 2 else 8)           # 2: 8-column passes on 16 x 16 patches; 3: 16-column passes on 16 x 8 patches
    g.gcge_hip_spmm_star_form(3 if which == 4 else 2)            # 4: third form (LDS-DMA strips)
    ops.spmm(mA, V, W[which], (0, 0), (m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(6):
        ops.spmm(mA, V, W[which], (0, 0), (m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    t = ms.value / cnt
    dots, yy = np.zeros(m), np.zeros(m)
    g.gcge_hip_profile_enable(1)
    for _ in range(4):
        g.gcge_hip_spmm_dot2_mv(mA, V, W[which], (C.c_int * 2)(0, 0), (C.c_int * 2)(m, m), dots.ctypes.data, yy.ctypes.data, hip.ops_handle)
    ms2, by2 = C.c_double(), C.c_double()
    cnt2 = g.gcge_hip_profile_spmm(m, C.byref(ms2), C.byref(by2))
    g.gcge_hip_profile_enable(0)
    print("form %d: product %.3f ms = %.1f %% of 8 TB/s on the CSR bytes; with column sums %.3f ms" % (which, t, by.value / cnt / t * 1e-6 / 80, ms2.value / cnt2), flush=True)
a = hip.mv_to_numpy(W[2], A.nrows, 0, m); b = hip.mv_to_numpy(W[3], A.nrows, 0, m)
print("max |16-column form - 8-column form| / max =", float(np.max(np.abs(a - b)) / np.max(np.abs(a))))
b = hip.mv_to_numpy(W[4], A.nrows, 0, m)
print("max |third form - 8-column form| / max =", float(np.max(np.abs(a - b)) / np.max(np.abs(a))))
g.gcge_hip_spmm_star_form(2)
g.gcge_hip_set_spmm_path.argtypes = [C.c_int]
g.gcge_hip_set_spmm_path(3)
ops.spmm(mA, V, W[3], (0, 0), (m, m))
c = hip.mv_to_numpy(W[3], A.nrows, 0, m)
print("max |grid form - pad-8 on the whole matrix| / max =", float(np.max(np.abs(a - c)) / np.max(np.abs(c))))
g.gcge_hip_set_spmm_path(0)
