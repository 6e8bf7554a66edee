"""The rows of the config-5 product beside the sweep: which rows seed a dense block (minimal row length) and what is left to the pad-8
list.    python3 tools/list_probe.py [G] [K] [m]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 171
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend()
g = hip.g
A, B = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
g.gcge_hip_spmm_dense_min_len.argtypes = [C.c_int]
g.gcge_hip_mat_form_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
ref = None
g.gcge_hip_spmm_dense_layers.argtypes = [C.c_int, C.c_int]
for min_len, layers in ((96, 1), (96, 4), (32, 1), (32, 4), (48, 4), (32, 8)):
    g.gcge_hip_spmm_dense_min_len(min_len)
    g.gcge_hip_spmm_dense_layers(layers, 32)
    mA = hip.matrix(A)
    st = (C.c_double * 12)()
    g.gcge_hip_mat_form_stats(mA, st)
    hip.set_random_mode(1, 7)
    V = hip.ops.mv_create(m, mA); hip.ops.set_random(V, 0, m)
    W = hip.ops.mv_create(m, mA)
    for _ in range(3):
        hip.ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(20):
        hip.ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    ms_, by_ = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_kind(0, m, C.byref(ms_), C.byref(by_))
    g.gcge_hip_profile_enable(0)
    y = hip.mv_to_numpy(W, A.nrows, 0, 4)
    if ref is None:
        ref = y
    err = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
    print("seed rows >= %3d entries, %d layers: %5d blocks, %6d row blocks, %.3e nnz in blocks (%.3e stored), %.3e nnz listed: %.3f ms = %.1f %% of 8 TB/s, max rel diff vs the first %.1e"
          % (min_len, layers, st[0], st[1], st[2], st[3], st[4], ms_.value / cnt, by_.value / ms_.value / 1e6 / 8000 * 100, err), flush=True)
    hip.ops.mv_destroy(V, m); hip.ops.mv_destroy(W, m)
    hip.free_matrix(mA)
