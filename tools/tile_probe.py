"""Time MatDotMultiVec on a matrix without a pattern form: supernode blocks on MFMA + remainder (spmm_dense.hip), the tile
path (spmm_tile.hip, TILE_MODE=1) and the pad-8 kernel. Tuning aid.
   python tools/tile_probe.py G K [m]      (SiO2-like matrix on a G^3 grid with K atoms, R = 2 + 5 u1 u2)"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
K = int(sys.argv[2]) if len(sys.argv) > 2 else 354
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_spmm_tile_mode.argtypes = [C.c_int]
g.gcge_hip_spmm_dense_mode.argtypes = [C.c_int]
if os.environ.get("TILE_MODE"):
    g.gcge_hip_spmm_tile_mode(int(os.environ["TILE_MODE"]))      # 1: also keep the tile form (of the remainder when blocks exist)
g.gcge_hip_spmm_star_mode.argtypes = [C.c_int]
if os.environ.get("STAR_MODE"):
    g.gcge_hip_spmm_star_mode(int(os.environ["STAR_MODE"]))      # -1: no grid form (spmm_star.hip)
if os.environ.get("DENSE_MODE"):
    g.gcge_hip_spmm_dense_mode(int(os.environ["DENSE_MODE"]))    # -1: no supernode blocks
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_mat_spmm_form.restype = C.c_char_p
g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
t0 = time.time()
A, B = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
t1 = time.time()
mA = hip.matrix(A)
t2 = time.time()
print("n", A.nrows, "nnz", A.nnz, "gen %.1f s upload %.1f s" % (t1 - t0, t2 - t1), "form", g.gcge_hip_mat_spmm_form(mA).decode(), flush=True)
g.gcge_hip_mat_form_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
st = (C.c_double * 12)()
if g.gcge_hip_mat_form_stats(mA, st):
    print("dense blocks %d (%d row blocks of 32), non-zeros in blocks %d = %.1f %% of the matrix, stored entries %d (fill %.1f %%), remainder %d non-zeros;"
          % (st[0], st[1], st[2], 100.0 * st[2] / A.nnz, st[3], 100.0 * st[2] / max(st[3], 1), st[4]), flush=True)
    if st[5] > 0:
        print("remainder tiles %d, X rows staged per matrix row %.2f, ELL entries per non-zero %.3f, overflow entries %d, brick %dx%dx%d"
              % (st[5], st[6], st[7], st[8], st[9], st[10], st[11]), flush=True)
g.gcge_hip_mat_star_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
ss = (C.c_long * 8)()
if g.gcge_hip_mat_star_stats(mA, ss):
    print("star rows %d of %d (%.1f %%) on a %dx%dx%d grid, arm length %d; the block / tile figures above are those of the other rows"
          % (ss[4], ss[5], 100.0 * ss[4] / ss[5], ss[0], ss[1], ss[2], ss[3]), flush=True)
hip.set_random_mode(1, 7)
ops = hip.ops
V = ops.mv_create(2 * m, mA); ops.set_random(V, 0, 2 * m)
W1 = ops.mv_create(m, mA); W2 = ops.mv_create(m, mA); W4 = ops.mv_create(m, mA)
res = {}
for path in (0, 4, 3):
    g.gcge_hip_set_spmm_path(path)
    Wv = {0: W1, 4: W4, 3: W2}[path]
    ops.spmm(mA, V, Wv, (m, 0), (2 * m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(5):
        ops.spmm(mA, V, Wv, (m, 0), (2 * m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    t = ms.value / cnt
    print("path %d (%s) m=%d: %.3f ms  %.1f GB/s on CSR bytes (%.1f%% of 8 TB/s)" % (path, g.gcge_hip_mat_spmm_form(mA).decode(), m, t, by.value / cnt / t * 1e-6, by.value / cnt / t * 1e-6 / 80), flush=True)
g.gcge_hip_set_spmm_path(0)
a = hip.mv_to_numpy(W1, A.nrows, 0, 4); b = hip.mv_to_numpy(W2, A.nrows, 0, 4)
print("max |automatic form - pad8| / max|pad8| =", float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
