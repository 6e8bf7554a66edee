"""K1 on the SiO2-like operator restricted to the BALL inscribed in the grid (rows = grid points inside a sphere in scan order, the
PARSEC layout of the matrices behind BASELINE config 5): with the geometry named (gcge_hip_mat_create_grid: the star rows take the
plane sweep through a row map), with the geometry recovered from the rows (what gcge_hip_mat_create does by itself) and without
it (dense blocks + pad-8), and the pad-8 kernel alone.  Measurement aid.
    python tools/ball_probe.py G K [m]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend, make_problem
from gcge_amd.lib import ball_geometry
G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
K = int(sys.argv[2]) if len(sys.argv) > 2 else 354
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_mat_spmm_form.restype = C.c_char_p
g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
g.gcge_hip_set_spmm_path.argtypes = [C.c_int]
g.gcge_hip_mat_form_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
g.gcge_hip_mat_star_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
t0 = time.time()
A, _ = make_problem("sio2ball", G, K=K, R0=2.0, R1=5.0, seed=12345)
box = ball_geometry(G)
print("ball in %d^3: n %d of %d box points, nnz %d (%.1f per row), gen %.1f s" % (G, A.nrows, G ** 3, A.nnz, A.nnz / A.nrows, time.time() - t0), flush=True)
ops = hip.ops
rng = np.random.default_rng(7)
X = np.asfortranarray(rng.random((A.nrows, m)) - 0.5)      # the SAME operand for every form
res = {}
g.gcge_hip_spmm_star_infer.argtypes = [C.c_int]
g.gcge_hip_spmm_star_masked_third.argtypes = [C.c_int]
g.gcge_hip_mat_star_masked_form.argtypes = [C.c_void_p]
for tag in ("with the geometry", "with the geometry, second form of the sweep", "geometry recovered from the rows", "without"):
    t1 = time.time()
    g.gcge_hip_spmm_star_infer(0 if tag == "without" else 1)
    g.gcge_hip_spmm_star_masked_third(0 if "second form" in tag else 1)
    mA = hip.matrix_grid(A, (G, G, G), box) if tag.startswith("with the geometry") else hip.matrix(A)
    form = g.gcge_hip_mat_spmm_form(mA).decode()
    st = (C.c_double * 12)(); ss = (C.c_long * 8)()
    note = ""
    if g.gcge_hip_mat_form_stats(mA, st):
        note = "; %d dense blocks hold %d non-zeros, %d in the listed / remaining rows" % (st[0], st[2], st[4])
    if g.gcge_hip_mat_star_stats(mA, ss):
        note += "; star rows %d of %d (%.1f %%), sweep form %d" % (ss[4], ss[5], 100.0 * ss[4] / ss[5], g.gcge_hip_mat_star_masked_form(mA))
    print("%s: form %s, upload %.1f s%s" % (tag, form, time.time() - t1, note), flush=True)
    V = hip.mv_from_numpy(mA, X)
    W = ops.mv_create(m, mA)
    for path in ((0, 3) if tag == "without" else (0,)):
        g.gcge_hip_set_spmm_path(path)
        ops.spmm(mA, V, W, (0, 0), (m, m)); hip.sync()
        g.gcge_hip_profile_enable(1)
        for _ in range(12):
            ops.spmm(mA, V, W, (0, 0), (m, m))
        hip.sync()
        ms, by = C.c_double(), C.c_double()
        cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
        g.gcge_hip_profile_enable(0)
        t = ms.value / cnt
        print("  %-40s m=%d: %.3f ms  %.1f GB/s on the CSR bytes (%.1f %% of 8 TB/s)" % (g.gcge_hip_mat_spmm_form(mA).decode(), m, t, by.value / cnt / t * 1e-6, by.value / cnt / t * 1e-6 / 80), flush=True)
        res[(tag, path)] = hip.mv_to_numpy(W, A.nrows, 0, 8)
    g.gcge_hip_set_spmm_path(0)
    ops.mv_destroy(V, m); ops.mv_destroy(W, m)
    hip.free_matrix(mA)
ref = res[("without", 3)]
for k, v in res.items():
    print("max |%s, path %d - pad-8| / max = %.2e" % (k[0], k[1], float(np.max(np.abs(v - ref)) / np.max(np.abs(ref)))))

# z ranges per patch of the masked third form (balance against the planes of warm-up every range re-reads)
g.gcge_hip_spmm_star_masked_zchunks.argtypes = [C.c_int]
g.gcge_hip_spmm_star_infer(1); g.gcge_hip_spmm_star_masked_third(1)
mA = hip.matrix_grid(A, (G, G, G), box)
V = hip.mv_from_numpy(mA, X); W = ops.mv_create(m, mA)
for zc in (1, 2, 3, 4):
    g.gcge_hip_spmm_star_masked_zchunks(zc)
    ops.spmm(mA, V, W, (0, 0), (m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(12):
        ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    t = ms.value / cnt
    err = float(np.max(np.abs(hip.mv_to_numpy(W, A.nrows, 0, 8) - ref)) / np.max(np.abs(ref)))
    print("third form, %d z range(s) per patch: %.3f ms (%.1f %% of 8 TB/s), max rel diff vs pad-8 %.1e" % (zc, t, by.value / cnt / t * 1e-6 / 80, err), flush=True)
g.gcge_hip_spmm_star_masked_zchunks(0)
