#!/usr/bin/env python3
"""K1 on matrices with NO grid: what today's upload makes of them and what a bandwidth-reducing reorder at
upload would buy with today's kernels.
  (i)  the SiO2-like matrix of BASELINE config 5 (G^3 grid, K atoms) under a random symmetric permutation — and the same matrix
       brought back to a banded form by reverse Cuthill-McKee (scipy), the reorder an opaque-handle back-end could apply at upload;
  (ii) an unstructured tetrahedral P1 stiffness matrix (Delaunay triangulation of random points in a cube; the reference's FE
       provider works on tetrahedra: test/get_mat_phg.c:148) in its random node order and after RCM.
Prints one line per matrix: K1 form, ms per 64-column product, fraction of 8 TB/s on the CSR bytes.
    python tools/generic_probe.py [G] [K] [npoints] [m]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
from scipy.sparse.csgraph import reverse_cuthill_mckee  # noqa: E402
import torch  # noqa: E402,F401
from gcge_amd import HipBackend, make_problem  # noqa: E402
from helpers import csr_from_scipy, csr_to_scipy  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 840
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 400000
m = int(sys.argv[4]) if len(sys.argv) > 4 else 64
hip = HipBackend()
g = hip.g
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_mat_spmm_form.restype = C.c_char_p
g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
if os.environ.get("PROBE_TILE_MODE") is not None:          # 1: row tiles (spmm_tile.hip) for every matrix without a pattern form
    g.gcge_hip_spmm_tile_mode.argtypes = [C.c_int]
    g.gcge_hip_spmm_tile_mode(int(os.environ["PROBE_TILE_MODE"]))
if os.environ.get("PROBE_REORDER_MODE") is not None:       # -1: keep the rows as given (the state before round 5)
    g.gcge_hip_spmm_reorder_mode.argtypes = [C.c_int]
    g.gcge_hip_spmm_reorder_mode(int(os.environ["PROBE_REORDER_MODE"]))


def measure(tag, S, check=None):
    A, keep = csr_from_scipy(S)
    t0 = time.perf_counter()
    mA = hip.matrix(A)
    hip.sync()
    up = time.perf_counter() - t0
    form = g.gcge_hip_mat_spmm_form(mA).decode()
    hip.set_random_mode(1, 7)
    V = hip.ops.mv_create(m, mA); hip.ops.set_random(V, 0, m)
    W = hip.ops.mv_create(m, mA)
    for _ in range(2):
        hip.ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(8):
        hip.ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    t = ms.value / cnt
    # parity on the way: a few columns against scipy
    x = hip.mv_to_numpy(V, S.shape[0], 0, 4)
    y = hip.mv_to_numpy(W, S.shape[0], 0, 4)
    err = float(np.max(np.abs(y - S @ x)) / np.max(np.abs(y)))
    print("%-44s n %8d nnz %10d (%.1f per row): %-34s %7.3f ms = %5.1f %% of 8 TB/s (upload %.1f s, max rel err vs scipy %.1e)"
          % (tag, S.shape[0], S.nnz, S.nnz / S.shape[0], form, t, by.value / cnt / t * 1e-6 / 80, up, err), flush=True)
    hip.ops.mv_destroy(V, m); hip.ops.mv_destroy(W, m)
    hip.free_matrix(mA)


# (i) SiO2-like, permuted
A, _ = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
S = csr_to_scipy(A).tocsr()
measure("SiO2-like %d^3, natural (grid) order" % G, S)
rng = np.random.default_rng(5)
p = rng.permutation(S.shape[0])
Sp = S[p][:, p].tocsr()
Sp.sort_indices()
measure("  ... random symmetric permutation", Sp)
t0 = time.perf_counter()
q = reverse_cuthill_mckee(Sp, symmetric_mode=True)
t_rcm = time.perf_counter() - t0
Sr = Sp[q][:, q].tocsr()
Sr.sort_indices()
bw = int(np.max(np.abs(Sr.tocoo().row - Sr.tocoo().col)))
measure("  ... + reverse Cuthill-McKee (%.1f s, bandwidth %d)" % (t_rcm, bw), Sr)
del S, Sp, Sr

# (ii) unstructured tetrahedral P1 stiffness matrix
from scipy.spatial import Delaunay  # noqa: E402
pts = rng.random((npts, 3))
t0 = time.perf_counter()
tri = Delaunay(pts)
T = tri.simplices                       # (ntet, 4)
X = pts[T]                              # (ntet, 4, 3)
# gradients of the barycentric coordinates: rows of inv([1 x y z])
M = np.concatenate([np.ones((T.shape[0], 4, 1)), X], axis=2)
vol = np.abs(np.linalg.det(M)) / 6.0
ok = vol > 1e-14
T, M, vol = T[ok], M[ok], vol[ok]
Gm = np.linalg.inv(M)[:, 1:, :]         # (ntet, 3, 4): gradient of basis j = Gm[:, :, j]
Ke = np.einsum("tdi,tdj->tij", Gm, Gm) * vol[:, None, None]
I = np.repeat(T[:, :, None], 4, axis=2).ravel()
J = np.repeat(T[:, None, :], 4, axis=1).ravel()
Kfull = sp.coo_matrix((Ke.ravel(), (I, J)), shape=(npts, npts)).tocsr()
hull = np.unique(tri.convex_hull.ravel())
inner = np.setdiff1d(np.arange(npts), hull)        # Dirichlet: the hull nodes are eliminated
Kfe = Kfull[inner][:, inner].tocsr()
Kfe.sort_indices()
print("tetrahedral mesh: %d points, %d tetrahedra, assembled in %.1f s" % (npts, T.shape[0], time.perf_counter() - t0), flush=True)
measure("P1 stiffness on Delaunay tetrahedra, random order", Kfe)
q = reverse_cuthill_mckee(Kfe, symmetric_mode=True)
Kr = Kfe[q][:, q].tocsr()
Kr.sort_indices()
measure("  ... + reverse Cuthill-McKee (bandwidth %d)" % int(np.max(np.abs(Kr.tocoo().row - Kr.tocoo().col))), Kr)
