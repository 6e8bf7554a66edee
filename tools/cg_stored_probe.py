"""The fused block CG with the product STORED (matrices without a pattern form, or GCGE_CG_NO_RECOMPUTE=1): host-scalar loop
(GCGE_CG_STORED_HOST=1) against the device-scalar loop on the same systems — iterations, true residuals, time per iteration.
    python tools/cg_stored_probe.py kind size [nrhs]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend, make_problem
kind = sys.argv[1] if len(sys.argv) > 1 else "sio2"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nrhs = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
g.gcge_hip_bpcg_residual_form.argtypes = [C.c_int]
g.gcge_hip_bpcg_stored_dev_iters.restype = C.c_long
kw = dict(K=max(4, size ** 3 // 2500), R0=2.0, R1=5.0, seed=12345) if kind.startswith("sio2") else {}
A, _ = make_problem(kind, size, **kw)
n = A.nrows
mat = hip.matrix(A)
g.gcge_hip_mat_spmm_form.restype = C.c_char_p; g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
print(kind, size, "n", n, "form", g.gcge_hip_mat_spmm_form(mat).decode(), flush=True)
rng = np.random.default_rng(3)
Bm = np.asfortranarray(rng.random((n, nrhs)) - 0.5)
rp = np.ctypeslib.as_array(A.rowptr, shape=(n + 1,)); ci = np.ctypeslib.as_array(A.colidx, shape=(rp[-1],)); va = np.ctypeslib.as_array(A.val, shape=(rp[-1],))
import scipy.sparse as sp
S = sp.csr_matrix((va, ci, rp), shape=(n, n))
nb = np.linalg.norm(Bm, axis=0)
for rate, maxit in ((1e-2, 30), (1e-2, 2000), (1e-8, 2000)):
    for form in (0, 2):
        for host in (1, 0):
            if host: os.environ["GCGE_CG_STORED_HOST"] = "1"
            else: os.environ.pop("GCGE_CG_STORED_HOST", None)
            g.gcge_hip_bpcg_residual_form(form)
            g.gcge_hip_bpcg_setup(hip.ops_handle, maxit, rate, 1e-300, b"abs")
            b = hip.mv_from_numpy(mat, Bm); x = hip.mv_from_numpy(mat, np.zeros((n, nrhs)))
            hip.ops.multi_linear_solver(mat, b, x, (0, 0), (nrhs, nrhs)); hip.sync()      # warm (ring, workspaces)
            hip.g.gcge_hip_mv_from_host(x, 0, nrhs, np.zeros((n, nrhs), order="F").ctypes.data_as(C.POINTER(C.c_double)), n)
            before = g.gcge_hip_bpcg_stored_dev_iters()
            t0 = time.time()
            hip.ops.multi_linear_solver(mat, b, x, (0, 0), (nrhs, nrhs)); hip.sync()
            dt = time.time() - t0
            it = C.c_int(); g.gcge_hip_bpcg_stats(None, None, C.byref(it))
            X = hip.mv_to_numpy(x, n, 0, nrhs)
            tr = np.linalg.norm(Bm - S @ X, axis=0) / nb
            print("rate %.0e max_it %4d residual %s %s scalars: %4d iterations, true residual max %.3e min %.3e, %.3f ms per iteration (device-scalar stored iterations %d)" % (
                rate, maxit, "stored " if form == 2 else "rebuilt", "host  " if host else "device", it.value, tr.max(), tr.min(), 1e3 * dt / max(it.value, 1),
                g.gcge_hip_bpcg_stored_dev_iters() - before), flush=True)
            hip.ops.mv_destroy(b, nrhs); hip.ops.mv_destroy(x, nrhs)
