"""One fused block-CG solve (30 iterations, 64 right-hand sides) on the SiO2-like matrix — the stored-product form that matrices
without a pattern take.  Run under `rocprofv3 --kernel-trace --stats` for the kernel shares of an iteration.
    python tools/cg_generic_probe.py [G K m]        GCGE_CG_TRACE=1 prints the wall time of each call"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 171
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
g.gcge_hip_mat_spmm_form.restype = C.c_char_p
g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
A, _ = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
mA = hip.matrix(A)
print("n", A.nrows, "nnz", A.nnz, "form", g.gcge_hip_mat_spmm_form(mA).decode(), flush=True)
hip.set_random_mode(1, 7)
ops = hip.ops
b = ops.mv_create(m, mA); x = ops.mv_create(m, mA)
ops.set_random(b, 0, m)
g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-30, 1e-300, b"abs")     # 30 iterations whatever the residual
g.gcge_hip_bpcg_residual_form.argtypes = [C.c_int]
form = int(os.environ.get("RESIDUAL_FORM", "1"))                      # 1: r rebuilt from the ring (what the GCG harness's rate 1e-2 selects), 2: stored
g.gcge_hip_bpcg_residual_form(form)
print("residual form", form, flush=True)
for rep in range(3):
    ops.axpby(0.0, None, 0.0, x, (0, 0), (m, m))
    hip.sync(); t0 = time.perf_counter()
    ops.multi_linear_solver(mA, b, x, (0, 0), (m, m))
    hip.sync(); dt = time.perf_counter() - t0
    it = C.c_int(); g.gcge_hip_bpcg_stats(None, None, C.byref(it))
    print("solve %d: %d iterations, %.1f ms = %.2f ms per iteration" % (rep, it.value, 1e3 * dt, 1e3 * dt / max(1, it.value)), flush=True)
