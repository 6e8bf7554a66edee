"""r = b - A x with b formed (MODE 5) against b = x diag(s) taken as scale factors (MODE 6): bitwise comparison (debug aid)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gcge_amd import HipBackend, make_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hip = HipBackend(); g = hip.g
A, _ = make_problem("lap3d", N)
mA = hip.matrix(A); n = A.nrows
rng = np.random.default_rng(5)
X = rng.random((n, m)) - 0.5
s = rng.random(m) * 0.2 + 0.01
Bm = X * s
x, b = hip.mv_from_numpy(mA, X), hip.mv_from_numpy(mA, Bm)
r1, p1 = hip.mv_from_numpy(mA, np.zeros((n, m))), hip.mv_from_numpy(mA, np.zeros((n, m)))
r2, p2 = hip.mv_from_numpy(mA, np.zeros((n, m))), hip.mv_from_numpy(mA, np.zeros((n, m)))
g.gcge_hip_cg_start_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
g.gcge_hip_cg_start_scaled_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
rho1, rho2 = np.zeros(m), np.zeros(m)
print("rc", g.gcge_hip_cg_start_mv(mA, x, 0, b, 0, r1, p1, 0, m, rho1.ctypes.data),
      g.gcge_hip_cg_start_scaled_mv(mA, x, 0, s.ctypes.data, r2, p2, 0, m, rho2.ctypes.data))
R1, R2 = hip.mv_to_numpy(r1, n, 0, m), hip.mv_to_numpy(r2, n, 0, m)
P1, P2 = hip.mv_to_numpy(p1, n, 0, m), hip.mv_to_numpy(p2, n, 0, m)
print("r equal", np.array_equal(R1, R2), "p equal", np.array_equal(P1, P2), "rho equal", np.array_equal(rho1, rho2),
      "max |dr|", np.abs(R1 - R2).max(), "max |r|", np.abs(R1).max())
