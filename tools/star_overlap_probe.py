"""Would the blocks + listed rows hide behind the plane sweep if they ran beside it?  MEASUREMENT ONLY: gcge_hip_star_race_probe puts
them on a second stream without ordering them against the sweep (the results are wrong then); the time of the product tells what
a properly ordered form (compact scratch for the listed rows + one add) could reach.
    python tools/star_overlap_probe.py G K [m]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
K = int(sys.argv[2]) if len(sys.argv) > 2 else 354
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_star_race_probe.argtypes = [C.c_int]
A, _ = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
mA = hip.matrix(A)
hip.set_random_mode(1, 7)
ops = hip.ops
V = ops.mv_create(m, mA); ops.set_random(V, 0, m)
W = ops.mv_create(m, mA)
for mode in (0, 1, 2, 0, 1, 2):
    g.gcge_hip_star_race_probe(mode)
    ops.spmm(mA, V, W, (0, 0), (m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(8):
        ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    t = ms.value / cnt
    print("%s: product %.3f ms = %.1f %% of 8 TB/s on the CSR bytes" % (("in order", "beside the sweep (launched after it)", "beside the sweep (launched before it)")[mode], t, by.value / cnt / t * 1e-6 / 80), flush=True)
g.gcge_hip_star_race_probe(0)
