"""Would the plane sweep gain from requesting a plane's own points and its halo strips in the SAME step (a sweep that scatters a
plane's z contributions forward instead of gathering 13 planes)?  Then neighbouring patches on one XCD ask for the same lines
at about the same time and the strips can be L2 hits; today the owner asks 6 steps before its neighbours, and 32 workgroups x
52 KB per step push a line out of a 4 MB L2 in ~2.4 steps.  Timing probe (gcge_hip_spmm_star_dbg(32): results wrong) under the
XCD-aware workgroup orders of gcge_hip_spmm_star_xcd.
    python tools/star_lead_probe.py [G] [K] [m]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 171
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_spmm_star_xcd.argtypes = [C.c_int]
g.gcge_hip_spmm_star_dbg.argtypes = [C.c_int]
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
A, B = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
mA = hip.matrix(A)
hip.set_random_mode(1, 7)
ops = hip.ops
V = ops.mv_create(m, mA); ops.set_random(V, 0, m)
W = ops.mv_create(m, mA)
only = os.environ.get("PROBE_ONLY")
for dbg in (0, 32):
    for xcd in (0, 11, 22, 33, 44, 66, 1):
        if only is not None and only != "%d,%d" % (dbg, xcd):
            continue
        g.gcge_hip_spmm_star_dbg(dbg); g.gcge_hip_spmm_star_xcd(xcd)
        ops.spmm(mA, V, W, (0, 0), (m, m)); hip.sync()
        g.gcge_hip_profile_enable(1)
        for _ in range(8):
            ops.spmm(mA, V, W, (0, 0), (m, m))
        hip.sync()
        ms, by = C.c_double(), C.c_double()
        cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
        g.gcge_hip_profile_enable(0)
        print("own-point lead %s, xcd order %2d: product %.3f ms" % ("2 (probe)" if dbg else "8 (form 3)", xcd, ms.value / cnt), flush=True)
