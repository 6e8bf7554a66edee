"""What bounds the plane sweep: the second form with parts of its work switched off (results are wrong then; timing only).
    python tools/star_dbg_probe.py G K [m]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
K = int(sys.argv[2]) if len(sys.argv) > 2 else 354
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_spmm_star_dbg.argtypes = [C.c_int]
A, B = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
mA = hip.matrix(A)
hip.set_random_mode(1, 7)
ops = hip.ops
V = ops.mv_create(m, mA); ops.set_random(V, 0, m)
W = ops.mv_create(m, mA)
import time
for bits, what in [(0, "everything"), (1, "no halo loads"), (2, "no LDS arm reads"), (4, "no stores"), (8, "no own-plane loads"), (3, "no halo loads, no LDS reads"),
                   (9, "no global loads at all"), (13, "no global traffic at all"), (15, "nothing but the loop"), (6, "no LDS reads, no stores"), (16, "no barriers"), (0, "everything")]:
    g.gcge_hip_spmm_star_dbg(bits)
    for _ in range(2):
        ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    print("%-32s whole product %.3f ms" % (what, (time.perf_counter() - t0) * 100), flush=True)
g.gcge_hip_spmm_star_dbg(0)
