"""Time the two passes of the recompute block CG and the plain product on the Laplacian (tuning aid).
   python tools/cg_pass_probe.py [N] [m] [V_COLS]      env: CHAIN2_NW=16|8|4   (V_COLS: width of the block the m columns sit in)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gcge_amd import HipBackend, make_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_kind.restype = C.c_long
g.gcge_hip_profile_kind.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_cg_pass1_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
g.gcge_hip_cg_pass2_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
A, _ = make_problem("lap3d", N)
mA = hip.matrix(A)
hip.set_random_mode(1, 7)
ops = hip.ops
vc = int(sys.argv[3]) if len(sys.argv) > 3 else m
if os.environ.get("CHAIN2_XCD"):
    g.gcge_hip_spmm_chain2_xcd(int(os.environ["CHAIN2_XCD"]))
if os.environ.get("RING"):          # "0": chain2 kernel for the read-only passes; "1,2" / "1,3": LDS ring, planes ahead
    rr = [int(v) for v in os.environ["RING"].split(",")] + [3]
    g.gcge_hip_spmm_ring_tune.argtypes = [C.c_int, C.c_int]
    g.gcge_hip_spmm_ring_tune(rr[0], rr[1])
if os.environ.get("RING_PRODUCT"):
    g.gcge_hip_spmm_ring_product(int(os.environ["RING_PRODUCT"]))
if os.environ.get("RING_WIDE"):
    g.gcge_hip_spmm_ring_wide(int(os.environ["RING_WIDE"]))
if os.environ.get("RING_XCD"):
    g.gcge_hip_spmm_ring_xcd(int(os.environ["RING_XCD"]))
if os.environ.get("PASS_STREAMS"):   # the 16-column passes of a CG sweep on that many side streams (0 / 1: one after the other)
    g.gcge_hip_cg_pass_streams(int(os.environ["PASS_STREAMS"]))
if os.environ.get("CHAIN2_NW"):
    g.gcge_hip_spmm_chain2_tune(int(os.environ["CHAIN2_NW"]))
vr = int(os.environ.get("R_COLS", vc))      # width of the block r sits in (p, p', w: V_COLS)
p = ops.mv_create(vc, mA); ops.set_random(p, 0, vc)
r = ops.mv_create(vr, mA); ops.set_random(r, 0, vr)
pn = ops.mv_create(vc, mA)
w = ops.mv_create(vc, mA)
al = torch.full((m,), 1e-3, dtype=torch.float64, device="cuda"); be = torch.full((m,), 0.5, dtype=torch.float64, device="cuda")
fl = torch.ones(m, dtype=torch.int32, device="cuda")
pw, ww, rho = np.zeros(m), np.zeros(m), np.zeros(m)


def run(kind, fn, reps=6):
    fn(); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(reps):
        fn()
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_kind(kind, m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    return ms.value / cnt, by.value / cnt / (ms.value / cnt) * 1e-6


t0 = run(0, lambda: ops.spmm(mA, p, w, (0, 0), (m, m)))
t2 = run(2, lambda: g.gcge_hip_cg_pass1_mv(mA, p, 0, m, pw.ctypes.data, ww.ctypes.data))
g.gcge_hip_cg_pass2i_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
if os.environ.get("PASS2_STORED_R"):     # the second pass with a stored residual (reads p, r; writes r, p'): 4 block streams
    t3 = run(3, lambda: g.gcge_hip_cg_pass2_mv(mA, p, r, pn, 0, m, al.data_ptr(), be.data_ptr(), fl.data_ptr(), rho.ctypes.data))
else:                                     # what the solver runs: r rebuilt from the previous direction (here: the block r), 3 block streams
    t3 = run(3, lambda: g.gcge_hip_cg_pass2i_mv(mA, p, r, pn, 0, m, al.data_ptr(), be.data_ptr(), fl.data_ptr(), be.data_ptr(), rho.ctypes.data))
print("N=%d m=%d ld(p)=%d ld(r)=%d nw=%s xcd=%s ring=%s:  product %.3f ms (%.0f GB/s)   pass1 %.3f ms (%.0f GB/s)   pass2 %.3f ms (%.0f GB/s)"
      % (N, m, vc, vr, os.environ.get("CHAIN2_NW", "16"), os.environ.get("CHAIN2_XCD", "0"), os.environ.get("RING", "1,3") + "/x" + os.environ.get("RING_XCD", "0") + "/w" + os.environ.get("RING_WIDE", "0") + "/p" + os.environ.get("RING_PRODUCT", "1"), t0[0], t0[1], t2[0], t2[1], t3[0], t3[1]), flush=True)
