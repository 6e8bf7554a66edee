"""Sweep of the LDS-ring passes (spmm_ring.hip) over grid sizes, column counts, ring depths and both addressing forms
against the chain + line-exchange kernel and scipy (a robustness aid, run by hand on a GPU box; the pinned cases live in
tests/test_hip_parity.py::test_ring_sweep_equals_chain_kernel_and_numpy).
    python tests/sweep_ring.py          (last run: 200 cases, all through the ring, 0 mismatches)"""
import ctypes as C
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, "tests"), os.path.join(R, "oracle"), R]
import numpy as np
from helpers import csr_to_scipy, uniform
from gcge_amd import HipBackend, make_problem

hip = HipBackend(); g = hip.g
g.gcge_hip_cg_pass1_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
g.gcge_hip_spmm_ring_launches.restype = C.c_long
g.gcge_hip_spmm_ring_tune.argtypes = [C.c_int, C.c_int]
FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
g.gcge_hip_residual_hook.restype = C.c_void_p
hook = FN(g.gcge_hip_residual_hook())
bad = 0; ran = 0; ring_cases = 0
for size in (16, 24, 32, 40, 48, 56, 64, 72, 80, 96):
    A, _ = make_problem("lap3d", size)
    S = csr_to_scipy(A); n = A.nrows
    mat = hip.matrix(A)
    for m in (2, 6, 16, 18, 30, 32, 34, 48, 62, 64):
        for depth, wide in ((3, 0), (2, 1)):
            P = uniform(100 + m, (n, m + 2)) - 0.5
            p = hip.mv_from_numpy(mat, P)
            W = S @ P[:, 2:2 + m]
            lam = uniform(7, (m,)) * 3.0
            Rz = W - P[:, 2:2 + m] * lam
            res = {}
            for on in (1, 0):
                g.gcge_hip_spmm_ring_tune(on, depth); g.gcge_hip_spmm_ring_wide(wide)
                n0 = g.gcge_hip_spmm_ring_launches()
                pw, ww, rs = np.zeros(m), np.zeros(m), np.zeros(m)
                assert g.gcge_hip_cg_pass1_mv(mat, p, 2, m, pw.ctypes.data, ww.ctypes.data) == 0
                assert hook(mat, None, p, 2, 2 + m, lam.ctypes.data, rs.ctypes.data) == 1
                took = g.gcge_hip_spmm_ring_launches() - n0
                res[on] = (pw, ww, rs, took)
            ok = True
            for a, ref in zip(res[1][:3], (np.sum(P[:, 2:2 + m] * W, axis=0), np.sum(W * W, axis=0), np.sum(Rz * Rz, axis=0))):
                ok &= bool(np.allclose(a, ref, rtol=1e-11, atol=1e-12 * n))
            for a, b in zip(res[1][:3], res[0][:3]):
                ok &= bool(np.allclose(a, b, rtol=1e-12, atol=1e-13 * n))
            ran += 1; ring_cases += res[1][3] > 0
            if not ok:
                bad += 1; print("MISMATCH size %d m %d depth %d wide %d" % (size, m, depth, wide), flush=True)
            hip.ops.mv_destroy(p, m + 2)
    hip.free_matrix(mat)
g.gcge_hip_spmm_ring_tune(1, 3); g.gcge_hip_spmm_ring_wide(0)
print("ring sweep: %d cases, %d through the ring, %d mismatches" % (ran, ring_cases, bad))
sys.exit(1 if bad else 0)
