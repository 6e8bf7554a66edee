import ctypes as C, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
from gcge_amd import HipBackend, make_problem
from helpers import uniform
hip = HipBackend(); g = hip.g
g.gcge_hip_set_mgs_fusion.argtypes = [C.c_int]
A, _ = make_problem("lap3d", 11); n = A.nrows; mh = hip.matrix(A)
V0 = uniform(41, (n, 40)) - 0.5
def run(fuse, steps):
    g.gcge_hip_set_mgs_fusion(fuse)
    v = hip.mv_from_numpy(mh, V0); ws = hip.ops.mv_create(40, mh); end = 29; outs = []
    W = V0.copy()
    for k in range(5, min(end, 5 + steps)):
        r = hip.ops.qtap("S", "N", v, None, v, (k, k), (end, k + 1), ws, ld=end - k)[:, 0]
        rn = W[:, k:end].T @ W[:, k]
        nrm = np.sqrt(r[0])
        hip.ops.axpby(0.0, None, 1.0 / nrm, v, (k, k), (k + 1, k + 1))
        W[:, k] /= np.sqrt(rn[0])
        if k < end - 1:
            coef = np.ascontiguousarray(-r[1:] / nrm)
            hip.ops.lincomb(v, v, (k, k + 1), (k + 1, end), coef, 1, beta=np.ones(1), incb=0)
            W[:, k + 1:end] += np.outer(W[:, k], -rn[1:] / np.sqrt(rn[0]))
        print("fuse", fuse, "k", k, "dots err", np.max(np.abs(r - rn)) / np.max(np.abs(rn)))
    out = hip.mv_to_numpy(v, n, 0, 40)
    print("fuse", fuse, "final err vs numpy", np.max(np.abs(out - W)))
    return out
a = run(1, 40); b = run(0, 40)
print("fused vs plain", np.max(np.abs(a - b)), "per column", np.max(np.abs(a - b), axis=0)[:32])
