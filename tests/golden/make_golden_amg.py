#!/usr/bin/env python3
"""Regenerates tests/golden/amg.json from the REAL reference compiled under oracle/_ref: its BlockAMG
(src/ops_lin_sol.c:466-715) and MultiVecFromItoJ (src/ops_multi_grid.c:69-117) over its dense back-end (app/app_lapack.c — the
only built-in back-end that can run them: app_ccs.c:140-150 refuses a rectangular P^T), and its toy MultiGridCreate
(app_lapack.c:863-929).

Fixtures are data: seeds / shapes of the inputs (re-created from the gcge_uniform stream and the generators) and the outputs."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import pyoracle as po  # noqa: E402
from helpers import uniform, csr_to_scipy, mg_hierarchy  # noqa: E402
from gcge_amd.lib import make_problem  # noqa: E402

ref = po.ref_lib()
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)


def F(a):
    return np.asfortranarray(a, dtype=np.float64)


def dptr(a):
    return a.ctypes.data_as(DP)


def flat(mats):
    return np.concatenate([F(m).ravel(order="F") for m in mats]) if mats else np.zeros(1)


def ref_amg(As, Ps, b, x0, max_iter, rate, tol):
    L = len(As)
    n = (C.c_int * L)(*[a.shape[0] for a in As])
    Af, Pf = flat(As), flat(Ps)
    x = F(x0.copy()); bb = F(b.copy())
    mi = (C.c_int * len(max_iter))(*max_iter); ra = (C.c_double * L)(*rate); to = (C.c_double * L)(*tol)
    niter, res = C.c_int(), C.c_double()
    ref.ref_block_amg_dense(L, n, dptr(Af), dptr(Pf), b.shape[1], dptr(bb), dptr(x), mi, ra, to, b"abs", C.byref(niter), C.byref(res))
    return x, niter.value, res.value


out = {}

# ---- the reference's own toy hierarchy of the 1-D Laplacian, n = 31 -> 15 -> 7 ------------------------------------
n0, L = 31, 3
A0 = F(2.0 * np.eye(n0) - np.eye(n0, k=1) - np.eye(n0, k=-1))
n_out = (C.c_int * L)()
A_out = np.zeros(n0 * n0 * 2); P_out = np.zeros(n0 * n0 * 2)
Lr = ref.ref_dense_multigrid(n0, dptr(A0), L, n_out, dptr(A_out), dptr(P_out))
ns = [n_out[i] for i in range(Lr)]
As, Ps, oa, op = [], [], 0, 0
for lev in range(Lr):
    As.append(A_out[oa:oa + ns[lev] ** 2].reshape(ns[lev], ns[lev], order="F").copy()); oa += ns[lev] ** 2
    if lev + 1 < Lr:
        Ps.append(P_out[op:op + ns[lev] * ns[lev + 1]].reshape(ns[lev], ns[lev + 1], order="F").copy()); op += ns[lev] * ns[lev + 1]
out["toy1d_hierarchy"] = {"n0": n0, "levels": ns, "A": [a.tolist() for a in As[1:]], "P_nnz_rows": [int((p != 0).sum()) for p in Ps]}
m = 3
b = F(uniform(101, (n0, m))); x0 = F(uniform(102, (n0, m)))
max_iter = [2, 3, 3, 2, 2, 10, 0]; rate = [1e-30] * 3; tol = [1e-30] * 3
x, niter, res = ref_amg(As, Ps, b, x0, max_iter, rate, tol)
out["toy1d_amg"] = {"seed_b": 101, "seed_x": 102, "m": m, "max_iter": max_iter, "rate": rate, "tol": tol,
                    "x": x.T.tolist(), "niter": niter, "residual": res}
# stopping by the residual of the cycles: tol[0] reached after the first cycle
max_iter2 = [5, 4, 4, 3, 3, 7, 0]; tol2 = [1e-1, 1e-30, 1e-30]
x, niter, res = ref_amg(As, Ps, b, np.zeros_like(b), max_iter2, rate, tol2)
out["toy1d_amg_stop"] = {"seed_b": 101, "m": m, "max_iter": max_iter2, "rate": rate, "tol": tol2,
                         "x": x.T.tolist(), "niter": niter, "residual": res}
# transfers across two levels
for key, (li, lj) in (("toy1d_from_2_to_0", (2, 0)), ("toy1d_from_0_to_2", (0, 2)), ("toy1d_from_1_to_1", (1, 1))):
    src = F(uniform(111, (ns[li], 2))); dst = F(np.zeros((ns[lj], 2)))
    nn = (C.c_int * Lr)(*ns); Pf = flat(Ps)
    ref.ref_from_i_to_j_dense(Lr, nn, dptr(Pf), li, lj, 2, dptr(src), dptr(dst))
    out[key] = {"seed": 111, "from": li, "to": lj, "y": dst.T.tolist()}

# ---- our aggregation hierarchy (include/gcge_multigrid.h) of Lap3D 8^3, the reference's BlockAMG over it -----------
A, _ = make_problem("lap3d", 8)
lev = mg_hierarchy(A, 3, scale=0.5, min_rows=4)
As = [csr_to_scipy(A).toarray()] + [a.toarray() for a in lev["A"][1:]]
Ps = [p.toarray() for p in lev["P"]]
m = 2
b = F(uniform(121, (512, m))); x0 = F(uniform(122, (512, m)))
max_iter = [2, 3, 3, 2, 2, 6, 0]
x, niter, res = ref_amg(As, Ps, b, x0, max_iter, [1e-30] * 3, [1e-30] * 3)
out["lap3d8_amg"] = {"seed_b": 121, "seed_x": 122, "m": m, "levels": [a.shape[0] for a in As], "scale": 0.5, "min_rows": 4, "max_iter": max_iter,
                     "rate": [1e-30] * 3, "tol": [1e-30] * 3, "x": x.T.tolist(), "niter": niter, "residual": res}

with open(os.path.join(HERE, "amg.json"), "w") as f:
    json.dump(out, f)
print("wrote amg.json:", {k: (v.get("niter"), v.get("residual")) for k, v in out.items() if "niter" in v})
