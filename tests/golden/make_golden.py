#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the REAL reference compiled under oracle/_ref
(`make -C oracle ref`, needs /root/reference + MKL: this container only).

Fixtures are data only: seeds/shapes of the inputs (inputs are re-created from the
gcge_uniform stream) and the outputs the reference produced.
"""
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
from helpers import uniform  # noqa: E402
from gcge_amd.lib import make_problem  # noqa: E402

ref = po.ref_lib()
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
DP = C.POINTER(C.c_double)


def dptr(a):
    return a.ctypes.data_as(DP)


def pair(a, b):
    return (C.c_int * 2)(a, b)


def F(a):
    return np.asfortranarray(a, dtype=np.float64)


def csr_args(A):
    if A is None:
        return None, None, None
    return A.rowptr, A.colidx, A.val


out = {}
A5, _ = make_problem("lap3d", 5)
Afe, Bfe = make_problem("fe3d", 5)
n = A5.nrows
assert n == 125

# ---- spmm -------------------------------------------------------------------
for name, mat in (("spmm_lap", A5), ("spmm_mass", Bfe), ("spmm_null", None)):
    x = F(uniform(11, (n, 7))); y = F(uniform(12, (n, 7)))
    ref.ref_spmm(n, *csr_args(mat), dptr(x), 7, dptr(y), 7, pair(1, 2), pair(6, 7))
    out[name] = {"seed_x": 11, "seed_y": 12, "ncx": 7, "ncy": 7, "start": [1, 2], "end": [6, 7], "y": y.T.tolist()}

# ---- inner products -----------------------------------------------------------
x = F(uniform(21, (n, 5))); y = F(uniform(22, (n, 6)))
for key, nsd, s, e, ld in (("ip_N", "N", (1, 2), (4, 6), 5), ("ip_S", "S", (1, 1), (4, 4), 3),
                           ("ip_D1", "D", (0, 1), (4, 5), 1), ("ip_D3", "D", (0, 1), (4, 5), 3)):
    k, m = e[0] - s[0], e[1] - s[1]
    ip = np.zeros(ld * m + 8)
    yy = x if nsd == "S" else y
    ncy = 5 if nsd == "S" else 6
    ref.ref_inner_prod(C.c_char(nsd.encode()), n, dptr(x), 5, dptr(yy), ncy, pair(*s), pair(*e), dptr(ip), ld)
    out[key] = {"seed_x": 21, "seed_y": 21 if nsd == "S" else 22, "ncx": 5, "ncy": ncy, "nsd": nsd,
                "start": list(s), "end": list(e), "ld": ld, "ip": ip[:ld * m].tolist()}

# ---- QtAP ---------------------------------------------------------------------
q = F(uniform(31, (n, 6))); p = F(uniform(32, (n, 7)))
for key, ntsd, mat, s, e in (("qtap_N", "N", Afe, (1, 2), (5, 6)), ("qtap_S", "S", Afe, (1, 1), (5, 5)),
                             ("qtap_D", "D", Afe, (1, 1), (5, 5)), ("qtap_T", "T", Bfe, (0, 3), (4, 6)),
                             ("qtap_T_null", "T", None, (0, 3), (4, 6)), ("qtap_N_null", "N", None, (0, 3), (4, 6))):
    k, m = e[0] - s[0], e[1] - s[1]
    pp = q if ntsd in ("S", "D") else p
    ncp = 6 if ntsd in ("S", "D") else 7
    ld = 1 if ntsd == "D" else (m if ntsd == "T" else k)
    qap = np.zeros(k * m + 8)
    ws = F(uniform(33, (n, 7)))
    ref.ref_qtap(C.c_char(b"S"), C.c_char(ntsd.encode()), n, dptr(q), 6, *csr_args(mat), dptr(pp), ncp,
                 pair(*s), pair(*e), dptr(qap), ld, dptr(ws), 7)
    out[key] = {"seed_q": 31, "seed_p": 31 if ntsd in ("S", "D") else 32, "seed_ws": 33, "ncq": 6, "ncp": ncp,
                "ntsd": ntsd, "mat": None if mat is None else ("A" if mat is Afe else "B"),
                "start": list(s), "end": list(e), "ld": ld, "qap": qap[:k * m].tolist(), "ws": ws.T.tolist()}

# ---- axpby --------------------------------------------------------------------
for key, alpha, beta, has_x, s, e in (("axpby_gen", 2.0, -0.5, True, (1, 2), (4, 5)),
                                      ("axpby_beta0", 1.5, 0.0, True, (0, 3), (2, 5)),
                                      ("axpby_scale", 0.0, 3.0, False, (0, 1), (3, 4)),
                                      ("axpby_inplace", 1.0, 0.0, "same", (0, 3), (2, 5))):
    x = F(uniform(41, (n, 5))); y = F(uniform(42, (n, 6)))
    if has_x == "same":
        ref.ref_axpby(C.c_double(alpha), n, dptr(y), 6, C.c_double(beta), dptr(y), 6, pair(*s), pair(*e))
    else:
        ref.ref_axpby(C.c_double(alpha), n, dptr(x) if has_x else None, 5, C.c_double(beta), dptr(y), 6, pair(*s), pair(*e))
    out[key] = {"seed_x": 41, "seed_y": 42, "alpha": alpha, "beta": beta, "x": has_x, "start": list(s), "end": list(e),
                "y": y.T.tolist()}

# ---- linear combination ---------------------------------------------------------
for key, mode in (("lc_betavec", "vec"), ("lc_betascalar", "scalar"), ("lc_nobeta", "none"),
                  ("lc_scaleonly", "scaleonly"), ("lc_inplace", "inplace")):
    x = F(uniform(51, (n, 5))); y = F(uniform(52, (n, 8)))
    s, e = (1, 2), (4, 7)
    k, m = 3, 5
    coef = F(uniform(53, (4, m))) - 0.5      # ldc = 4 > k
    beta = np.arange(1, 2 * m + 1, dtype=np.float64) * 0.25
    if mode == "vec":
        ref.ref_lincomb(n, dptr(x), 5, dptr(y), 8, pair(*s), pair(*e), dptr(coef), 4, dptr(beta), 2)
    elif mode == "scalar":
        ref.ref_lincomb(n, dptr(x), 5, dptr(y), 8, pair(*s), pair(*e), dptr(coef), 4, dptr(beta), 0)
    elif mode == "none":
        ref.ref_lincomb(n, dptr(x), 5, dptr(y), 8, pair(*s), pair(*e), dptr(coef), 4, None, 0)
    elif mode == "scaleonly":
        ref.ref_lincomb(n, None, 5, dptr(y), 8, pair(*s), pair(*e), None, 0, dptr(beta), 1)
    else:   # y[:,5:8) += y[:,0:3) coef(3x3): x == y, disjoint ranges
        s, e = (0, 5), (3, 8)
        one = np.array([1.0])
        ref.ref_lincomb(n, dptr(y), 8, dptr(y), 8, pair(*s), pair(*e), dptr(coef), 4, dptr(one), 0)
    out[key] = {"seed_x": 51, "seed_y": 52, "seed_c": 53, "mode": mode, "start": list(s), "end": list(e),
                "y": y.T.tolist()}

# ---- random fill -----------------------------------------------------------------
x = np.zeros((n, 3), order="F")
ref.ref_set_random(0, n, dptr(x), 3, 1, 3)
out["set_random"] = {"seed": 0, "start": 1, "end": 3, "x": x.T.tolist()}

# ---- block orthonormalisation ----------------------------------------------------
for key, method, mat, bs in (("orth_mgs_B", 0, Bfe, 4), ("orth_mgs_I", 0, None, -1), ("orth_bgs_B", 1, Bfe, -1)):
    x = F(uniform(61, (n, 20 if method == 1 else 10)))
    nc = x.shape[1]
    if method == 0:
        x[:, 5:10] = x[:, 0:5]          # duplicated columns: rank deficiency (test_orth.c:44-59)
    else:
        x[:, 17] = x[:, 3]
    newend = ref.ref_orth(method, n, dptr(x), nc, 0, nc, *csr_args(mat), bs, 2 if method == 0 else 3, C.c_double(1e-10 if method == 0 else 1e-12))
    out[key] = {"seed_x": 61, "ncols": nc, "method": method, "mat": None if mat is None else "B", "block": bs, "zero_tol": 1e-10 if method == 0 else 1e-12,
                "end": int(newend), "x": x[:, :newend].T.tolist()}
# orthonormalise new columns against existing orthonormal ones (start > 0)
x = F(uniform(62, (n, 8)))
e0 = ref.ref_orth(0, n, dptr(x), 8, 0, 5, *csr_args(Bfe), -1, 2, C.c_double(4.4e-16))
assert e0 == 5
e1 = ref.ref_orth(0, n, dptr(x), 8, 5, 8, *csr_args(Bfe), -1, 2, C.c_double(4.4e-16))
out["orth_mgs_start"] = {"seed_x": 62, "end": int(e1), "x": x[:, :e1].T.tolist()}

# ---- block PCG -------------------------------------------------------------------
b = F(uniform(71, (n, 6))); x = np.zeros((n, 7), order="F")
niter = C.c_int(); resid = C.c_double()
ref.ref_block_pcg(n, *csr_args(A5), dptr(b), 6, dptr(x), 7, pair(1, 2), pair(5, 6), 25, C.c_double(1e-3),
                  C.c_double(1e-14), b"abs", C.byref(niter), C.byref(resid))
out["block_pcg"] = {"seed_b": 71, "start": [1, 2], "end": [5, 6], "max_iter": 25, "rate": 1e-3, "tol": 1e-14,
                    "niter": niter.value, "x": x.T.tolist()}

with open(os.path.join(HERE, "slots.json"), "w") as f:
    json.dump(out, f)
print("slots.json: %d cases" % len(out))

# ---- whole eigensolves ----------------------------------------------------------
runs = {}
for key, kind, size, nev, extra, kw in (
        ("lap3d_12_nev10", "lap3d", 12, 10, (), {}),
        ("lap3d_20_nev20", "lap3d", 20, 20, (), {}),
        ("lap3d_16_nev12_b8", "lap3d", 16, 12, (), {"nev_max": 24, "block": 8}),
        ("fe3d_12_nev10", "fe3d", 12, 10, (), {}),
        ("fe3d_20_nev20", "fe3d", 20, 20, (), {}),
        ("fe1d_807_nev30", "fe1d", 807, 30, (), {}),
        ("sio2_12_nev10", "sio2", 12, 10, (), {}),
        # the same operator on the ball inscribed in the box, rows in scan order (the PARSEC layout behind BASELINE config 5)
        ("sio2ball_16_nev10", "sio2ball", 16, 10, (), {}),
        ("lap3d_12_nev10_bgsX", "lap3d", 12, 10, ("-gcge_initX_orth_method", "bgs"), {}),
        ("lap3d_12_nev10_bqrP", "lap3d", 12, 10, ("-gcge_compP_orth_method", "bqr"), {}),
        # X grows from nevInit to nevMax as pairs lock (ops_eig_sol_gcg.c:1281,1395-1412)
        ("fe3d_14_nev20_init30", "fe3d", 14, 20, (), {"nev_max": 40, "block": 10, "nev_init": 30}),
        ("lap3d_16_nev20_init24", "lap3d", 16, 20, (), {"nev_max": 40, "block": 8, "nev_init": 24}),
        # shifted W solves (A + sigma B) W = ..., fixed and automatic sigma (ops_eig_sol_gcg.c:483-497,709)
        ("lap3d_12_nev10_shift1", "lap3d", 12, 10, ("-gcge_compW_cg_shift", "1.0"), {}),
        ("lap3d_12_nev10_autoshift", "lap3d", 12, 10, ("-gcge_compW_cg_auto_shift", "1"), {}),
        ("fe3d_12_nev10_autoshift", "fe3d", 12, 10, ("-gcge_compW_cg_auto_shift", "1"), {}),
        # second-order W: two CG legs on half of the unconverged columns (ComputeW12, ops_eig_sol_gcg.c:697-923)
        ("lap3d_12_nev10_order2", "lap3d", 12, 10, ("-gcge_compW_cg_order", "2"), {}),
        ("fe3d_12_nev10_order2", "fe3d", 12, 10, ("-gcge_compW_cg_order", "2"), {}),
        ("lap3d_16_nev12_b8_order2_shift", "lap3d", 16, 12, ("-gcge_compW_cg_order", "2", "-gcge_compW_cg_shift", "0.5"),
         {"nev_max": 24, "block": 8}),
):
    A, B = make_problem(kind, size, K=6, R0=1.5, R1=2.0, seed=12345)
    ev, conv, it, sec = po.ref_gcg(A, B, nev, nev_max=kw.get("nev_max", 0), block=kw.get("block", 0),
                                  nev_init=kw.get("nev_init", 0), extra=extra)
    runs[key] = {"kind": kind, "size": size, "nev": nev, "nev_max": kw.get("nev_max", 0), "block": kw.get("block", 0),
                 "nev_init": kw.get("nev_init", 0), "extra": list(extra), "n": A.nrows, "nnz": int(A.nnz), "nevConv": conv, "numIter": it,
                 "eval": ev[:conv].tolist()}
    print(key, "conv", conv, "it", it, "%.3fs" % sec, "lambda1 %.14e" % ev[0])
with open(os.path.join(HERE, "gcg.json"), "w") as f:
    json.dump(runs, f, indent=0)
