#!/usr/bin/env python3
"""Why do the three forms of the fused CG take 34 / 29 / 34 outer iterations on Lap3D 16^3, nev 12?
Runs the case of tests/test_hip_parity.py::test_gcg_recompute_cg_equals_stored_product_cg in every form, with and without the
one-sweep start of the stored form, for nev = 12 (the 12th pair lies INSIDE the six-fold cluster (3,2,1): the locking rule of
src/ops_eig_sol_gcg.c:253-259 then needs all 17 pairs) and nev = 11 (the count ends on a cluster boundary), and prints the
iteration counts and the per-iteration converged counts."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from gcge_amd import HipBackend  # noqa: E402
from helpers import gcg_on, lap3d_exact  # noqa: E402

hip = HipBackend()
g = hip.g
g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
forms = (("recompute", {}), ("stored", {"GCGE_CG_NO_RECOMPUTE": "1"}), ("stored, formed b", {"GCGE_CG_NO_RECOMPUTE": "1", "GCGE_CG_NO_FUSED_START": "1"}),
         ("stored, host scalars", {"GCGE_CG_NO_RECOMPUTE": "1", "GCGE_CG_STORED_HOST": "1"}),
         ("stored, host scalars, formed b", {"GCGE_CG_NO_RECOMPUTE": "1", "GCGE_CG_STORED_HOST": "1", "GCGE_CG_NO_FUSED_START": "1"}),
         ("recompute, formed b", {"GCGE_NO_RHS_SCALE": "1"}))
for nev in (12, 11, 17):
    ex = lap3d_exact(16, 24)
    print("nev = %d: lambda_%d = %.6f, lambda_%d = %.6f (relative gap %.1e)" % (nev, nev, ex[nev - 1], nev + 1, ex[nev], (ex[nev] - ex[nev - 1]) / ex[nev]))
    for tag, env in forms:
        os.environ.update(env)
        try:
            g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
            hip.set_random_mode(0)
            ev, res = gcg_on(hip, "lap3d", 16, ["-nevConv", nev, "-nevMax", 24, "-blockSize", 8], flag=1)
            err = np.max(np.abs(ev[:res.nevConv] - ex[:res.nevConv]) / ex[:res.nevConv])
            print("  %-34s outer iterations %3d, converged %2d, max rel err %.1e" % (tag, res.numIter, res.nevConv, err), flush=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
