"""(script, not collected by pytest: python tests/sweep_fused_cg.py)  GPU robustness sweep: GCG with the fused one-pass block CG (flag 1) against the reference-form BlockPCG over the
HIP slots (flag 0) — converged count, iteration count, Ritz values — over problem kinds, sizes and block shapes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):     # tests/helpers.py loads the oracle
    sys.path.insert(0, _p)
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend
from helpers import gcg_on
hip = HipBackend()
hip.g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
hip.g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
bad = 0
for kind, size in (("lap3d", 16), ("lap3d", 24), ("lap3d", 32), ("fe3d", 16), ("fe3d", 24), ("fe1d", 2000), ("sio2", 12), ("sio2", 16)):
    for nev, blk in ((10, 0), (20, 8), (40, 16), (64, 32)):
        res = {}
        for flag in (0, 1):
            hip.set_random_mode(0)
            args = ["-nevConv", nev, "-gcge_initX_orth_method", "chol", "-gcge_compW_orth_method", "chol"]
            if blk:
                args += ["-blockSize", blk, "-nevMax", nev + 2 * blk]
            ev, r = gcg_on(hip, kind, size, args, flag=flag, K=5, R0=1.5, R1=2.0, seed=7)
            res[flag] = (r.nevConv, r.numIter, ev[:nev].copy(), r.seconds)
        a, b = res[0], res[1]
        ok = a[0] >= nev and b[0] >= nev and abs(a[1] - b[1]) <= max(3, a[1] // 5) and np.max(np.abs(a[2] - b[2]) / np.abs(a[2])) < 1e-9
        bad += not ok
        print("%-6s %5d nev %2d blk %2d | slots BlockPCG conv %3d it %3d %6.2fs | fused CG conv %3d it %3d %6.2fs %s"
              % (kind, size, nev, blk, a[0], a[1], a[3], b[0], b[1], b[3], "" if ok else "<<<<<"))
print("bad", bad)
