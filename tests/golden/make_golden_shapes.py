"""Fixtures at the SOLVER SHAPES of BASELINE.json configs 2-4 (block size, nevMax and with them the workspace
relations of ops_eig_sol_gcg.c:1275-1280,1641-1645) on reduced grids, produced by the compiled reference
(oracle/_ref).  Run in the build container:  python tests/golden/make_golden_shapes.py
  C2 shape: standard problem,    nev  50, block  64, nevMax 128   (Lap3D 24^3)
  C3 shape: generalised problem, nev 100, block 128, nevMax 256   (P1 stiffness/mass pair 20^3)
  C4 shape: standard problem,    nev 200, block 128, nevMax 400   (Lap3D 28^3)
  C5 shape: standard problem,    nev 100, block  64, nevMax 200   (SiO2-like matrix on a 24^3 grid, 12 atoms of up to 6.5 cells;
            parameters of test/test_eig_sol_SiO2_MAT.c:39-76 of the reference)
  C1 itself: Lap3D 50^3, nev 20 with the harness defaults (nevMax 40, block 20): BASELINE config 1 at full size
Data only: parameters of the generated inputs and the reference's outputs."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyoracle as po  # noqa: E402
from gcge_amd.lib import make_problem  # noqa: E402

assert po.ref_lib() is not None, "build oracle/_ref first (make -C oracle)"
runs = {}
for key, kind, size, nev, block, nev_max, kw in (("c2shape_lap3d_24", "lap3d", 24, 50, 64, 128, {}),
                                                  ("c3shape_fe3d_20", "fe3d", 20, 100, 128, 256, {}),
                                                  ("c4shape_lap3d_28", "lap3d", 28, 200, 128, 400, {}),
                                                  ("c5shape_sio2_24", "sio2", 24, 100, 64, 200, {"K": 12, "R0": 2.0, "R1": 4.5, "seed": 12345}),
                                                  ("c1_lap3d_50", "lap3d", 50, 20, 0, 0, {})):
    A, B = make_problem(kind, size, **kw)
    ev, conv, it, sec = po.ref_gcg(A, B, nev, nev_max=nev_max, block=block)
    runs[key] = {"kind": kind, "size": size, "kw": kw, "nev": nev, "nev_max": nev_max, "block": block, "nev_init": 0, "extra": [],
                 "n": A.nrows, "nnz": int(A.nnz), "nevConv": conv, "numIter": it, "eval": ev[:conv].tolist()}
    print(key, "conv", conv, "it", it, "%.1fs" % sec, "lambda1 %.14e" % ev[0], flush=True)
with open(os.path.join(HERE, "gcg_shapes.json"), "w") as f:
    json.dump(runs, f, indent=0)
