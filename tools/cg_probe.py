"""Time the fused block CG alone (one MultiLinearSolver call on Lap3D N^3, 64 right-hand sides). Tuning aid."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend, make_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "lap3d"
hip = HipBackend()
hip.g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
A, _ = make_problem(kind, N, K=60, R0=1.5, R1=2.0, seed=12345)
mA = hip.matrix(A)
hip.set_random_mode(1, 7)
ops = hip.ops
V = ops.mv_create(256, mA)
ops.set_random(V, 0, 256)
if os.environ.get('PAD8'):
    hip.g.gcge_hip_spmm_pad8_tune(*[int(t) for t in os.environ['PAD8'].split(',')])
if os.environ.get('CHAIN2_NW') is not None:
    hip.g.gcge_hip_spmm_chain2_tune(int(os.environ['CHAIN2_NW']))
if os.environ.get('CHAIN_LPR'):
    hip.g.gcge_hip_spmm_chain_tune(int(os.environ['CHAIN_LPR']))
hip.g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
for rep in range(2):
    hip.sync(); t = time.perf_counter()
    ops.multi_linear_solver(mA, V, V, (0, 192), (64, 256))
    hip.sync(); print("block CG 30 its: %.1f ms" % (1e3 * (time.perf_counter() - t)))
