"""Time MatDotMultiVec (and the fused SpMM+dot of the block CG) through the operator table. Tuning aid.
   python tools/spmm_probe.py [kind] [size] [m] [path]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from gcge_amd import HipBackend, make_problem
kind = sys.argv[1] if len(sys.argv) > 1 else "lap3d"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
path = int(sys.argv[4]) if len(sys.argv) > 4 else 0
hip = HipBackend(); g = hip.g
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_mat_patterns.argtypes = [C.c_void_p]
K, R0, R1 = (float(t) for t in os.environ.get("SIO2", "60,1.5,2.0").split(","))   # SiO2-like generator: atoms, radius law
A, B = make_problem(kind, N, K=int(K), R0=R0, R1=R1, seed=12345)
mA = hip.matrix(A)
print("n", A.nrows, "nnz", A.nnz, "patterns", g.gcge_hip_mat_patterns(mA))
hip.set_random_mode(1, 7)
ops = hip.ops
vc = int(os.environ.get('V_COLS', '256'))
V = ops.mv_create(vc, mA); ops.set_random(V, 0, vc)
Wv = ops.mv_create(m, mA)
g.gcge_hip_set_spmm_path(path)
if os.environ.get('PAT_LINE'):
    g.gcge_hip_spmm_pattern_tune_line(int(os.environ['PAT_LINE']))
if os.environ.get('CHAIN2_NW') is not None:
    g.gcge_hip_spmm_chain2_tune(int(os.environ['CHAIN2_NW']))
if os.environ.get('CHAIN2_XCD') is not None:
    g.gcge_hip_spmm_chain2_xcd(int(os.environ['CHAIN2_XCD']))
if os.environ.get('PAD8'):
    g.gcge_hip_spmm_pad8_tune(*[int(t) for t in os.environ['PAD8'].split(',')])
if os.environ.get('CHAIN_LPR'):
    g.gcge_hip_spmm_chain_tune(int(os.environ['CHAIN_LPR']))
if os.environ.get('PAT_GRID'):
    g.gcge_hip_spmm_pattern_tune(int(os.environ['PAT_GRID']))
for x0 in (192, 128):
    ops.spmm(mA, V, Wv, (x0, 0), (x0 + m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(5):
        ops.spmm(mA, V, Wv, (x0, 0), (x0 + m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    t = ms.value / cnt
    print("path %d  m=%d x0=%d (V cols %d -> ld %d): %.3f ms  %.1f GB/s on CSR bytes (%.1f%% of 8 TB/s)" % (path, m, x0, vc, m, t, by.value / cnt / t * 1e-6, by.value / cnt / t * 1e-6 / 80))
