"""K1 products of BASELINE config 5's matrix (SiO2-like 171^3, K = 2000) on 64 columns, nothing else: what tools/pmc_traffic_c5.py
runs under rocprofv3 --pmc.    python3 tools/k1_c5_probe.py [G] [K] [m] [products]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 171
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
nprod = int(sys.argv[4]) if len(sys.argv) > 4 else 4
hip = HipBackend()
A, B = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
mA = hip.matrix(A)
hip.set_random_mode(1, 7)
V = hip.ops.mv_create(m, mA); hip.ops.set_random(V, 0, m)
W = hip.ops.mv_create(m, mA)
for _ in range(nprod):
    hip.ops.spmm(mA, V, W, (0, 0), (m, m))
hip.sync()
print("done", A.nrows, A.nnz, flush=True)
