"""Experiment: per-XCD brick schedules for the generic pad-8 SpMM on a wide-stencil grid matrix (SiO2-like, 37-point).
   SIO2=K,R0,R1 python tools/brick_probe.py G m
The schedule only permutes the order in which 4-row chunks are processed (every chunk exactly once), so the result
is the same matrix product; the question is how much fabric traffic the order saves (see profiles/r01_cases)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gcge_amd import HipBackend, make_problem
G = int(sys.argv[1]) if len(sys.argv) > 1 else 171
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K, R0, R1 = (float(t) for t in os.environ.get("SIO2", "2000,2,5").split(","))
hip = HipBackend(); g = hip.g
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_spmm_pad8_schedule.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
g.gcge_hip_spmm_pad8_tune.argtypes = [C.c_int] * 4
A, _ = make_problem("sio2", G, K=int(K), R0=R0, R1=R1, seed=12345)
mA = hip.matrix(A)
n = A.nrows
print("n", n, "nnz", A.nnz, flush=True)
hip.set_random_mode(1, 7)
ops = hip.ops
V = ops.mv_create(m, mA); ops.set_random(V, 0, m)
W = ops.mv_create(m, mA); W2 = ops.mv_create(m, mA)


def timeit(tag):
    ops.spmm(mA, V, W, (0, 0), (m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(4):
        ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    print("%-40s %.3f ms" % (tag, ms.value / cnt), flush=True)


for pas in (0, 16):
    g.gcge_hip_spmm_pad8_tune(1, 8, 1, pas)
    g.gcge_hip_spmm_pad8_schedule(None, 0, 0, 0)
    timeit("natural pass=%d" % pas)
    if pas == 0:
        ops.spmm(mA, V, W2, (0, 0), (m, m)); hip.sync()
        ref = hip.mv_to_numpy(W2, n, 0, m)
    nchunk = (n + 3) // 4
    r0 = np.arange(nchunk, dtype=np.int64) * 4
    i, j, k = r0 % G, (r0 // G) % G, r0 // (G * G)
    for (bx, by, bz, Jx) in ((12, 12, 12, 96), (16, 16, 16, 96), (24, 8, 8, 96), (171, 4, 4, 96), (16, 16, 16, 48), (8, 8, 8, 96), (20, 20, 20, 128)):
        nbx, nby = (G + bx - 1) // bx, (G + by - 1) // by
        brick = (i // bx) + nbx * ((j // by) + nby * (k // bz))
        order = np.lexsort((i, j, k, brick))          # by brick, inside: k, j, i
        b_sorted = brick[order]
        ub, start = np.unique(b_sorted, return_index=True)
        lists = [[] for _ in range(8)]
        ends = list(start[1:]) + [nchunk]
        for q, (s, e) in enumerate(zip(start, ends)):
            lists[q % 8].append(order[s:e])
        cat = [np.concatenate(l) for l in lists]
        ln = max(len(c) for c in cat)
        ln = (ln + Jx - 1) // Jx * Jx
        sched = np.full((8, ln), -1, dtype=np.int32)
        for x in range(8):
            sched[x, :len(cat[x])] = cat[x]
        d = torch.from_numpy(sched.ravel()).cuda()
        g.gcge_hip_spmm_pad8_schedule(C.c_void_p(d.data_ptr()), ln, 1, 8 * Jx)
        timeit("brick %dx%dx%d Jx=%d pass=%d" % (bx, by, bz, Jx, pas))
        got = hip.mv_to_numpy(W, n, 0, m)
        assert np.array_equal(got, ref), "schedule changed the product"
        g.gcge_hip_spmm_pad8_schedule(None, 0, 0, 0)
        del d
