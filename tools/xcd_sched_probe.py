"""Experiment: XCD-aware chunk schedule for the GENERIC pad-8 SpMM on the 7-point Laplacian.
   python tools/xcd_sched_probe.py [N] [m]
Each XCD (blocks with the same blockIdx.x % 8) walks, plane after plane, the SAME eighth of every grid plane — rows r with
(r mod S) in [x S/8, (x+1) S/8), S = N^2 the dominant far offset — so the "+S" rows it fetched for one plane are the
centre rows of the next one in ITS L2 (an eighth of a plane of 16 columns is 1 MB).  Only the order of the chunks changes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gcge_amd import HipBackend, make_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
hip = HipBackend(); g = hip.g
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g.gcge_hip_spmm_pad8_schedule.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
g.gcge_hip_spmm_pad8_tune.argtypes = [C.c_int] * 4
A, _ = make_problem("lap3d", N)
mA = hip.matrix(A)
n = A.nrows
hip.set_random_mode(1, 7)
ops = hip.ops
V = ops.mv_create(m, mA); ops.set_random(V, 0, m)
W = ops.mv_create(m, mA); W2 = ops.mv_create(m, mA)
g.gcge_hip_set_spmm_path(2)


def timeit(tag):
    ops.spmm(mA, V, W, (0, 0), (m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(5):
        ops.spmm(mA, V, W, (0, 0), (m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    print("%-52s %.3f ms  %.1f %% of 8 TB/s on the CSR bytes" % (tag, ms.value / cnt, by.value / cnt / (ms.value / cnt) / 8e9 * 100), flush=True)


ref = None
S = N * N
for rpw in (8, 4):
    for pas in (0, 16, 32):
        g.gcge_hip_spmm_pad8_tune(rpw, 8, 1, pas)
        g.gcge_hip_spmm_pad8_schedule(None, 0, 0, 0)
        timeit("natural rpw=%d pass=%d" % (rpw, pas))
        if ref is None:
            ops.spmm(mA, V, W2, (0, 0), (m, m)); hip.sync()
            ref = hip.mv_to_numpy(W2, n, 0, 8)
        rows_per_chunk = 4 * rpw
        nchunk = (n + rows_per_chunk - 1) // rows_per_chunk
        r0 = np.arange(nchunk, dtype=np.int64) * rows_per_chunk
        for Jx in (64, 128, 256):
            xcd = ((r0 % S) * 8) // S
            lists = [np.nonzero(xcd == x)[0] for x in range(8)]     # ascending chunk id = plane after plane
            ln = max(len(c) for c in lists)
            ln = (ln + Jx - 1) // Jx * Jx
            sched = np.full((8, ln), -1, dtype=np.int32)
            for x in range(8):
                sched[x, :len(lists[x])] = lists[x]
            d = torch.from_numpy(sched.ravel()).cuda()
            g.gcge_hip_spmm_pad8_schedule(C.c_void_p(d.data_ptr()), ln, rpw, 8 * Jx)
            timeit("XCD owns 1/8 of every plane, rpw=%d pass=%d Jx=%d" % (rpw, pas, Jx))
            got = hip.mv_to_numpy(W, n, 0, 8)
            assert np.array_equal(got, ref), "schedule changed the product"
            g.gcge_hip_spmm_pad8_schedule(None, 0, 0, 0)
            del d
