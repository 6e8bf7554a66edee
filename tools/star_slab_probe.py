"""Time the product of BASELINE config 5's matrix as W row slabs would take it — on ONE GPU, over the production transport:
every slab of a plane-aligned W-way cut (by non-zeros) is uploaded in turn with its halo plan looped back onto this rank
(halo row with global id g is served by an own row; RCCL allows send/recv to the own rank), so each product runs pack ->
grouped ncclSend/ncclRecv on the transfer stream || sweep of the inner planes -> unpack -> boundary planes -> blocks +
listed rows.  Printed: per slab the form, planes, halo rows and the time of a 64-column product with and without the
interior / boundary split, the sum over the slabs next to the one-rank product.  Measurement aid.
    python tools/star_slab_probe.py G K W [m]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GCGE_COMM_KEEP_SINGLE"] = "1"
import torch  # noqa: E402,F401

from gcge_amd import HipBackend, make_problem  # noqa: E402
from gcge_amd import dist as gdist  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 171
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m = int(sys.argv[4]) if len(sys.argv) > 4 else 64
kw = dict(K=K, R0=2.0, R1=5.0, seed=12345)
plane, n = G * G, G ** 3
hip = HipBackend(device=0)
g = hip.g
comm = gdist.NativeComm(hip, None, 0, 1)
ip_ = C.POINTER(C.c_int)
g.gcge_hip_mat_create_local_ghosts.restype = C.c_void_p
g.gcge_hip_mat_create_local_ghosts.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip_, ip_, C.POINTER(C.c_double), ip_]
g.gcge_hip_mat_set_halo_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, ip_, ip_, ip_, ip_, C.c_int]
g.gcge_hip_mat_spmm_form.restype = C.c_char_p
g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
g.gcge_hip_mat_star_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
g.gcge_hip_set_halo_overlap.argtypes = [C.c_int]
g.gcge_hip_profile_enable.argtypes = [C.c_int]
g.gcge_hip_profile_spmm.restype = C.c_long
g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
hip.set_random_mode(1, 7)


def timed(mat, reps=5):
    ops = hip.ops
    V = ops.mv_create(m, mat); ops.set_random(V, 0, m)
    Y = ops.mv_create(m, mat)
    ops.spmm(mat, V, Y, (0, 0), (m, m)); hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(reps):
        ops.spmm(mat, V, Y, (0, 0), (m, m))
    hip.sync()
    ms, by = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_spmm(m, C.byref(ms), C.byref(by))
    g.gcge_hip_profile_enable(0)
    ops.mv_destroy(V, m); ops.mv_destroy(Y, m)
    return ms.value / cnt, by.value / cnt


# the cuts: non-zeros per plane, cuts at the plane boundaries nearest to k / W of the total
Ag, _ = make_problem("sio2", G, **kw)
rp = np.ctypeslib.as_array(Ag.rowptr, shape=(n + 1,)).astype(np.int64)
part = gdist.cuts_by_weight(np.add.reduceat(np.diff(rp), np.arange(0, n, plane)).astype(float), W, plane, n)
out = {"G": G, "K": K, "slabs": W, "m": m, "part_planes": [p // plane for p in part]}
mat = hip.matrix(Ag)
t, by = timed(mat)
out["one_rank"] = {"form": g.gcge_hip_mat_spmm_form(mat).decode(), "ms": t, "csr_bytes": by, "frac_of_8TBs": by / t * 1e-6 / 8000}
hip.free_matrix(mat)
del Ag
tot = {0: 0.0, 1: 0.0}
out["slab"] = []
for r in range(W):
    A, _ = make_problem("sio2", G, row_begin=part[r], row_end=part[r + 1], **kw)
    nloc = A.nrows
    ghosts = np.ascontiguousarray(gdist.localize_slab(A), dtype=np.int32)
    ng = int(ghosts.size)
    # halo row g -> an own row: planes below map to the slab's last planes, planes above to its first ones
    own = np.where(ghosts < part[r], ghosts - part[r] + nloc, ghosts - part[r + 1]).astype(np.int32)
    assert ng == 0 or (own.min() >= 0 and own.max() < nloc)
    matp = C.c_void_p(g.gcge_hip_mat_create_local_ghosts(nloc, A.ncols, n, part[r], A.rowptr, A.colidx, A.val, ghosts.ctypes.data_as(ip_)))
    peer, scnt, rcnt = (C.c_int * 2)(0, 0), (C.c_int * 2)(0, ng), (C.c_int * 2)(0, ng)
    assert g.gcge_hip_mat_set_halo_rccl(matp, n, 2, peer, scnt, rcnt, np.ascontiguousarray(own).ctypes.data_as(ip_), m) == 0
    st = (C.c_long * 8)()
    has = g.gcge_hip_mat_star_stats(matp, st)
    e = {"rank": r, "rows": nloc, "nnz": int(A.nnz), "halo_rows": ng, "form": g.gcge_hip_mat_spmm_form(matp).decode(),
         "planes": [int(st[6]), int(st[7])] if has else None, "star_rows": int(st[4]) if has else 0}
    for ov in (0, 1):
        g.gcge_hip_set_halo_overlap(ov)
        t, by = timed(matp)
        e["ms_overlap%d" % ov] = t
        tot[ov] += t
    out["slab"].append(e)
    hip.free_matrix(matp)
out["sum_over_slabs_ms"] = {"no_split": tot[0], "interior_swept_during_exchange": tot[1]}
out["sum_over_one_rank"] = tot[1] / out["one_rank"]["ms"]
comm.finalize()
print(json.dumps(out))
