"""Per-level product check of the SLAB multigrid hierarchy of the SiO2-like matrix (ranks sharing cuda:0 over gloo):
every level's slab (gcge_mg_build_slab, csrc/host/multigrid.c) is uploaded through the slab constructor of this transport
(gcge_amd.dist.hip_slab_matrix — what multigrid_create_slab does through the registered factory) and multiplied; the
result is compared with the same slab's rows applied on the host (scipy).  Narrows down where BlockAMG on slabs of config 5
stops converging with more than two levels.

python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 \
    tests/slab_level_worker.py SIZE [K,R0,R1] [LEVELS] [PLANES PER CUT UNIT]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gcge_amd import HipBackend, make_problem
    from gcge_amd import dist as gdist
    from gcge_amd.lib import CSR
    from helpers import uniform, mg_hierarchy_slab
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    K, R0, R1 = (sys.argv[2] if len(sys.argv) > 2 else "100,2.0,5.0").split(",")
    levels = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    kw = dict(K=int(K), R0=float(R0), R1=float(R1), seed=12345)
    n_global = N ** 3
    part0 = gdist.row_partition(n_global, world)
    A0, _ = make_problem("sio2", N, row_begin=part0[rank], row_end=part0[rank + 1], **kw)
    # cuts on plane numbers that stay EVEN down the hierarchy (gcge_mg_build_slab stops coarsening at the first odd cut)
    # (argv[4] = 1: cuts on ANY plane boundary — every rank pairs its own planes, the cells next to an odd cut are the rank's own)
    unit = int(sys.argv[4]) if len(sys.argv) > 4 else (1 << (levels - 1))
    part = gdist.partition_by_nnz(dist, A0, part0, align=unit * N * N)
    A, _ = make_problem("sio2", N, row_begin=part[rank], row_end=part[rank + 1], **kw)
    hier = mg_hierarchy_slab(A, (N, N, N), part, rank, levels, scale=0.5)
    be = HipBackend(device=0)
    if os.environ.get("PROBE_DENSE_MODE"):
        be.g.gcge_hip_spmm_dense_mode.argtypes = [C.c_int]
        be.g.gcge_hip_spmm_dense_mode(int(os.environ["PROBE_DENSE_MODE"]))
    comm = gdist.install(be, dist, rank, world, stage_through_host=True)
    g = be.g
    g.gcge_hip_mat_spmm_form.restype = C.c_char_p
    g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
    g.gcge_hip_spmm_dot2_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
    g.gcge_hip_set_halo_overlap.argtypes = [C.c_int]
    worst, forms = 0.0, []
    for lev, (S, pl, dims) in enumerate(zip(hier["A"], hier["part"], hier["dims"])):
        S = S.tocsr(); S.sort_indices()
        n_l, n_loc = int(pl[-1]), int(pl[rank + 1] - pl[rank])
        rp = np.ascontiguousarray(S.indptr, dtype=np.int32)
        ci = np.array(S.indices, dtype=np.int32, copy=True)           # GLOBAL columns: hip_slab_matrix renumbers them in place
        va = np.ascontiguousarray(S.data, dtype=np.float64)
        Al = CSR(n_loc, n_l, int(pl[rank]), int(S.nnz), rp.ctypes.data_as(C.POINTER(C.c_int)), ci.ctypes.data_as(C.POINTER(C.c_int)),
                 va.ctypes.data_as(C.POINTER(C.c_double)))
        mat = gdist.hip_slab_matrix(be, comm, Al, n_l, [int(v) for v in pl], cap_cols=128)
        form = g.gcge_hip_mat_spmm_form(mat).decode()
        forms.append(form)
        X = uniform(7 + lev, (n_l, 66)) - 0.5
        Y = S @ X                                                    # this rank's rows of the level's product
        x = be.mv_from_numpy(mat, X[pl[rank]:pl[rank + 1], :])
        y = be.ops.mv_create(66, mat)
        line = []
        for overlap in (1, 0):
            g.gcge_hip_set_halo_overlap(overlap)
            for m, a, b in [(64, 0, 0), (8, 3, 1), (66, 0, 0), (17, 2, 4)]:
                be.ops.spmm(mat, x, y, (a, b), (a + m, b + m))
                err = float(np.max(np.abs(be.mv_to_numpy(y, n_loc, b, b + m) - Y[:, a:a + m])) / max(1e-300, np.max(np.abs(Y))))
                line.append("spmm[%d,ov%d] %.1e" % (m, overlap, err)); worst = max(worst, err)
            for m, a, b in [(64, 0, 0), (24, 4, 2)]:
                dots, yy = np.zeros(m), np.zeros(m)
                g.gcge_hip_spmm_dot2_mv(mat, x, y, (C.c_int * 2)(a, b), (C.c_int * 2)(a + m, b + m), dots.ctypes.data, yy.ctypes.data, be.ops_handle)
                Yl, Xl = Y[:, a:a + m], X[pl[rank]:pl[rank + 1], a:a + m]
                err = float(np.max(np.abs(be.mv_to_numpy(y, n_loc, b, b + m) - Yl)) / max(1e-300, np.max(np.abs(Y))))
                # (the sums are LOCAL or global depending on the entry point: report both distances)
                loc = np.concatenate([(Xl * Yl).sum(0), (Yl * Yl).sum(0)])
                glo = loc.copy(); t = __import__("torch").from_numpy(glo); dist.all_reduce(t)
                got = np.concatenate([dots, yy])
                e_loc = float(np.max(np.abs(got - loc)) / np.max(np.abs(glo))); e_glo = float(np.max(np.abs(got - glo)) / np.max(np.abs(glo)))
                line.append("dot2[%d,ov%d] y %.1e sums %.1e" % (m, overlap, err, min(e_loc, e_glo))); worst = max(worst, err, min(e_loc, e_glo))
        print("rank %d level %d grid %s rows %d of %d (planes %d..%d) nnz %d form %s\n    %s" % (
            rank, lev, dims, n_loc, n_l, pl[rank] // (dims[0] * dims[1]), pl[rank + 1] // (dims[0] * dims[1]), S.nnz, form, "  ".join(line)), flush=True)
        be.ops.mv_destroy(x, 66); be.ops.mv_destroy(y, 66)
        dist.barrier()
    print("rank %d: worst relative error %.2e -> %s" % (rank, worst, "products OK on every level" if worst < 1e-11 else "A LEVEL'S PRODUCT IS WRONG"), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not worst < 1e-11:
        sys.exit(1)
    if rank == 0:
        print("PASS levels=%d forms=%s" % (len(hier["A"]), ",".join(forms)))


if __name__ == "__main__":
    main()
