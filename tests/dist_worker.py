"""Worker of the 2-rank tests (launched by tests/test_dist.py with RANK/WORLD_SIZE/MASTER_* set).
mode 'oracle': CPU oracle back-end, gloo.   mode 'hip': HIP back-end on cuda:0 for every rank, gloo
transport staged through the host (what differs from production is only the torch.distributed backend).
mode 'hip_native': one device per rank, RCCL from C inside the back-end (needs as many GPUs as ranks)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def box_exact(dims, count):
    cs = [2.0 * np.cos(np.arange(1, d + 1) * np.pi / (d + 1)) for d in dims]
    lam = (6.0 - cs[0][:, None, None] - cs[1][None, :, None] - cs[2][None, None, :]).ravel()
    return np.sort(lam)[:count]


def main():
    mode = sys.argv[1]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gcge_amd import dist as gdist
    from gcge_amd.lib import CSR, host_lib, run_gcg
    from helpers import uniform, csr_to_scipy
    h = host_lib()
    dims = (8, 8, 10)      # planes of 64 rows: the slab matrices qualify for the chain layout of the pattern SpMM
    sio2 = None            # "sio2:G": SiO2-like matrix on a G^3 grid (rows of very different length), rows split by nnz
    star = False           # "sio2star:G": the same with cuts on plane boundaries — every slab keeps the plane sweep of spmm_star.hip
    ball = None            # "sio2ball:G": the same operator on the BALL inside the box (a masked grid, rows in scan order), cuts between grid lines
    geometry = None
    ballfile = False       # "ballfile:G": the ball matrix as a Matrix-Market FILE — no geometry named: recovered from the rows of the file
    if len(sys.argv) > 2 and sys.argv[2].startswith("ballfile:"):
        ball, ballfile = int(sys.argv[2].split(":")[1]), True
    elif len(sys.argv) > 2 and sys.argv[2].startswith("sio2ball:"):
        ball = int(sys.argv[2].split(":")[1])
    elif len(sys.argv) > 2 and sys.argv[2].startswith("sio2star:"):
        sio2, star = int(sys.argv[2].split(":")[1]), True
    elif len(sys.argv) > 2 and sys.argv[2].startswith("sio2:"):
        sio2 = int(sys.argv[2].split(":")[1])
    elif len(sys.argv) > 2:
        dims = tuple(int(t) for t in sys.argv[2].split(","))
    if ball:
        from gcge_amd.lib import make_problem, ball_geometry
        kw = dict(K=12, R0=1.5, R1=3.0, seed=12345)
        box = ball_geometry(ball)
        n_global = int(box.size)
        Ag, _ = make_problem("sio2ball", ball, **kw)
        assert Ag.nrows == n_global
        S = csr_to_scipy(Ag)
        bdims = (ball, ball, ball)
        keep = []
        if ballfile:
            # what a user of the reference's SiO2 file has: a file, no grid.  Rank 0 writes it, every rank reads it, recovers the geometry
            # from the rows (gcge_hip_star_infer_grid: host only), cuts between the recovered lines and takes its rows out of the file's CSR
            import tempfile
            from gcge_amd import load_matrix_market
            from gcge_amd.lib import hip_lib
            path = os.path.join(tempfile.gettempdir(), "gcge_ballfile_%s_%d.mtx" % (os.environ.get("MASTER_PORT", "0"), ball))
            h.gcge_save_matrix_market.argtypes = [C.c_char_p, C.POINTER(CSR), C.c_int]
            if rank == 0:
                assert h.gcge_save_matrix_market(path.encode(), C.byref(Ag), 1) == 0
            dist.barrier()
            L = load_matrix_market(path)
            dist.barrier()
            if rank == 0:
                os.remove(path)
            assert (L.nrows, L.nnz) == (Ag.nrows, Ag.nnz)
            gl = hip_lib()
            gl.gcge_hip_star_infer_grid.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
            fd = (C.c_int * 3)()
            fbox = np.zeros(n_global, dtype=np.int32)
            assert gl.gcge_hip_star_infer_grid(L.nrows, L.rowptr, L.colidx, fd, fbox.ctypes.data_as(C.POINTER(C.c_int))) == 1, "no grid found in the rows of the file"
            bdims, box = tuple(int(v) for v in fd), fbox
            assert all(d <= ball for d in bdims) and np.all(np.diff(box.astype(np.int64)) > 0), (bdims,)
        part = gdist.partition_lines(box, bdims[0], world)
        assert part[0] == 0 and part[-1] == n_global and all(part[q] < part[q + 1] for q in range(world)), part
        assert all(q == 0 or box[part[q]] // bdims[0] != box[part[q] - 1] // bdims[0] for q in range(world)), "cuts must lie between grid lines"
        if ballfile:
            rp = np.ctypeslib.as_array(L.rowptr, shape=(n_global + 1,))
            lo, hi = int(rp[part[rank]]), int(rp[part[rank + 1]])
            srp = np.ascontiguousarray(rp[part[rank]:part[rank + 1] + 1] - lo, dtype=np.int32)
            sci = np.ascontiguousarray(np.ctypeslib.as_array(L.colidx, shape=(int(L.nnz),))[lo:hi], dtype=np.int32)
            sva = np.ascontiguousarray(np.ctypeslib.as_array(L.val, shape=(int(L.nnz),))[lo:hi], dtype=np.float64)
            keep += [srp, sci, sva]
            A = CSR(part[rank + 1] - part[rank], n_global, part[rank], hi - lo, srp.ctypes.data_as(C.POINTER(C.c_int)),
                    sci.ctypes.data_as(C.POINTER(C.c_int)), sva.ctypes.data_as(C.POINTER(C.c_double)))
        else:
            A, _ = make_problem("sio2ball", ball, row_begin=part[rank], row_end=part[rank + 1], **kw)
        geometry = (bdims, box)
        sio2 = ball            # (the checks below that only ask "an SiO2-like matrix?")
    elif sio2:
        from gcge_amd.lib import make_problem
        kw = dict(K=8 if star else 40, R0=1.5, R1=3.0, seed=12345)
        n_global = sio2 ** 3
        Ag, _ = make_problem("sio2", sio2, **kw)
        S = csr_to_scipy(Ag)
        part0 = gdist.row_partition(n_global, world)
        A0, _ = make_problem("sio2", sio2, row_begin=part0[rank], row_end=part0[rank + 1], **kw)
        if star:
            assert gdist.grid_of(A0) == (sio2, sio2, sio2, 6)
        # (with BlockAMG of L levels behind it: cuts in units of 2^(L-1) planes, so that they stay even on every level that is coarsened)
        amg_unit = (1 << (int(os.environ["GCGE_TEST_AMG"]) - 1)) if os.environ.get("GCGE_TEST_AMG") else 1
        part = gdist.partition_by_nnz(dist, A0, part0, align=amg_unit * sio2 * sio2 if star else None)
        assert not star or all(p % (sio2 * sio2) == 0 for p in part), part
        A, _ = make_problem("sio2", sio2, row_begin=part[rank], row_end=part[rank + 1], **kw)
        nnz_all = [None] * world
        dist.all_gather_object(nnz_all, int(A.nnz))
        nnz0 = [None] * world
        dist.all_gather_object(nnz0, int(A0.nnz))
        assert max(nnz_all) <= (1.25 if star else 1.03) * sum(nnz_all) / world, ("partition_by_nnz left an imbalance", nnz_all, nnz0)
        assert sum(nnz_all) == int(Ag.nnz) and part[0] == 0 and part[-1] == n_global
    else:
        n_global = dims[0] * dims[1] * dims[2]
        part = gdist.row_partition(n_global, world)
        # the global matrix (for checking) and this rank's slab
        Ag = CSR(); h.gcge_problem_lap3d_box(dims[0], dims[1], dims[2], C.c_int64(0), C.c_int64(-1), C.byref(Ag))
        S = csr_to_scipy(Ag)
        A = CSR(); h.gcge_problem_lap3d_box(dims[0], dims[1], dims[2], C.c_int64(part[rank]), C.c_int64(part[rank + 1]), C.byref(A))
    n_loc = part[rank + 1] - part[rank]
    X = uniform(5, (n_global, 6)) - 0.5
    Yref = S @ X

    if mode == "oracle":
        import pyoracle as po
        from helpers import OracleBackend
        be = OracleBackend()
        comm = gdist.Comm(dist, rank, world, device=None)
        comm.install()
        ghosts = gdist.localize_slab(A)
        send_rows, send_cnt, recv_cnt = comm.plan_halo(ghosts, part)
        cb, sp, rp = comm.make_exchange(send_cnt, recv_cnt, 1)

        class Halo(C.Structure):
            _fields_ = [("nsend", C.c_int), ("send_rows", C.POINTER(C.c_int)), ("exchange", C.c_void_p), ("ctx", C.c_void_p)]
        halo = Halo(int(send_rows.size), send_rows.ctypes.data_as(C.POINTER(C.c_int)), C.cast(cb, C.c_void_p), None)
        po.oracle_lib().oracle_set_halo(C.byref(halo))
        po.oracle_lib().oracle_set_partition.argtypes = [C.c_long, C.c_long]
        po.oracle_lib().oracle_set_partition(part[rank], n_global)
        mat = be.matrix(A)             # ORACLE_CCS view: nrows = n_loc, ncols = n_loc + nghost
    elif mode == "hip_native":
        # every rank on its OWN device, RCCL called from C inside the back-end (csrc/hip/rccl_comm.hip): the production path of
        # bench.py --gpus N.  gloo only hands rank 0's communicator id round.
        from gcge_amd import HipBackend
        torch.cuda.set_device(rank)
        be = HipBackend(device=rank)
        if star or ball:
            be.g.gcge_hip_spmm_dense_mode.argtypes = [C.c_int]
            be.g.gcge_hip_spmm_dense_mode(1)          # small atoms: rows of >= 24 entries may seed a block
        comm = gdist.NativeComm(be, dist, rank, world)
        mat = comm.slab_matrix(A, part, cap_cols=64 if (star or ball) else 8, geometry=geometry)
        be.set_random_mode(1, 777)
    else:
        from gcge_amd import HipBackend
        be = HipBackend(device=0)
        if star or ball:
            be.g.gcge_hip_spmm_dense_mode.argtypes = [C.c_int]
            be.g.gcge_hip_spmm_dense_mode(1)          # small atoms: rows of >= 24 entries may seed a block
        comm = gdist.install(be, dist, rank, world, stage_through_host=True)
        mat = gdist.hip_slab_matrix(be, comm, A, n_global, part, cap_cols=64 if (star or ball) else 4, geometry=geometry)   # small cap: exercises the column chunking
        be.set_random_mode(1, 777)
        be.g.gcge_hip_mat_pattern_chain.argtypes = [C.c_void_p]
        assert sio2 or be.g.gcge_hip_mat_pattern_chain(mat) >= 1, "slab matrix with halo columns should keep the chain layout"

    if ball and mode in ("hip", "hip_native"):
        # every slab of the masked grid kept the plane sweep (third form through the line table: own and halo rows), and multiplies right
        g = be.g
        g.gcge_hip_mat_spmm_form.restype = C.c_char_p
        g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
        g.gcge_hip_mat_star_masked_form.argtypes = [C.c_void_p]
        g.gcge_hip_mat_star_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        g.gcge_hip_star_product_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        form = g.gcge_hip_mat_spmm_form(mat).decode()
        assert form.startswith("spmm_star+spmm_dense"), (rank, form)
        assert g.gcge_hip_mat_star_masked_form(mat) == 3, "a slab of a masked grid takes the third form of the sweep"
        st = (C.c_long * 8)()
        assert g.gcge_hip_mat_star_stats(mat, st) == 1 and tuple(st[:4]) == tuple(bdims) + (6,) and st[5] == n_loc, list(st)
        assert st[4] >= 0.7 * n_loc, ("most rows of the slab are star rows", list(st))
        plane = bdims[0] * bdims[1]
        assert (st[6], st[7]) == (int(box[part[rank]]) // plane, int(box[part[rank + 1] - 1]) // plane + 1), (list(st), part)
        Xw = uniform(6, (n_global, 66)) - 0.5
        Yw = S @ Xw
        xw = be.mv_from_numpy(mat, Xw[part[rank]:part[rank + 1], :])
        yw = be.ops.mv_create(66, mat)
        g.gcge_hip_spmm_dot2_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        g.gcge_hip_set_halo_overlap.argtypes = [C.c_int]
        for overlap in (0, 1):
            # (overlap: the planes at least 6 away from a plane that holds halo lines — a neighbour's planes, or the slab's own first /
            #  last plane when the cut runs through it — are swept while the halo travels, the others after the exchange)
            g.gcge_hip_set_halo_overlap(overlap)
            p0, s0_ = C.c_long(), C.c_long()
            g.gcge_hip_star_product_stats(C.byref(p0), C.byref(s0_))
            for m, a, b in [(64, 0, 0), (16, 2, 4), (17, 1, 0), (30, 3, 2), (66, 0, 0), (2, 8, 0)]:
                be.ops.spmm(mat, xw, yw, (a, b), (a + m, b + m))
                got = be.mv_to_numpy(yw, n_loc, b, b + m)
                err = np.max(np.abs(got - Yw[part[rank]:part[rank + 1], a:a + m]))
                assert err < 1e-12, "sweep on a slab of a masked grid (overlap=%d, m=%d, columns %d -> %d) differs: %g" % (overlap, m, a, b, err)
            for m, a, b in [(64, 0, 0), (30, 4, 2)]:
                dots, yy = np.zeros(m), np.zeros(m)
                g.gcge_hip_spmm_dot2_mv(mat, xw, yw, (C.c_int * 2)(a, b), (C.c_int * 2)(a + m, b + m), dots.ctypes.data, yy.ctypes.data, be.ops_handle)
                Yl, Xl = Yw[part[rank]:part[rank + 1], a:a + m], Xw[part[rank]:part[rank + 1], a:a + m]
                assert np.max(np.abs(be.mv_to_numpy(yw, n_loc, b, b + m) - Yl)) < 1e-12
                assert np.allclose(dots, (Xl * Yl).sum(0), rtol=1e-11, atol=1e-9) and np.allclose(yy, (Yl * Yl).sum(0), rtol=1e-11), (overlap, m)
            p1, s1_ = C.c_long(), C.c_long()
            g.gcge_hip_star_product_stats(C.byref(p1), C.byref(s1_))
            assert world == 1 or p1.value - p0.value >= 7, "the products did not go through the grid form"   # (one rank: no exchange, the whole-matrix path — not counted)
            inner = (st[7] - st[6]) - 7 * ((rank > 0) + (rank < world - 1))       # planes that surely need no halo row
            if world > 1 and overlap and inner > 0:
                assert s1_.value - s0_.value >= 5, "no product swept its interior planes while the halo travelled"
            if not overlap:
                assert s1_.value == s0_.value
        be.ops.mv_destroy(xw, 66); be.ops.mv_destroy(yw, 66)
    if star and mode in ("hip", "hip_native"):
        # every slab took the grid form, sits where the partition put it, and sweeps its inner planes while the halo travels
        g = be.g
        g.gcge_hip_mat_spmm_form.restype = C.c_char_p
        g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
        g.gcge_hip_mat_star_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        g.gcge_hip_star_product_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        form = g.gcge_hip_mat_spmm_form(mat).decode()
        assert form.startswith("spmm_star+spmm_dense"), (rank, form)
        st = (C.c_long * 8)()
        plane = sio2 * sio2
        assert g.gcge_hip_mat_star_stats(mat, st) == 1 and tuple(st[:4]) == (sio2, sio2, sio2, 6) and st[5] == n_loc, list(st)
        assert (st[6], st[7]) == (part[rank] // plane, part[rank + 1] // plane), (list(st), part)
        Xw = uniform(6, (n_global, 66)) - 0.5
        Yw = S @ Xw
        xw = be.mv_from_numpy(mat, Xw[part[rank]:part[rank + 1], :])
        yw = be.ops.mv_create(66, mat)
        g.gcge_hip_set_halo_overlap.argtypes = [C.c_int]
        g.gcge_hip_spmm_dot2_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        for overlap in (0, 1):
            g.gcge_hip_set_halo_overlap(overlap)
            p0, s0_ = C.c_long(), C.c_long()
            g.gcge_hip_star_product_stats(C.byref(p0), C.byref(s0_))
            for m, a, b in [(64, 0, 0), (16, 2, 4), (17, 1, 0), (30, 3, 2), (66, 0, 0), (2, 8, 0)]:
                be.ops.spmm(mat, xw, yw, (a, b), (a + m, b + m))
                got = be.mv_to_numpy(yw, n_loc, b, b + m)
                err = np.max(np.abs(got - Yw[part[rank]:part[rank + 1], a:a + m]))
                assert err < 1e-12, "star sweep on a slab (overlap=%d, m=%d, columns %d -> %d) differs: %g" % (overlap, m, a, b, err)
            for m, a, b in [(64, 0, 0), (30, 4, 2)]:          # the product with its column sums (LOCAL parts), as the fused CG asks for it
                dots, yy = np.zeros(m), np.zeros(m)
                g.gcge_hip_spmm_dot2_mv(mat, xw, yw, (C.c_int * 2)(a, b), (C.c_int * 2)(a + m, b + m), dots.ctypes.data, yy.ctypes.data, be.ops_handle)
                Yl, Xl = Yw[part[rank]:part[rank + 1], a:a + m], Xw[part[rank]:part[rank + 1], a:a + m]
                assert np.max(np.abs(be.mv_to_numpy(yw, n_loc, b, b + m) - Yl)) < 1e-12
                assert np.allclose(dots, (Xl * Yl).sum(0), rtol=1e-11, atol=1e-9) and np.allclose(yy, (Yl * Yl).sum(0), rtol=1e-11), (overlap, m)
            p1, s1_ = C.c_long(), C.c_long()
            g.gcge_hip_star_product_stats(C.byref(p1), C.byref(s1_))
            assert p1.value - p0.value >= 7, "the products did not go through the grid form"
            inner = (st[7] - st[6]) - 6 * ((st[6] > 0) + (st[7] < sio2))       # planes that need no halo row
            if overlap and inner > 0:
                assert s1_.value - s0_.value >= 5, "no product swept its interior planes while the halo travelled"
            if not overlap:
                assert s1_.value == s0_.value
        be.ops.mv_destroy(xw, 66); be.ops.mv_destroy(yw, 66)
    # 1. distributed SpMM == rows of the global product
    x = be.mv_from_numpy(mat, X[part[rank]:part[rank + 1], :])
    y = be.ops.mv_create(6, mat)
    be.ops.spmm(mat, x, y, (1, 0), (6, 5))
    got = be.mv_to_numpy(y, n_loc, 0, 5)
    err = np.max(np.abs(got - Yref[part[rank]:part[rank + 1], 1:6]))
    assert err < 1e-13, "distributed SpMM differs: %g" % err
    # 2. global inner product == numpy on the full vectors
    ip = be.ops.inner_prod("N", x, x, (0, 1), (3, 5))
    assert np.max(np.abs(ip - X[:, 0:3].T @ X[:, 1:5])) < 1e-14 * n_global      # (sums of n_global products of magnitude <= 1/4)
    # 3. whole eigensolve, SPMD
    if mode in ("hip", "hip_native"):
        be.g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
        be.g.gcge_hip_bpcg_setup(be.ops_handle, 30, 1e-2, 1e-14, b"abs")
    ev, res = run_gcg(be.ops_handle, mat, None, ["-nevConv", 8, "-gcge_compW_orth_method", os.environ.get("GCGE_TEST_ORTH", "chol")], flag=1 if mode in ("hip", "hip_native") else 0)
    if sio2:
        import scipy.sparse.linalg as sla
        exact = np.sort(sla.eigsh(S.tocsc(), k=res.nevConv, sigma=0.0, which="LM", return_eigenvectors=False))
    else:
        exact = box_exact(dims, res.nevConv)
    rel = np.max(np.abs(ev[:res.nevConv] - exact) / exact)
    assert res.nevConv >= 8 and rel < 1e-10, (res.nevConv, res.numIter, rel, list(ev[:10]))
    note = ""
    if mode in ("hip", "hip_native") and os.environ.get("GCGE_TEST_AMG"):
        # 3b. BlockAMG across ranks (round 5): every rank coarsens its own slab (whole planes, even cuts), the coarse slabs get their
        # own halo plans through the slab constructor of THIS transport, the fused CG smooths every level with its sums reduced over
        # the ranks, transfers are local.  Same Ritz values, no more outer iterations than the plain solver.
        if mode == "hip":
            gdist.install_slab_factory(be, comm)
        levels = int(os.environ["GCGE_TEST_AMG"])
        ev_a, res_a = run_gcg(be.ops_handle, mat, None, ["-nevConv", 8, "-gcge_compW_orth_method", "chol", "-gcge_amg_levels", levels], flag=1)
        ex_a = exact if res_a.nevConv <= res.nevConv else (box_exact(dims, res_a.nevConv) if not sio2 else None)
        k_a = min(res_a.nevConv, len(ex_a))
        rel_a = np.max(np.abs(ev_a[:k_a] - ex_a[:k_a]) / ex_a[:k_a])
        assert res_a.nevConv >= 8 and rel_a < 1e-10, ("BlockAMG across ranks", res_a.nevConv, res_a.numIter, rel_a)
        # (the SiO2-like matrix with the reference's smoothing counts needs MORE outer iterations than 30 plain CG steps: its coarse
        #  levels carry the projected atoms and want more smoothing; the Ritz values are what is pinned)
        assert res_a.numIter <= (2 * res.numIter if sio2 else res.numIter + 3), (res_a.numIter, res.numIter)
        note += " amg(%d levels): nevConv=%d numIter=%d rel=%.2e" % (levels, res_a.nevConv, res_a.numIter, rel_a)
    if mode in ("hip", "hip_native") and not sio2:
        # 4. the REFERENCE's compiled GCG / orthonormalisation (oracle/_ref, its own OPS_Setup, flag 1 = the back-end's
        # solver) over a table only OPS_HIP_Set has touched, on the same slab matrices: OPS_HIP_Set installs
        # MultiVecInnerProd / MultiVecQtAP with the all-reduce, so no line of the reference's src/ is edited (its own
        # default would not reduce without OPS_USE_MPI, src/ops_multi_vec.c:202-230)
        import pyoracle as po
        ref = po.ref_lib()
        if ref is not None:
            ops2 = C.c_void_p()
            be.h.OPS_Create(C.byref(ops2))
            be.g.OPS_HIP_Set(ops2)
            be.g.gcge_hip_bpcg_setup(ops2, 30, 1e-2, 1e-14, b"abs")
            be.set_random_mode(1, 778)
            ref.ref_gcg_solve_foreign.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double),
                                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
            ev2 = np.zeros(16)
            conv2, it2, sec2 = C.c_int(), C.c_int(), C.c_double()
            rc = ref.ref_gcg_solve_foreign(ops2, mat, None, 8, 16, 0, 0, 1e-1, 1e-8, 500, 1, ev2.ctypes.data_as(C.POINTER(C.c_double)),
                                           C.byref(conv2), C.byref(it2), C.byref(sec2))
            assert rc == 0 and conv2.value >= 8, (rc, conv2.value)
            rel2 = np.max(np.abs(ev2[:conv2.value] - box_exact(dims, conv2.value)) / box_exact(dims, conv2.value))
            assert rel2 < 1e-10, ("reference stack over the HIP slots, %d ranks" % world, rel2, list(ev2[:conv2.value]))
            note = " refstack: nevConv=%d numIter=%d rel=%.2e" % (conv2.value, it2.value, rel2)
            # 5. flag 0 — the reference's OWN BlockPCG over the slots — across ranks with a NON-MPI build of the reference: its
            # dots go through MultiVecLocalInnerProd and are summed by MPI_Allreduce under OPS_USE_MPI only
            # (src/ops_lin_sol.c:306-321,355-369); GCGE_SetLocalInnerProdReduces(1) makes the back-end's local slot return the
            # sum over the ranks.  And our own BlockPCG with the switch on: it must not reduce a second time.
            be.h.GCGE_SetLocalInnerProdReduces(1)
            assert be.h.GCGE_GetLocalInnerProdReduces() == (1 if world > 1 else 0)     # (a world of one rank keeps no communicator)
            try:
                ops3 = C.c_void_p()
                be.h.OPS_Create(C.byref(ops3))
                be.g.OPS_HIP_Set(ops3)
                be.set_random_mode(1, 779)
                a0 = comm.n_allreduce
                rc = ref.ref_gcg_solve_foreign(ops3, mat, None, 8, 16, 0, 0, 1e-1, 1e-8, 500, 0, ev2.ctypes.data_as(C.POINTER(C.c_double)),
                                               C.byref(conv2), C.byref(it2), C.byref(sec2))
                assert rc == 0 and conv2.value >= 8, (rc, conv2.value)
                ex3 = box_exact(dims, conv2.value)
                rel3 = np.max(np.abs(ev2[:conv2.value] - ex3) / ex3)
                assert rel3 < 1e-10, ("reference stack with ITS BlockPCG (flag 0) over the HIP slots, %d ranks" % world, rel3)
                assert world == 1 or comm.n_allreduce - a0 > 4 * it2.value, "the reference's BlockPCG did not reduce through the local slot"
                ev4, res4 = run_gcg(be.ops_handle, mat, None, ["-nevConv", 8], flag=0)      # our MGS + our BlockPCG, switch on
                ex4 = box_exact(dims, res4.nevConv)
                assert res4.nevConv >= 8 and np.max(np.abs(ev4[:res4.nevConv] - ex4) / ex4) < 1e-10
                note += " flag0: nevConv=%d numIter=%d rel=%.2e (ours: %d its)" % (conv2.value, it2.value, rel3, res4.numIter)
            finally:
                be.h.GCGE_SetLocalInnerProdReduces(0)
    allc = [None] * world
    dist.all_gather_object(allc, (res.nevConv, res.numIter, float(ev[0])))
    assert all(a[:2] == allc[0][:2] for a in allc), "ranks disagree: %r" % (allc,)
    print("rank %d ok: nevConv=%d numIter=%d rel=%.2e allreduces=%d%s" % (rank, res.nevConv, res.numIter, rel, comm.n_allreduce, note))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
