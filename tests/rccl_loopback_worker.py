"""Single-GPU rehearsal of the production transport (backend "nccl" == RCCL) — worker of
tests/test_dist.py::test_rccl_loopback_one_gpu.

Two ranks cannot share a GPU under RCCL, so the 2-rank tests run over gloo.  What they cannot see is the part of
gcge_amd/dist.py that only exists for device buffers on the real transport: P2POps on device tensors posted from inside
the ctypes callback, the split begin / end form with the interior rows multiplied in between, stream ordering between
the back-end's stream and RCCL's, and the device round trip of the small all-reduce.  Here a world of ONE rank runs
exactly that code against itself: rank 0 plays the lower slab of a 2-slab Laplacian and its "neighbour" is itself
(RCCL allows send/recv to the own rank inside one group), so the ghost plane it receives is its own last plane.  The
operator this defines is known in closed form: the local block with the diagonal of the last plane reduced by one.

argv[2] == "native": the production path — RCCL called from C inside libgcge_hip.so (csrc/hip/rccl_comm.hip:
gcge_hip_comm_init, gcge_hip_mat_set_halo_rccl with both virtual slabs mapped to rank 0, all-reduces through
GCGE_COMM -> ncclAllReduce); torch.distributed is not initialised at all.  Otherwise: the torch.distributed callbacks.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def star_loopback(G, ball=False):
    """argv[2] == "native_star": slab 0 of a TWO-slab SiO2-like matrix (cut on a plane boundary, inside atom blocks) whose
    neighbour is this rank itself, over RCCL from C: halo row with global id g (a row of slab 1) is served by own row
    g - n_loc, i.e. the planes above the slab are the slab's own first planes.  The operator this defines is
    S[:, own] + S[:, halo] P with P the 0/1 matrix of that map — checked against scipy: the plane sweep on a slab reading its
    upper z-neighbours from halo rows that arrive by grouped ncclSend/ncclRecv on the transfer stream while the inner planes
    are swept, then the boundary planes, the blocks and the listed rows; plain products, odd ranges, products with sums."""
    os.environ["GCGE_COMM_KEEP_SINGLE"] = "1"
    import scipy.sparse as sp
    import torch
    torch.cuda.set_device(0)
    from gcge_amd import HipBackend
    from gcge_amd import dist as gdist
    from gcge_amd.lib import make_problem
    from helpers import csr_to_scipy, uniform
    kw = dict(K=8, R0=1.5, R1=3.0, seed=12345)
    plane, n_global = G * G, G ** 3
    n_loc = (G // 2) * plane
    if ball:                   # "native_ball": the same on the BALL inside the box (a masked grid), the cut between two grid lines
        from gcge_amd.lib import ball_geometry
        box = ball_geometry(G)
        n_global = int(box.size)
        n_loc = gdist.partition_lines(box, G, 2)[1]
    be = HipBackend(device=0)
    g = be.g
    g.gcge_hip_spmm_dense_mode.argtypes = [C.c_int]
    g.gcge_hip_spmm_dense_mode(1)
    comm = gdist.NativeComm(be, None, 0, 1)
    A, _ = make_problem("sio2ball" if ball else "sio2", G, row_begin=0, row_end=n_loc, **kw)
    Sg = csr_to_scipy(A).tocsr()                               # n_loc x n_global, global columns
    ghosts = np.ascontiguousarray(gdist.localize_slab(A), dtype=np.int32)
    ng = int(ghosts.size)
    assert (ball or ng >= 6 * plane) and ghosts[0] == n_loc and np.all(ghosts - n_loc < n_loc)
    if ball:
        box_local = np.ascontiguousarray(np.concatenate([box[:n_loc], box[ghosts]]), dtype=np.int32)
        g.gcge_hip_star_next_geometry_cols.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        g.gcge_hip_star_next_geometry_cols(int(box_local.size), G, G, G, box_local.ctypes.data_as(C.POINTER(C.c_int)))
    P = sp.csr_matrix((np.ones(ng), (np.arange(ng), ghosts - n_loc)), shape=(ng, n_loc))
    S = (Sg[:, :n_loc] + Sg[:, ghosts] @ P).tocsr()
    ip_ = C.POINTER(C.c_int)
    g.gcge_hip_mat_create_local_ghosts.restype = C.c_void_p
    g.gcge_hip_mat_create_local_ghosts.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip_, ip_, C.POINTER(C.c_double), ip_]
    mat = C.c_void_p(g.gcge_hip_mat_create_local_ghosts(A.nrows, A.ncols, n_global, 0, A.rowptr, A.colidx, A.val, ghosts.ctypes.data_as(ip_)))
    send_rows = np.ascontiguousarray(ghosts - n_loc, dtype=np.int32)
    peer, scnt, rcnt = (C.c_int * 2)(0, 0), (C.c_int * 2)(0, ng), (C.c_int * 2)(0, ng)
    g.gcge_hip_mat_set_halo_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, ip_, ip_, ip_, ip_, C.c_int]
    assert g.gcge_hip_mat_set_halo_rccl(mat, n_global, 2, peer, scnt, rcnt, send_rows.ctypes.data_as(ip_), 64) == 0
    g.gcge_hip_mat_spmm_form.restype = C.c_char_p
    g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
    form = g.gcge_hip_mat_spmm_form(mat).decode()
    assert form.startswith("spmm_star+spmm_dense"), form
    if ball:
        g.gcge_hip_mat_star_masked_form.argtypes = [C.c_void_p]
        assert g.gcge_hip_mat_star_masked_form(mat) == 3
    g.gcge_hip_star_product_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
    g.gcge_hip_set_halo_overlap.argtypes = [C.c_int]
    g.gcge_hip_spmm_dot2_mv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, ip_, ip_, C.c_void_p, C.c_void_p, C.c_void_p]
    X = uniform(5, (n_loc, 66)) - 0.5
    y = be.ops.mv_create(66, mat)
    nsplit = []
    for overlap in (0, 1):
        g.gcge_hip_set_halo_overlap(overlap)
        p0, s0 = C.c_long(), C.c_long()
        g.gcge_hip_star_product_stats(C.byref(p0), C.byref(s0))
        for rep in range(3):              # repeated with other operands: a missing stream dependency shows as stale halo planes
            Xr = X * (1.0 + rep)
            x = be.mv_from_numpy(mat, Xr)
            Yr = S @ Xr
            for m, a, b in [(64, 0, 0), (16, 2, 4), (17, 1, 0), (66, 0, 0)]:
                be.ops.spmm(mat, x, y, (a, b), (a + m, b + m))
                err = np.max(np.abs(be.mv_to_numpy(y, n_loc, b, b + m) - Yr[:, a:a + m]))
                assert err < 1e-12, "loop-back star sweep (overlap=%d, rep=%d, m=%d) differs: %g" % (overlap, rep, m, err)
            m = 64
            dots, yy = np.zeros(m), np.zeros(m)
            g.gcge_hip_spmm_dot2_mv(mat, x, y, (C.c_int * 2)(0, 0), (C.c_int * 2)(m, m), dots.ctypes.data, yy.ctypes.data, be.ops_handle)
            assert np.max(np.abs(be.mv_to_numpy(y, n_loc, 0, m) - Yr[:, :m])) < 1e-12
            assert np.allclose(dots, (Xr[:, :m] * Yr[:, :m]).sum(0), rtol=1e-11, atol=1e-9) and np.allclose(yy, (Yr[:, :m] ** 2).sum(0), rtol=1e-11)
            be.ops.mv_destroy(x, 66)
        p1, s1 = C.c_long(), C.c_long()
        g.gcge_hip_star_product_stats(C.byref(p1), C.byref(s1))
        assert p1.value - p0.value >= 15
        nsplit.append(s1.value - s0.value)
    assert nsplit[0] == 0 and (nsplit[1] >= 12 or ball), nsplit           # (a slab of a masked grid sweeps once, after the exchange)
    # the fused CG with MORE right-hand sides than the halo buffers hold on a star matrix (stored-product form): 8-column
    # buffers, 16 columns — the device-scalar loop must decline (gcge_hip_spmm_dot2_dev_ok) and the host-scalar loop chunk the
    # columns; it used to abort.  The loop-back operator is not symmetric, so this is no solve: five iterations of the same
    # recurrences against the same loop with 64-column buffers (one chunk), column for column.
    g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
    g.gcge_hip_bpcg_release.argtypes = [C.c_void_p]
    nr = 16
    Bm = uniform(41, (n_loc, nr)) - 0.5
    sols = []
    for cap in (64, 8):
        assert g.gcge_hip_mat_set_halo_rccl(mat, n_global, 2, peer, scnt, rcnt, send_rows.ctypes.data_as(ip_), cap) == 0
        g.gcge_hip_bpcg_setup(be.ops_handle, 5, 1e-30, 1e-30, b"abs")
        vb = be.mv_from_numpy(mat, Bm)
        vx = be.mv_from_numpy(mat, np.zeros_like(Bm))
        be.ops.multi_linear_solver(mat, vb, vx, (0, 0), (nr, nr))
        sols.append(be.mv_to_numpy(vx, n_loc, 0, nr))
        be.ops.mv_destroy(vb, nr); be.ops.mv_destroy(vx, nr)
        g.gcge_hip_bpcg_release(be.ops_handle)
    assert np.all(np.isfinite(sols[1])) and np.max(np.abs(sols[0] - sols[1])) <= 1e-9 * np.max(np.abs(sols[0])), np.max(np.abs(sols[0] - sols[1]))
    be.free_matrix(mat)
    comm.finalize()
    print("rccl loop-back ok: star sweep on a slab of %s, %d rows, %d halo rows, %d products with the interior swept while the halo travelled"
          % ("the ball in %d^3" % G if ball else "%d planes of %d^2" % (G // 2, G), n_loc, ng, nsplit[1]))


def star_loopback_full_size(G, K):
    """argv[2] == "native_star_full": BOTH slabs of a two-way plane-aligned cut of BASELINE config 5's matrix at full size, each
    with its halo looped back onto this rank over RCCL from C — through a size-independent property: with X = 1 everywhere (halo
    rows included, whoever serves them) every row of A X is the row sum of the slab's CSR row.  Any wrong halo-plane address of the
    sweep, at any size of index, shows in some row."""
    os.environ["GCGE_COMM_KEEP_SINGLE"] = "1"
    import torch
    torch.cuda.set_device(0)
    from gcge_amd import HipBackend
    from gcge_amd import dist as gdist
    from gcge_amd.lib import make_problem
    kw = dict(K=K, R0=2.0, R1=5.0, seed=12345)
    plane, n = G * G, G ** 3
    be = HipBackend(device=0)
    g = be.g
    comm = gdist.NativeComm(be, None, 0, 1)
    ip_ = C.POINTER(C.c_int)
    g.gcge_hip_mat_create_local_ghosts.restype = C.c_void_p
    g.gcge_hip_mat_create_local_ghosts.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip_, ip_, C.POINTER(C.c_double), ip_]
    g.gcge_hip_mat_set_halo_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, ip_, ip_, ip_, ip_, C.c_int]
    g.gcge_hip_mat_spmm_form.restype = C.c_char_p
    g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
    g.gcge_hip_star_product_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
    part = [0, (G // 2) * plane, n]
    for r in range(2):
        A, _ = make_problem("sio2", G, row_begin=part[r], row_end=part[r + 1], **kw)
        nloc, nnz = A.nrows, int(A.nnz)
        rp = np.ctypeslib.as_array(A.rowptr, shape=(nloc + 1,)).astype(np.int64)
        va = np.ctypeslib.as_array(A.val, shape=(nnz,))
        rowsum = np.add.reduceat(va, rp[:-1]); absum = np.add.reduceat(np.abs(va), rp[:-1])
        ghosts = np.ascontiguousarray(gdist.localize_slab(A), dtype=np.int32)
        ng = int(ghosts.size)
        own = np.where(ghosts < part[r], ghosts - part[r] + nloc, ghosts - part[r + 1]).astype(np.int32)   # halo row -> an own row
        assert own.min() >= 0 and own.max() < nloc
        mat = C.c_void_p(g.gcge_hip_mat_create_local_ghosts(nloc, A.ncols, n, part[r], A.rowptr, A.colidx, A.val, ghosts.ctypes.data_as(ip_)))
        peer, scnt, rcnt = (C.c_int * 2)(0, 0), (C.c_int * 2)(0, ng), (C.c_int * 2)(0, ng)
        assert g.gcge_hip_mat_set_halo_rccl(mat, n, 2, peer, scnt, rcnt, np.ascontiguousarray(own).ctypes.data_as(ip_), 64) == 0
        assert g.gcge_hip_mat_spmm_form(mat).decode().startswith("spmm_star+spmm_dense"), g.gcge_hip_mat_spmm_form(mat).decode()
        x = be.mv_from_numpy(mat, np.ones((nloc, 16)))
        y = be.ops.mv_create(16, mat)
        p0, s0 = C.c_long(), C.c_long(); g.gcge_hip_star_product_stats(C.byref(p0), C.byref(s0))
        be.ops.spmm(mat, x, y, (0, 0), (16, 16))
        p1, s1 = C.c_long(), C.c_long(); g.gcge_hip_star_product_stats(C.byref(p1), C.byref(s1))
        assert s1.value == s0.value + 1, "the product did not sweep its inner planes while the halo travelled"
        got = be.mv_to_numpy(y, nloc, 0, 16)
        err = float(np.max(np.abs(got[:, 0] - rowsum) / absum))
        assert err < 1e-13 and np.array_equal(got[:, 0], got[:, 15]), (r, err)
        be.ops.mv_destroy(x, 16); be.ops.mv_destroy(y, 16)
        be.free_matrix(mat)
        print("slab %d: %d rows, %d halo rows, row sums to %.1e" % (r, nloc, ng, err), flush=True)
    comm.finalize()
    print("rccl loop-back ok: full-size slabs of the %d^3 matrix, row sums of every row through the sweep with looped-back halo planes" % G)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "native_ball":
        star_loopback(int(sys.argv[1]), ball=True)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "native_star":
        return star_loopback(int(sys.argv[1]))
    if len(sys.argv) > 2 and sys.argv[2] == "native_star_full":
        return star_loopback_full_size(int(sys.argv[1]), int(sys.argv[3]) if len(sys.argv) > 3 else 2000)
    dims = tuple(int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "8,8,10").split(","))
    native = len(sys.argv) > 2 and sys.argv[2] == "native"
    os.environ["GCGE_COMM_KEEP_SINGLE"] = "1"      # a world of one rank: keep the all-reduces on the transport
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if not native:
        dist.init_process_group("nccl", rank=0, world_size=1)
    from gcge_amd import HipBackend
    from gcge_amd import dist as gdist
    from gcge_amd.lib import CSR, host_lib, run_gcg
    from helpers import csr_to_scipy, uniform
    h = host_lib()
    nx, ny, nz = dims
    plane, n_global = nx * ny, nx * ny * nz
    part = gdist.row_partition(n_global, 2)          # virtual world of two slabs; this process is slab 0
    n_loc = part[1]
    assert n_loc % plane == 0

    class LoopbackComm(gdist.Comm):
        def _peer(self, q):
            return 0

        def plan_halo(self, ghosts, part_):
            gh = np.asarray(ghosts, dtype=np.int64)
            assert gh.size == plane and gh[0] == n_loc, "ghosts of slab 0 = first plane of slab 1"
            send_rows = (gh - plane).astype(np.int32)         # what slab 1 needs of slab 0: the plane below, same order
            return np.ascontiguousarray(send_rows), [0, plane], [0, plane]

    be = HipBackend(device=0)
    if native:
        comm = gdist.NativeComm(be, None, 0, 1)
    else:
        comm = LoopbackComm(dist, 0, 1, device=torch.device("cuda", 0))
        comm.install()
    A = CSR(); h.gcge_problem_lap3d_box(nx, ny, nz, C.c_int64(0), C.c_int64(n_loc), C.byref(A))
    S = csr_to_scipy(A).tocsr()[:, :n_loc].tolil()              # global columns -> local block
    for r in range(n_loc - plane, n_loc):
        S[r, r] -= 1.0                                        # ghost row (i,j,nz/2) == own row (i,j,nz/2-1), coefficient -1
    S = S.tocsr()
    if native:
        ghosts = gdist.localize_slab(A)
        assert ghosts.size == plane and ghosts[0] == n_loc
        mat = be.matrix(A)
        ip_ = C.POINTER(C.c_int)
        send_rows = np.ascontiguousarray(ghosts - plane, dtype=np.int32)       # slab 1 wants the plane below its first one
        peer, scnt, rcnt = (C.c_int * 2)(0, 0), (C.c_int * 2)(0, plane), (C.c_int * 2)(0, plane)
        be.g.gcge_hip_mat_set_halo_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, ip_, ip_, ip_, ip_, C.c_int]
        rc = be.g.gcge_hip_mat_set_halo_rccl(mat, n_global, 2, peer, scnt, rcnt, send_rows.ctypes.data_as(ip_), 64)
        assert rc == 0, rc
    else:
        mat = gdist.hip_slab_matrix(be, comm, A, n_global, part, cap_cols=64)
    be.set_random_mode(1, 777)

    # 1. SpMM through both forms of the exchange (single call / split with the interior rows multiplied in between)
    m = 24
    X = uniform(5, (n_loc, m)) - 0.5
    Yref = S @ X
    x = be.mv_from_numpy(mat, X)
    y = be.ops.mv_create(m, mat)
    be.g.gcge_hip_set_halo_overlap.argtypes = [C.c_int]
    for overlap in (0, 1):
        be.g.gcge_hip_set_halo_overlap(overlap)
        for rep in range(3):          # repeated: a missing stream dependency shows as a stale ghost plane
            Xr = X * (1.0 + rep)
            x = be.mv_from_numpy(mat, Xr)
            be.ops.spmm(mat, x, y, (0, 0), (m, m))
            got = be.mv_to_numpy(y, n_loc, 0, m)
            err = np.max(np.abs(got - (1.0 + rep) * Yref))
            assert err < 1e-12, "loop-back SpMM (overlap=%d, rep=%d) differs: %g" % (overlap, rep, err)
    be.g.gcge_hip_set_halo_overlap(1)
    # 2. the small all-reduce over RCCL (world 1: identity, but through the device round trip)
    ip = be.ops.inner_prod("N", x, x, (0, 1), (3, 5))
    Xl = 3.0 * X
    assert np.max(np.abs(ip - Xl[:, 0:3].T @ Xl[:, 1:5])) < 1e-10
    # 3. whole eigensolve on the loop-back operator (fused device CG: SpMM + dots with the split exchange inside)
    be.g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
    be.g.gcge_hip_bpcg_setup(be.ops_handle, 30, 1e-2, 1e-14, b"abs")
    ev, res = run_gcg(be.ops_handle, mat, None, ["-nevConv", 8, "-gcge_compW_orth_method", "chol"], flag=1)
    if n_loc <= 4096:
        exact = np.linalg.eigvalsh(S.toarray())[:res.nevConv]
    else:
        import scipy.sparse.linalg as sla
        exact = np.sort(sla.eigsh(S, k=res.nevConv, sigma=0.0, which="LM", return_eigenvectors=False))
    rel = np.max(np.abs(ev[:res.nevConv] - exact) / exact)
    assert res.nevConv >= 8 and rel < 1e-10, (res.nevConv, res.numIter, rel, list(ev[:10]))
    note = ""
    if native:
        # 3b. ADVICE r4: a block WIDER than the halo buffers through the fused CG.  The plan is re-installed with 8-column buffers;
        # 16 right-hand sides then cannot take the one-call product with sums (gcge_hip_spmm_dot2_dev refuses m > buf_cols) nor the
        # recompute passes (gcge_hip_cg_fusable declines): the solver must fall back to the loop that chunks the columns — it used
        # to abort ("refused operands the one-pass scheme had accepted").  Against a direct solve of the loop-back operator.
        import scipy.sparse.linalg as sla
        rc = be.g.gcge_hip_mat_set_halo_rccl(mat, n_global, 2, peer, scnt, rcnt, send_rows.ctypes.data_as(ip_), 8)
        assert rc == 0, rc
        nr = 16
        Bm = uniform(31, (n_loc, nr)) - 0.5
        Xs = sla.spsolve(S.tocsc(), Bm)
        for env in ({}, {"GCGE_CG_NO_RECOMPUTE": "1"}):
            os.environ.update(env)
            try:
                be.g.gcge_hip_bpcg_setup(be.ops_handle, 400, 1e-12, 1e-14, b"abs")
                vb = be.mv_from_numpy(mat, Bm)
                vx = be.mv_from_numpy(mat, np.zeros_like(Bm))
                be.ops.multi_linear_solver(mat, vb, vx, (0, 0), (nr, nr))
                got = be.mv_to_numpy(vx, n_loc, 0, nr)
                errw = np.max(np.abs(got - Xs)) / np.max(np.abs(Xs))
                assert errw < 1e-8, ("fused CG, 16 right-hand sides over 8-column halo buffers", env, errw)
                be.ops.mv_destroy(vb, nr); be.ops.mv_destroy(vx, nr)
            finally:
                for k_ in env:
                    os.environ.pop(k_, None)
        be.g.gcge_hip_bpcg_release.argtypes = [C.c_void_p]
        be.g.gcge_hip_bpcg_release(be.ops_handle)
        rc = be.g.gcge_hip_mat_set_halo_rccl(mat, n_global, 2, peer, scnt, rcnt, send_rows.ctypes.data_as(ip_), 64)
        assert rc == 0, rc
        note += " wide-block CG over 8-column halo buffers ok"
    # 4. the REFERENCE's compiled stack (oracle/_ref: its GCG, its OPS_Setup, flag 1) over a table only OPS_HIP_Set has
    # touched, on the same loop-back matrix and communicator: the back-end's own MultiVecInnerProd / MultiVecQtAP reduce,
    # nothing of the reference's src/ is edited
    import pyoracle as po
    ref = po.ref_lib()
    if ref is not None:
        ops2 = C.c_void_p()
        be.h.OPS_Create(C.byref(ops2))
        be.g.OPS_HIP_Set(ops2)
        be.g.gcge_hip_bpcg_setup(ops2, 30, 1e-2, 1e-14, b"abs")
        ref.ref_gcg_solve_foreign.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double),
                                              C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        ev2 = np.zeros(16)
        conv2, it2, sec2 = C.c_int(), C.c_int(), C.c_double()
        a0 = comm.n_allreduce
        rc = ref.ref_gcg_solve_foreign(ops2, mat, None, 8, 16, 0, 0, 1e-1, 1e-8, 500, 1, ev2.ctypes.data_as(C.POINTER(C.c_double)),
                                       C.byref(conv2), C.byref(it2), C.byref(sec2))
        assert rc == 0 and conv2.value >= 8, (rc, conv2.value)
        k2 = conv2.value
        ex2 = np.linalg.eigvalsh(S.toarray())[:k2] if n_loc <= 4096 else np.sort(__import__("scipy.sparse.linalg", fromlist=["eigsh"]).eigsh(S, k=k2, sigma=0.0, which="LM", return_eigenvectors=False))
        rel2 = np.max(np.abs(ev2[:k2] - ex2) / ex2)
        assert rel2 < 1e-10, ("reference stack over the HIP slots through the communicator", rel2)
        assert comm.n_allreduce > a0, "the reference stack's inner products did not go through the communicator"
        note = " refstack nevConv=%d numIter=%d rel=%.2e" % (k2, it2.value, rel2)
    if native:
        # latency of the small all-reduce (host buffer in, host buffer out) through the C path
        import time
        x3 = be.mv_from_numpy(mat, X[:, :4])
        be.ops.inner_prod("D", x3, x3, (0, 0), (4, 4))
        t0 = time.perf_counter()
        for _ in range(200):
            be.ops.inner_prod("D", x3, x3, (0, 0), (4, 4))
        t_with = (time.perf_counter() - t0) / 200
        n0 = comm.n_allreduce
        be.h.GCGE_SetComm(None)
        t0 = time.perf_counter()
        for _ in range(200):
            be.ops.inner_prod("D", x3, x3, (0, 0), (4, 4))
        t_without = (time.perf_counter() - t0) / 200
        note += " allreduce_us=%.1f (dot with %.1f us, without %.1f us)" % (1e6 * (t_with - t_without), 1e6 * t_with, 1e6 * t_without)
        assert n0 > 0
        be.free_matrix(mat)
        comm.finalize()
    print("rccl loop-back ok: dims=%s nevConv=%d numIter=%d rel=%.2e allreduces=%d%s" % (dims, res.nevConv, res.numIter, rel, comm.n_allreduce if not native else n0, note))
    if not native:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
