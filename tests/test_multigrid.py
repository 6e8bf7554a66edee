"""The multigrid leg of the hot path: BlockAMG (csrc/host/lin_sol.c; reference src/ops_lin_sol.c:466-715), the
transfers DefaultMultiVecFromItoJ (csrc/host/ops_table.c; reference src/ops_multi_grid.c:69-117), the hierarchy behind
ops->MultiGridCreate (include/gcge_multigrid.h, csrc/host/multigrid.c, csrc/hip/multigrid.hip) and GCG with BlockAMG as the
solver of its W systems (reference test/test_eig_sol_SiO2_MAT.c:96-128,160-170, OPS_USE_AMG).

Pinned against tests/golden/amg.json: the compiled reference's BlockAMG / MultiVecFromItoJ / toy MultiGridCreate over its dense
back-end (tests/golden/make_golden_amg.py)."""
import ctypes as C

import numpy as np
import pytest

from gcge_amd.lib import host_lib, make_problem, run_gcg
from helpers import (DenseBackend, OracleBackend, block_amg_solve, csr_from_scipy, csr_to_scipy, lap3d_exact, load_golden,
                     mg_hierarchy, uniform)

G = load_golden("amg.json")


def F(a):
    return np.asfortranarray(a, dtype=np.float64)


def toy1d_dense():
    n0 = G["toy1d_hierarchy"]["n0"]
    return F(2.0 * np.eye(n0) - np.eye(n0, k=1) - np.eye(n0, k=-1))


def toy1d_levels():
    """(A_l, P_l) of the reference's toy hierarchy as numpy arrays, rebuilt from its definition (app_lapack.c:863-929)."""
    ns = G["toy1d_hierarchy"]["levels"]
    As, Ps = [toy1d_dense()], []
    for lev in range(len(ns) - 1):
        P = np.zeros((ns[lev], ns[lev + 1]))
        for c in range(ns[lev + 1]):
            P[2 * c + 1, c] = 1.0
            P[2 * c, c] = P[2 * c + 2, c] = 0.5
        Ps.append(P)
        As.append(P.T @ As[-1] @ P)
    return As, Ps


def slot_multigrid(backend, A_handle, B_handle, levels):
    """ops->MultiGridCreate of `backend`'s table: (A handles, P handles, destroy())."""
    from gcge_amd.ops_struct import OPS
    st = C.cast(backend.ops_handle, C.POINTER(OPS)).contents
    A_arr, B_arr, P_arr, nl = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(levels)
    create = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p)(st.MultiGridCreate)
    destroy = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p)(st.MultiGridDestroy)
    create(C.byref(A_arr), C.byref(B_arr), C.byref(P_arr), C.byref(nl), A_handle, B_handle, backend.ops_handle)
    L = nl.value
    Ah = [C.c_void_p(v) for v in C.cast(A_arr, C.POINTER(C.c_void_p * L)).contents]
    Ph = [C.c_void_p(v) for v in C.cast(P_arr, C.POINTER(C.c_void_p * max(1, L - 1))).contents][:L - 1]

    def done():
        destroy(C.byref(A_arr), C.byref(B_arr) if B_handle is not None else None, C.byref(P_arr), C.byref(nl), backend.ops_handle)
    return Ah, Ph, done


# ---------------------------------------------------------------------------------------------- host: the hierarchy itself
def test_dense_toy_multigrid_equals_the_reference():
    """D_MultiGridCreate (dense_host.c) restates app_lapack.c:863-929: same level sizes, same Galerkin matrices."""
    db = DenseBackend()
    A0 = toy1d_dense()
    Ah, Ph, done = slot_multigrid(db, db.matrix(A0), None, 3)
    from helpers import DenseMat
    assert len(Ah) == 3
    for lev in (1, 2):
        m = C.cast(Ah[lev], C.POINTER(DenseMat)).contents
        got = np.ctypeslib.as_array(m.data, shape=(m.ncols, m.ldd)).T[:m.nrows]
        want = np.array(G["toy1d_hierarchy"]["A"][lev - 1])
        assert got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-14 * np.max(np.abs(want))
    done()


@pytest.mark.parametrize("kind,size,dims,arm", [("lap3d", 6, (6, 6, 6), 1), ("fe3d", 7, (7, 7, 7), 1), ("sio2", 14, (14, 14, 14), 6),
                                                 ("fe1d", 40, (40, 1, 1), 1)])
def test_grid_detection(kind, size, dims, arm):
    h = host_lib()
    from gcge_amd.lib import CSR
    A, _ = make_problem(kind, size)
    d = (C.c_int * 3)()
    a = C.c_int()
    h.gcge_mg_detect_grid.argtypes = [C.POINTER(CSR), C.POINTER(C.c_int * 3), C.POINTER(C.c_int)]
    assert h.gcge_mg_detect_grid(C.byref(A), C.byref(d), C.byref(a)) == 1
    assert tuple(d) == dims and a.value == arm


def test_grid_detection_refuses_a_permuted_matrix():
    import scipy.sparse as sp
    h = host_lib()
    from gcge_amd.lib import CSR
    A, _ = make_problem("lap3d", 6)
    S = csr_to_scipy(A)
    p = np.random.default_rng(3).permutation(S.shape[0])
    Sp = sp.csr_matrix(S[p][:, p])
    Ap, keep = csr_from_scipy(Sp)
    d = (C.c_int * 3)()
    h.gcge_mg_detect_grid.argtypes = [C.POINTER(CSR), C.POINTER(C.c_int * 3), C.POINTER(C.c_int)]
    assert h.gcge_mg_detect_grid(C.byref(Ap), C.byref(d), None) == 0
    # ... and the hierarchy then comes from the greedy aggregation over the graph: still P^T A P, still a partition of the rows
    lev = mg_hierarchy(Ap, 3, scale=1.0, min_rows=4)
    assert len(lev["A"]) >= 2 and lev["dims"][0] == (0, 0, 0)
    P = lev["P"][0]
    assert P.shape[0] == 216 and np.all(np.asarray(P.sum(axis=1)).ravel() == 1.0) and P.shape[1] < 216 / 1.5
    assert abs(lev["A"][1] - (P.T @ Sp @ P)).max() < 1e-13


@pytest.mark.parametrize("kind,size", [("lap3d", 9), ("fe3d", 8), ("sio2", 12)])
def test_aggregation_hierarchy_is_galerkin(kind, size):
    """A_{l+1} = scale P_l^T A_l P_l with P_l the piecewise-constant prolongation of 2 x 2 x 2 cells; PT is its transpose; odd grid
    sizes end in a thinner last cell; B (mass matrix) is coarsened without the rescaling."""
    A, B = make_problem(kind, size)
    lev = mg_hierarchy(A, 4, scale=0.5, min_rows=4, B=B)
    S = csr_to_scipy(A)
    assert lev["dims"][0] == (size, size, size) and len(lev["A"]) >= 3
    for lvl in range(len(lev["P"])):
        P, PT = lev["P"][lvl], lev["PT"][lvl]
        cd = lev["dims"][lvl + 1]
        assert cd == tuple((d + 1) // 2 for d in lev["dims"][lvl]) and P.shape[1] == cd[0] * cd[1] * cd[2]
        assert abs(P.T - PT).max() == 0.0 and np.all(np.asarray(P.sum(axis=1)).ravel() == 1.0)
        want = 0.5 * (P.T @ lev["A"][lvl] @ P)
        assert abs(lev["A"][lvl + 1] - want).max() <= 1e-13 * abs(want).max()
        if B is not None:
            wb = P.T @ lev["B"][lvl] @ P
            assert abs(lev["B"][lvl + 1] - wb).max() <= 1e-13 * abs(wb).max()
    assert abs(lev["A"][0] - S).max() == 0.0
    if kind == "lap3d":       # a coarse 7-point Laplacian again: 2 x (6, -1) in the interior (4 x by Galerkin, halved)
        Ac = lev["A"][1].toarray()
        c = (size + 1) // 2
        mid = 1 + c * (1 + c * 1)
        assert Ac[mid, mid] == 12.0 and Ac[mid, mid + 1] == -2.0 and Ac[mid, mid + c] == -2.0 and Ac[mid, mid + c * c] == -2.0


@pytest.mark.parametrize("kind,dims,world", [("lap3d", (6, 6, 16), 2), ("lap3d", (5, 7, 24), 3), ("sio2", (12, 12, 12), 1), ("lap3d", (6, 6, 8), 4),
                                             ("sio2", (16, 16, 16), 2), ("sio2", (12, 12, 12), 3), ("lap3d", (6, 5, 19), 2), ("sio2", (19, 19, 19), 2)])
def test_slab_hierarchy_equals_the_rows_of_the_whole_hierarchy(kind, dims, world):
    """gcge_mg_build_slab: every rank coarsens its own slab (whole planes; here cut on even plane numbers) — stacked, the coarse slabs ARE
    the coarse matrix of the whole-matrix hierarchy at every level, the local prolongations are the diagonal blocks of the global one,
    and every rank stops at the same level (the shared stopping rule)."""
    import ctypes
    import scipy.sparse as sp
    from helpers import mg_hierarchy_slab
    from gcge_amd.lib import CSR
    h = host_lib()
    plane, nz = dims[0] * dims[1], dims[2]
    n = plane * nz

    def gen(rb, re_):
        if kind == "sio2":
            A, _ = make_problem("sio2", dims[0], row_begin=rb, row_end=re_, K=6, R0=1.5, R1=2.0, seed=12345)
            return A
        A = CSR()
        h.gcge_problem_lap3d_box(dims[0], dims[1], dims[2], ctypes.c_int64(rb), ctypes.c_int64(re_), ctypes.byref(A))
        return A
    whole = mg_hierarchy(gen(0, -1 if kind == "sio2" else n), 6, scale=0.5, min_rows=1)
    zcut = [2 * ((nz // 2) * r // world) for r in range(world)] + [nz]           # even plane numbers
    part = [z * plane for z in zcut]
    slabs = [mg_hierarchy_slab(gen(part[r], part[r + 1]), dims, part, r, 6, scale=0.5) for r in range(world)]
    L = len(slabs[0]["A"])
    assert all(len(sl["A"]) == L for sl in slabs) and L >= 2
    compared = 0
    for lev in range(1, L):
        # (identical while the cuts of the level that is coarsened lie on even planes; past an odd cut every rank pairs its own planes and
        #  the cells next to the cut differ from the whole hierarchy's: test_slab_hierarchy_with_cuts_on_odd_planes_is_galerkin)
        fd = slabs[0]["dims"][lev - 1]
        if any((c // (fd[0] * fd[1])) % 2 for c in slabs[0]["part"][lev - 1][:-1]):
            break
        compared += 1
        stacked = sp.vstack([sl["A"][lev] for sl in slabs]).tocsr()
        want = whole["A"][lev]
        assert stacked.shape == want.shape and abs(stacked - want).max() <= 1e-13 * abs(want).max(), lev
        assert slabs[0]["dims"][lev] == whole["dims"][lev]
        Pg = sp.block_diag([sl["P"][lev - 1] for sl in slabs]).tocsr()
        assert abs(Pg - whole["P"][lev - 1]).max() == 0.0
        assert all(sl["part"][lev] == slabs[0]["part"][lev] for sl in slabs)
    assert compared >= 1


@pytest.mark.parametrize("kind,dims,zcut", [("lap3d", (6, 5, 19), [0, 7, 19]), ("lap3d", (4, 6, 24), [0, 5, 6, 17, 24]), ("sio2", (16, 16, 16), [0, 9, 16]),
                                            ("sio2", (19, 19, 19), [0, 5, 12, 19]), ("lap3d", (4, 4, 9), [0, 1, 2, 9]),
                                            ("lap3d", (4, 4, 43), [0, 5, 11, 16, 21, 27, 32, 38, 43]), ("sio2", (21, 21, 21), [0, 2, 5, 8, 10, 13, 16, 18, 21])])
def test_slab_hierarchy_with_cuts_on_odd_planes_is_galerkin(kind, dims, zcut):
    """Cuts on ANY plane boundary (where the non-zeros balance): every rank pairs its own planes from its first one, so the cells next
    to an odd cut differ from the whole-matrix hierarchy's — the levels are still A_{l+1} = scale P^T A_l P with P = the ranks' local
    prolongations stacked diagonally, P a partition of the rows into cells of at most 2 x 2 x 2 grid points of ONE rank, columns of a
    coarse slab global coarse indices through the shared partition, and every rank stops at the same level."""
    import ctypes
    import scipy.sparse as sp
    from helpers import mg_hierarchy_slab
    from gcge_amd.lib import CSR
    h = host_lib()
    plane, nz = dims[0] * dims[1], dims[2]
    world = len(zcut) - 1

    def gen(rb, re_):
        if kind == "sio2":
            A, _ = make_problem("sio2", dims[0], row_begin=rb, row_end=re_, K=6, R0=1.5, R1=2.0, seed=12345)
            return A
        A = CSR()
        h.gcge_problem_lap3d_box(dims[0], dims[1], dims[2], ctypes.c_int64(rb), ctypes.c_int64(re_), ctypes.byref(A))
        return A
    part = [z * plane for z in zcut]
    slabs = [mg_hierarchy_slab(gen(part[r], part[r + 1]), dims, part, r, 6, scale=0.5) for r in range(world)]
    L = len(slabs[0]["A"])
    assert all(len(sl["A"]) == L for sl in slabs) and L >= 3
    fine = sp.vstack([sl["A"][0] for sl in slabs]).tocsr()
    assert abs(fine - fine.T).max() == 0.0
    for lev in range(1, L):
        assert all(sl["part"][lev] == slabs[0]["part"][lev] and sl["dims"][lev] == slabs[0]["dims"][lev] for sl in slabs)
        pl, pf, d, df = slabs[0]["part"][lev], slabs[0]["part"][lev - 1], slabs[0]["dims"][lev], slabs[0]["dims"][lev - 1]
        cplane = d[0] * d[1]
        # every rank halves ITS plane count (rounded up); the coarse grid holds the sum
        assert [(pl[r + 1] - pl[r]) // cplane for r in range(world)] == [((pf[r + 1] - pf[r]) // (df[0] * df[1]) + 1) // 2 for r in range(world)]
        assert d == ((df[0] + 1) // 2, (df[1] + 1) // 2, pl[-1] // cplane) and pl[-1] == d[0] * d[1] * d[2]
        P = sp.block_diag([sl["P"][lev - 1] for sl in slabs]).tocsr()
        assert P.shape == (pf[-1], pl[-1]) and np.all(np.asarray(P.sum(axis=1)).ravel() == 1.0)
        cells = np.asarray(P.sum(axis=0)).ravel()
        assert cells.min() >= 1 and cells.max() <= 8
        stacked = sp.vstack([sl["A"][lev] for sl in slabs]).tocsr()
        want = 0.5 * (P.T @ fine @ P)
        assert stacked.shape == want.shape and abs(stacked - want).max() <= 1e-13 * abs(want).max(), lev
        fine = stacked


@pytest.mark.parametrize("zcut,min_star", [([0, 16, 32, 48], 0.95), ([0, 17, 31, 48], 0.40)])
def test_coarse_slabs_of_the_sio2_like_matrix_keep_the_grid_form(zcut, min_star):
    """Level 1 of a slab hierarchy of the SiO2-like matrix (48^3, three slabs) is a star again (arm length 3 on a 24 x 24 x nz grid) and
    passes the host self-check of the plane sweep on a slab (star rows + remainder == CSR bit for bit, halo planes where the sweep's
    addressing expects them): nearly every row with even cuts; with cuts on odd planes the rows within reach of a thin cell go to the
    remainder — fewer star rows, same form."""
    import ctypes as C
    from helpers import mg_hierarchy_slab
    from gcge_amd import dist as gdist
    from gcge_amd.lib import CSR, hip_lib
    g = hip_lib()
    g.gcge_hip_star_selfcheck_slab.restype = C.c_long
    g.gcge_hip_star_selfcheck_slab.argtypes = [C.c_int, C.c_int, C.c_long, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_double), C.POINTER(C.c_long)]
    N = 48
    part = [z * N * N for z in zcut]
    for r in range(3):
        A, _ = make_problem("sio2", N, row_begin=part[r], row_end=part[r + 1], K=10, R0=2.0, R1=5.0, seed=12345)
        hier = mg_hierarchy_slab(A, (N, N, N), part, r, 2, scale=0.5)
        S = hier["A"][1].tocsr(); S.sort_indices()
        pl, d = hier["part"][1], hier["dims"][1]
        assert d[:2] == (24, 24) and d[2] == sum((zcut[q + 1] - zcut[q] + 1) // 2 for q in range(3))
        rp = np.ascontiguousarray(S.indptr, dtype=np.int32); ci = np.array(S.indices, dtype=np.int32, copy=True); va = np.ascontiguousarray(S.data)
        Al = CSR(S.shape[0], int(pl[-1]), int(pl[r]), int(S.nnz), rp.ctypes.data_as(C.POINTER(C.c_int)), ci.ctypes.data_as(C.POINTER(C.c_int)),
                 va.ctypes.data_as(C.POINTER(C.c_double)))
        gh = np.ascontiguousarray(gdist.localize_slab(Al), dtype=np.int32)
        out = (C.c_long * 12)()
        bad = g.gcge_hip_star_selfcheck_slab(Al.nrows, Al.ncols, int(pl[r]), int(pl[-1]), gh.ctypes.data_as(C.POINTER(C.c_int)), Al.rowptr, Al.colidx, Al.val, out)
        assert bad == 0 and tuple(out[:4]) == (d[0], d[1], d[2], 3), (r, bad, list(out))
        assert out[4] >= min_star * Al.nrows, (r, out[4], Al.nrows)
        assert (out[5], out[6]) == (pl[r] // 576, pl[r + 1] // 576)


# ---------------------------------------------------------------------------------------------- host: BlockAMG vs the reference
def _check_amg_case(backend, Ah, Ph, key, n0):
    g = G[key]
    m = g["m"]
    b = F(uniform(g["seed_b"], (n0, m)))
    x0 = F(uniform(g["seed_x"], (n0, m))) if "seed_x" in g else np.zeros((n0, m))
    x, niter, res = block_amg_solve(backend, Ah, Ph, b, x0, g["max_iter"], g["rate"], g["tol"])
    want = np.array(g["x"]).T
    assert np.max(np.abs(x - want)) <= 1e-11 * np.max(np.abs(want)), (key, np.max(np.abs(x - want)))
    assert abs(res - g["residual"]) <= 1e-9 * abs(g["residual"])
    return niter


def test_block_amg_on_the_dense_table_equals_the_reference():
    db = DenseBackend()
    As, Ps = toy1d_levels()
    Ah = [db.matrix(a) for a in As]
    Ph = [db.matrix(p) for p in Ps]
    _check_amg_case(db, Ah, Ph, "toy1d_amg", As[0].shape[0])
    assert _check_amg_case(db, Ah, Ph, "toy1d_amg_stop", As[0].shape[0]) == 2      # tol[0] met after the second of five cycles (same x as the reference: it stopped there too)


def test_from_i_to_j_on_the_dense_table_equals_the_reference():
    from gcge_amd.ops_struct import OPS
    db = DenseBackend()
    As, Ps = toy1d_levels()
    Ah = [db.matrix(a) for a in As]
    Ph = [db.matrix(p) for p in Ps]
    L = len(As)
    st = C.cast(db.ops_handle, C.POINTER(OPS)).contents
    fn = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p)(st.MultiVecFromItoJ)
    P_arr = (C.c_void_p * (L - 1))(*[p.value for p in Ph])
    ws = (C.c_void_p * L)(*[db.ops.mv_create(2, Ah[lev]).value for lev in range(L)])
    for key in ("toy1d_from_2_to_0", "toy1d_from_0_to_2", "toy1d_from_1_to_1"):
        g = G[key]
        li, lj = g["from"], g["to"]
        src = F(uniform(g["seed"], (As[li].shape[0], 2)))
        mf = db.mv_from_numpy(Ah[li], src)
        mt = db.ops.mv_create(2, Ah[lj])
        fn(P_arr, li, lj, mf, mt, (C.c_int * 2)(0, 0), (C.c_int * 2)(2, 2), ws, db.ops_handle)
        got = db.mv_to_numpy(mt, As[lj].shape[0], 0, 2)
        want = np.array(g["y"]).T
        assert np.max(np.abs(got - want)) <= 1e-14 * max(1.0, np.max(np.abs(want))), key


def _lap3d8_on(backend, via_slot):
    """BlockAMG over OUR hierarchy of Lap3D 8^3 against the reference's BlockAMG over the same hierarchy (dense)."""
    import scipy.sparse as sp
    g = G["lap3d8_amg"]
    h = host_lib()
    h.gcge_mg_set_defaults.argtypes = [C.c_double, C.c_int, C.c_double]
    h.gcge_mg_set_defaults(g["scale"], g["min_rows"], -1.0)
    A, _ = make_problem("lap3d", 8)
    mA = backend.matrix(A)
    try:
        if via_slot:
            Ah, Ph, done = slot_multigrid(backend, mA, None, 3)
        else:
            lev = mg_hierarchy(A, 3, scale=g["scale"], min_rows=g["min_rows"])
            keep = [csr_from_scipy(sp.csr_matrix(a)) for a in lev["A"][1:]] + [csr_from_scipy(sp.csr_matrix(p)) for p in lev["P"]]
            Ah = [mA] + [backend.matrix(k[0]) for k in keep[:len(lev["A"]) - 1]]
            Ph = [backend.matrix_rect(k[0]) for k in keep[len(lev["A"]) - 1:]]
            done = lambda: None     # noqa: E731
        assert len(Ah) == 3
        _check_amg_case(backend, Ah, Ph, "lap3d8_amg", 512)
        done()
    finally:
        h.gcge_mg_set_defaults(0.5, 64, 0.25)


def test_block_amg_on_the_oracle_over_its_multigrid_slot_equals_the_reference(oracle):
    _lap3d8_on(oracle, via_slot=True)


# ---------------------------------------------------------------------------------------------- host: GCG with BlockAMG
def test_gcg_with_block_amg_on_the_oracle(oracle):
    """-gcge_amg_levels: the W systems through BlockAMG (1 cycle, 5 + 5 CG smoothing steps on the finest level) instead of 30
    CG iterations — same eigenvalues (closed form, <= 1e-10), no more outer iterations than the plain solver."""
    A, _ = make_problem("lap3d", 20)
    mA = oracle.matrix(A)
    ev0, r0 = run_gcg(oracle.ops_handle, mA, None, ["-nevConv", 20])
    ev1, r1 = run_gcg(oracle.ops_handle, mA, None, ["-nevConv", 20, "-gcge_amg_levels", 4])
    ex = lap3d_exact(20, 20)
    assert r1.nevConv >= 20 and np.max(np.abs(ev1[:20] - ex) / ex) < 1e-10
    assert r1.numIter <= r0.numIter + 2


def test_gcg_with_block_amg_generalised_problem_on_the_oracle(oracle):
    A, B = make_problem("fe3d", 12)
    mA, mB = oracle.matrix(A), oracle.matrix(B)
    ev0, r0 = run_gcg(oracle.ops_handle, mA, mB, ["-nevConv", 10])
    ev1, r1 = run_gcg(oracle.ops_handle, mA, mB, ["-nevConv", 10, "-gcge_amg_levels", 3])
    assert r1.nevConv >= 10 and np.max(np.abs(ev1[:10] - ev0[:10]) / ev0[:10]) < 1e-10
    assert r1.numIter <= r0.numIter + 2


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_rectangular_matrix_products_on_hip(hip, oracle):
    """P and P^T of the hierarchy through MatDotMultiVec / MatTransDotMultiVec of the HIP table (generic CSR kernel) against scipy,
    widths 1 ... 66, odd column offsets."""
    A, _ = make_problem("lap3d", 11)
    lev = mg_hierarchy(A, 2, scale=1.0, min_rows=4)
    P = lev["P"][0]
    Pc, keep = csr_from_scipy(P)
    mP = hip.matrix_rect(Pc)
    mA = hip.matrix(A)
    Ac, keep2 = csr_from_scipy(lev["A"][1])
    mAc = hip.matrix(Ac)
    nf, nc = P.shape
    for m, off in ((1, 0), (2, 1), (7, 3), (16, 0), (33, 1), (66, 2)):
        xc = uniform(5 + m, (nc, m + off)) - 0.5
        vx = hip.mv_from_numpy(mAc, xc)
        vy = hip.ops.mv_create(m + off, mA)
        hip.ops.spmm(mP, vx, vy, (off, off), (off + m, off + m))
        got = hip.mv_to_numpy(vy, nf, off, off + m)
        assert np.max(np.abs(got - P @ xc[:, off:])) <= 1e-14
        xf = uniform(9 + m, (nf, m + off)) - 0.5
        vf = hip.mv_from_numpy(mA, xf)
        vc = hip.ops.mv_create(m + off, mAc)
        hip.ops.fn("MatTransDotMultiVec")(mP, vf, vc, (C.c_int * 2)(off, off), (C.c_int * 2)(off + m, off + m), hip.ops_handle)
        got = hip.mv_to_numpy(vc, nc, off, off + m)
        assert np.max(np.abs(got - P.T @ xf[:, off:])) <= 1e-13
        for v in (vx, vy, vf, vc):
            hip.ops.mv_destroy(v, m + off)
    hip.free_matrix_rect(mP)
    hip.free_matrix(mA)
    hip.free_matrix(mAc)


@pytest.mark.gpu
@pytest.mark.parametrize("host_smoother", [False, True])
def test_block_amg_on_hip_equals_the_reference(hip, host_smoother, monkeypatch):
    """The reference's BlockAMG result (dense back-end, tests/golden/amg.json) from our BlockAMG over the HIP table: hierarchy from
    the HIP MultiGridCreate slot, smoother = the fused device CG (default) or the solver stack's BlockPCG over the slots."""
    if host_smoother:
        monkeypatch.setenv("GCGE_AMG_HOST_SMOOTHER", "1")
    hip.set_random_mode(0)
    _lap3d8_on(hip, via_slot=True)
    _lap3d8_on(hip, via_slot=False)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,size,nev,extra", [("lap3d", 24, 20, []), ("sio2", 20, 10, []), ("fe3d", 16, 10, []),
                                                  ("lap3d", 21, 12, ["-gcge_amg_smooth0", 3, "-gcge_amg_smooth", 2])])
def test_gcg_with_block_amg_on_hip_matches_oracle(hip, oracle, kind, size, nev, extra):
    """GCG with the W systems solved by BlockAMG over the HIP hierarchy (-gcge_amg_levels) against the CPU oracle's plain run:
    Ritz values <= 1e-10 relative (north_star), no more outer iterations."""
    hip.set_random_mode(0)
    A, B = make_problem(kind, size)
    o_ev, o_res = run_gcg(oracle.ops_handle, oracle.matrix(A), oracle.matrix(B) if B is not None else None, ["-nevConv", nev])
    mA = hip.matrix(A)
    mB = hip.matrix(B) if B is not None else None
    C.CDLL(None).srand(0)
    ev, res = run_gcg(hip.ops_handle, mA, mB, ["-nevConv", nev, "-gcge_amg_levels", 4, "-gcge_initX_orth_method", "chol",
                                               "-gcge_compW_orth_method", "chol"] + extra)
    k = min(res.nevConv, o_res.nevConv)
    assert res.nevConv >= nev and k >= nev
    assert np.max(np.abs(ev[:k] - o_ev[:k]) / np.abs(o_ev[:k])) < 1e-10
    assert res.numIter <= o_res.numIter + 3
    hip.free_matrix(mA)
    if mB is not None:
        hip.free_matrix(mB)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,size,m", [("lap3d", 16, 8), ("lap3d", 24, 22), ("fe3d", 12, 6), ("sio2", 14, 4)])
def test_block_amg_fused_vcycle_steps_equal_the_slot_calls(hip, kind, size, m, monkeypatch):
    """The V-cycle's residual r = b - A x and its correction x += P e as one sweep each on the HIP table (GCGE_SetBlockAMGFusions:
    the CG's start sweep with a single store, a prolongation kernel that adds in place) against the slot calls they replace
    (MatDotMultiVec + MultiVecAxpby, MultiVecFromItoJ + MultiVecAxpby; reference src/ops_lin_sol.c:596-640): the same x bit for
    bit, and the solve converges to the direct solution."""
    import scipy.sparse.linalg as sla
    A, _ = make_problem(kind, size)
    S = csr_to_scipy(A)
    n = A.nrows
    mA = hip.matrix(A)
    Ah, Ph, done = slot_multigrid(hip, mA, None, 3)
    b = uniform(301, (n, m)) - 0.5
    x0 = uniform(302, (n, m)) - 0.5
    L = len(Ah)
    max_iter = [2] + [3, 4] * L
    rate, tol = [1e-30] * L, [1e-30] * L
    out = {}
    for tag in ("fused", "slots"):
        if tag == "slots":
            monkeypatch.setenv("GCGE_AMG_NO_FUSIONS", "1")
        out[tag] = block_amg_solve(hip, Ah, Ph, b, x0, max_iter, rate, tol)
    monkeypatch.delenv("GCGE_AMG_NO_FUSIONS")
    assert np.array_equal(out["fused"][0], out["slots"][0])
    assert out["fused"][1] == out["slots"][1] and out["fused"][2] == out["slots"][2]
    x, _, _ = block_amg_solve(hip, Ah, Ph, b, x0, [40] + [3, 4] * L, rate, [1e-11] + [1e-30] * (L - 1))
    ref = sla.spsolve(S.tocsc(), b)
    assert np.max(np.abs(x - ref)) < 1e-8 * np.max(np.abs(ref))
    done()
    hip.free_matrix(mA)


@pytest.mark.gpu
def test_reference_block_amg_over_the_hip_table(hip):
    """The literal drop-in of the multigrid leg: the REFERENCE's compiled BlockAMG + BlockPCG + DefaultMultiVecFromItoJ
    (oracle/_ref/libgcge_ref.so, src/ops_lin_sol.c:466-715, src/ops_multi_grid.c:69-117) over a table only OPS_HIP_Set touched —
    hierarchy from the HIP MultiGridCreate slot, rectangular prolongations through MatDotMultiVec / MatTransDotMultiVec, work
    blocks from MultiVecCreateByMat of the level matrices (set-up of test/test_eig_sol_SiO2_MAT.c:96-128,160-170) — against OUR
    BlockAMG over the same table with the slot-level smoother: the same x; and against a direct solve after enough cycles."""
    import os
    import scipy.sparse.linalg as sla
    import pyoracle as po
    ref = po.ref_lib()
    if ref is None:
        pytest.skip("oracle/_ref/libgcge_ref.so not present")
    A, _ = make_problem("lap3d", 12)
    S = csr_to_scipy(A)
    n, m = A.nrows, 6
    mA = hip.matrix(A)
    ops2 = C.c_void_p()
    hip.h.OPS_Create(C.byref(ops2))
    hip.g.OPS_HIP_Set(ops2)
    b = F(uniform(201, (n, m)) - 0.5)
    vb = hip.mv_from_numpy(mA, b)
    vx = hip.mv_from_numpy(mA, np.zeros_like(b))
    max_iter = [3, 4, 4, 3, 3, 8, 0]
    rate, tol = [1e-30] * 3, [1e-30] * 3
    mi = (C.c_int * len(max_iter))(*max_iter); ra = (C.c_double * 3)(*rate); to = (C.c_double * 3)(*tol)
    res = C.c_double()
    ref.ref_block_amg_foreign.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L = ref.ref_block_amg_foreign(ops2, mA, 3, mi, ra, to, m, vb, vx, C.byref(res))
    assert L == 3
    x_ref = hip.mv_to_numpy(vx, n, 0, m)
    os.environ["GCGE_AMG_HOST_SMOOTHER"] = "1"
    try:
        Ah, Ph, done = slot_multigrid(hip, mA, None, 3)
        x_own, _, res_own = block_amg_solve(hip, Ah, Ph, b, np.zeros_like(b), max_iter, rate, tol)
        done()
    finally:
        os.environ.pop("GCGE_AMG_HOST_SMOOTHER", None)
    assert np.max(np.abs(x_ref - x_own)) <= 1e-10 * np.max(np.abs(x_own))
    assert abs(res.value - res_own) <= 1e-8 * abs(res_own)
    xs = sla.spsolve(S.tocsc(), b)
    assert np.max(np.abs(x_ref - xs)) / np.max(np.abs(xs)) < 0.05          # three V-cycles: two digits
    hip.ops.mv_destroy(vb, m)
    hip.ops.mv_destroy(vx, m)
    hip.free_matrix(mA)
