"""Shared test helpers: the CPU oracle behind the same Python face as gcge_amd.HipBackend,
seeded inputs (gcge_uniform stream) and the problems used by both GPU and CPU tests."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as po
from gcge_amd.lib import host_lib, make_problem, run_gcg
from gcge_amd.ops_struct import OpsTable

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def uniform(seed, shape):
    """The gcge_uniform(seed, index) stream, vectorised (splitmix64)."""
    n = int(np.prod(shape))
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape, order="F")


def test_matrix(kind, size, **kw):
    return make_problem(kind, size, **kw)


def csr_to_scipy(A):
    import scipy.sparse as sp
    n = A.nrows
    rp = np.ctypeslib.as_array(A.rowptr, shape=(n + 1,)).copy()
    ci = np.ctypeslib.as_array(A.colidx, shape=(int(A.nnz),)).copy()
    va = np.ctypeslib.as_array(A.val, shape=(int(A.nnz),)).copy()
    return sp.csr_matrix((va, ci, rp), shape=(n, A.ncols))


def csr_from_scipy(S):
    """scipy sparse matrix -> (CSR struct, keepalive arrays) with ascending column indices inside every row."""
    import ctypes as C
    from gcge_amd.lib import CSR
    S = S.tocsr()
    S.sort_indices()
    rp = np.ascontiguousarray(S.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(S.indices, dtype=np.int32)
    va = np.ascontiguousarray(S.data, dtype=np.float64)
    A = CSR(S.shape[0], S.shape[1], 0, int(S.nnz), rp.ctypes.data_as(C.POINTER(C.c_int)),
            ci.ctypes.data_as(C.POINTER(C.c_int)), va.ctypes.data_as(C.POINTER(C.c_double)))
    return A, (rp, ci, va)


class OracleBackend:
    """CPU oracle with the interface of gcge_amd.hip_backend.HipBackendImpl."""

    def __init__(self, quiet=True):
        self.h = host_lib()
        self.o = po.oracle_lib()
        self.ops_handle = po.make_ops(quiet)
        self.ops = OpsTable(self.ops_handle)
        self._keep = []

    def matrix(self, csr):
        m = po.ccs_from_csr(csr)
        self._keep.append((m, csr))
        return C.cast(C.pointer(m), C.c_void_p)

    def free_matrix(self, m):
        pass

    def mv_from_numpy(self, mat, arr):
        a = np.asfortranarray(arr, dtype=np.float64)
        mv = self.ops.mv_create(a.shape[1], mat)
        v = C.cast(mv, C.POINTER(po.OVec)).contents
        dst = np.ctypeslib.as_array(v.data, shape=(v.ncols, v.ldd))
        dst[:, :a.shape[0]] = a.T
        return mv

    def mv_to_numpy(self, mv, n, c0, c1):
        v = C.cast(mv, C.POINTER(po.OVec)).contents
        src = np.ctypeslib.as_array(v.data, shape=(v.ncols, v.ldd))
        return np.asfortranarray(src[c0:c1, :n].T.copy())

    def set_random_mode(self, mode, seed=0):
        pass

    def sync(self):
        pass


def gcg_on(backend, kind, size, args, flag=0, **kw):
    """Run the GCG harness through `backend` on a generated problem."""
    A, B = make_problem(kind, size, **kw)
    mA = backend.matrix(A)
    mB = backend.matrix(B) if B is not None else None
    ev, res = run_gcg(backend.ops_handle, mA, mB, args, flag=flag)
    backend.free_matrix(mA)
    if mB is not None:
        backend.free_matrix(mB)
    return ev, res


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def lap3d_exact(N, count):
    c = 2.0 * np.cos(np.arange(1, N + 1) * np.pi / (N + 1))
    lam = (6.0 - c[:, None, None] - c[None, :, None] - c[None, None, :]).ravel()
    return np.sort(lam)[:count]


def sine_start_block(N, modes, eps, seed):
    """Deterministic warm-start vectors on an N^3 grid (natural order, i fastest): tensor sine modes (p,q,r) — the exact
    eigenvectors of the 7-point Laplacian, good approximations for the P1 pair — plus eps * (uniform - 0.5) noise."""
    t = np.arange(1, N + 1) * np.pi / (N + 1)
    cols = []
    for (p, q, r) in modes:
        v = np.sin(r * t)[:, None, None] * np.sin(q * t)[None, :, None] * np.sin(p * t)[None, None, :]   # [k, j, i]
        cols.append(v.ravel())
    X = np.stack(cols, axis=1)
    return X + eps * (uniform(seed, X.shape) - 0.5)


WARM_MODES = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 1), (2, 1, 2)]


def perturbed_stencil(kind, size, seed, which="A"):
    """The matrix of make_problem(kind, size) with the same pattern and DIFFERENT coefficients in every row (symmetric,
    strictly diagonally dominant: SPD): a_ij = -(1 + 0.2 w_i w_j) |a_ij(orig)| off the diagonal, a_ii = sum_j |a_ij| + 0.1.
    Returns (CSR struct, keepalive)."""
    import scipy.sparse as sp
    A, B = make_problem(kind, size)
    S = csr_to_scipy(A if which == "A" else B).tocoo()
    w = uniform(seed, (S.shape[0],))
    off = S.row != S.col
    data = np.where(off, -(1.0 + 0.2 * w[S.row] * w[S.col]) * np.abs(S.data), 0.0)
    M = sp.coo_matrix((data, (S.row, S.col)), shape=S.shape).tocsr()
    d = np.asarray(np.abs(M).sum(axis=1)).ravel() + 0.1
    M = (M + sp.diags(d)).tocsr()
    return csr_from_scipy(M)


class DenseMat(C.Structure):
    """GCGE_DENSE (include/gcge_ops.h) == the reference's LAPACKMAT / LAPACKVEC (app/app_lapack.h:17-20)."""
    _fields_ = [("data", C.POINTER(C.c_double)), ("nrows", C.c_int), ("ncols", C.c_int), ("ldd", C.c_int)]


class DenseBackend:
    """The host dense table (OPS_DENSE_Set, csrc/host/dense_host.c) behind the Python face of the other back-ends:
    matrices and blocks are column-major numpy arrays."""

    def __init__(self, quiet=True):
        self.h = host_lib()
        self.ops_handle = C.c_void_p()
        self.h.OPS_Create(C.byref(self.ops_handle))
        self.h.OPS_DENSE_Set(self.ops_handle)
        self.h.OPS_Setup(self.ops_handle)
        self.h.GCGE_SetQuiet(self.ops_handle, 1 if quiet else 0)
        self.ops = OpsTable(self.ops_handle)
        self._keep = []

    def matrix(self, arr):
        a = np.asfortranarray(arr, dtype=np.float64)
        m = DenseMat(a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], a.shape[1], a.shape[0])
        self._keep.append((m, a))
        return C.cast(C.pointer(m), C.c_void_p)

    def free_matrix(self, m):
        pass

    def mv_from_numpy(self, mat, arr):
        a = np.asfortranarray(arr, dtype=np.float64)
        mv = self.ops.mv_create(a.shape[1], mat)
        v = C.cast(mv, C.POINTER(DenseMat)).contents
        np.ctypeslib.as_array(v.data, shape=(v.ncols, v.ldd))[:, :a.shape[0]] = a.T
        return mv

    def mv_to_numpy(self, mv, n, c0, c1):
        v = C.cast(mv, C.POINTER(DenseMat)).contents
        return np.asfortranarray(np.ctypeslib.as_array(v.data, shape=(v.ncols, v.ldd))[c0:c1, :n].T.copy())


def block_amg_solve(backend, A_handles, P_handles, b, x0, max_iter, rate, tol, tol_type="abs"):
    """x = BlockAMG(b) from x0 through `backend`'s table (csrc/host/lin_sol.c; reference src/ops_lin_sol.c:466-715) over the
    hierarchy A_handles / P_handles (matrix handles of that back-end).  max_iter = [cycles, pre_0, post_0, pre_1, post_1, ...].
    Returns (x, niter, residual)."""
    h = host_lib()
    L = len(A_handles)
    m = b.shape[1]
    n0 = b.shape[0]
    ops = backend.ops
    A_arr = (C.c_void_p * L)(*[a.value if isinstance(a, C.c_void_p) else a for a in A_handles])
    P_arr = (C.c_void_p * max(1, L - 1))(*[p.value if isinstance(p, C.c_void_p) else p for p in P_handles])
    ws_arrays = [(C.c_void_p * L)() for _ in range(5)]
    made = []
    for i in range(5):
        for lev in range(L):
            mv = ops.mv_create(m, C.c_void_p(A_arr[lev]))
            ws_arrays[i][lev] = mv.value
            made.append(mv)
    ws_ptrs = (C.POINTER(C.c_void_p) * 5)(*[C.cast(a, C.POINTER(C.c_void_p)) for a in ws_arrays])
    mi = (C.c_int * len(max_iter))(*max_iter)
    ra = (C.c_double * len(rate))(*rate)
    to = (C.c_double * len(tol))(*tol)
    dbl = (C.c_double * (6 * m + 8))()
    iw = (C.c_int * (2 * m + 8))()
    h.MultiLinearSolverSetup_BlockAMG.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    h.MultiLinearSolverSetup_BlockAMG(mi, ra, to, tol_type.encode(), A_arr, P_arr, L, ws_ptrs, dbl, iw, None, backend.ops_handle)
    A0 = C.c_void_p(A_arr[0])
    mb = backend.mv_from_numpy(A0, b)
    mx = backend.mv_from_numpy(A0, x0)
    ops.multi_linear_solver(A0, mb, mx, (0, 0), (m, m))
    x = backend.mv_to_numpy(mx, n0, 0, m)

    class S(C.Structure):       # BlockAMGSolver: the leading members up to niter / residual (include/gcge_solver.h)
        _fields_ = [("max_iter", C.c_void_p), ("rate", C.c_void_p), ("tol", C.c_void_p), ("tol_type", C.c_char * 8),
                    ("A_array", C.c_void_p), ("P_array", C.c_void_p), ("num_levels", C.c_int), ("ws", C.c_void_p * 5),
                    ("dbl_ws", C.c_void_p), ("int_ws", C.c_void_p), ("pc", C.c_void_p), ("niter", C.c_int), ("residual", C.c_double)]
    st = C.cast(backend.ops.struct.multi_linear_solver_workspace, C.POINTER(S)).contents
    niter, residual = st.niter, st.residual
    for mv in made + [mb, mx]:
        ops.mv_destroy(mv, m)
    return x, niter, residual


def mg_hierarchy(A, max_levels, scale=0.0, min_rows=0, B=None):
    """The aggregation hierarchy of include/gcge_multigrid.h (csrc/host/multigrid.c) of a CSR struct as scipy matrices:
    {"A": [...], "P": [...], "PT": [...], "dims": [...], "B": [...] or None}."""
    import scipy.sparse as sp
    from gcge_amd.lib import CSR
    h = host_lib()

    class MG(C.Structure):
        _fields_ = [("num_levels", C.c_int), ("A", C.POINTER(CSR)), ("B", C.POINTER(CSR)), ("P", C.POINTER(CSR)),
                    ("PT", C.POINTER(CSR)), ("dims", C.POINTER(C.c_int * 3))]
    mg = MG()
    h.gcge_mg_build.argtypes = [C.POINTER(CSR), C.POINTER(CSR), C.c_int, C.c_int, C.c_double, C.POINTER(MG)]
    rc = h.gcge_mg_build(C.byref(A), C.byref(B) if B is not None else None, max_levels, min_rows, scale, C.byref(mg))
    assert rc == 0, rc

    def to_sp(c):
        rp = np.ctypeslib.as_array(c.rowptr, shape=(c.nrows + 1,)).copy()
        ci = np.ctypeslib.as_array(c.colidx, shape=(max(1, int(c.nnz)),))[:int(c.nnz)].copy()
        va = np.ctypeslib.as_array(c.val, shape=(max(1, int(c.nnz)),))[:int(c.nnz)].copy()
        return sp.csr_matrix((va, ci, rp), shape=(c.nrows, c.ncols))
    L = mg.num_levels
    out = {"A": [to_sp(mg.A[lev]) for lev in range(L)], "P": [to_sp(mg.P[lev]) for lev in range(L - 1)],
           "PT": [to_sp(mg.PT[lev]) for lev in range(L - 1)], "dims": [tuple(mg.dims[lev]) for lev in range(L)],
           "B": [to_sp(mg.B[lev]) for lev in range(L)] if B is not None else None}
    h.gcge_mg_free.argtypes = [C.POINTER(MG)]
    h.gcge_mg_free(C.byref(mg))
    return out


def mg_hierarchy_slab(A_slab, dims, part, rank, max_levels, scale=0.0):
    """gcge_mg_build_slab (csrc/host/multigrid.c) of one row slab with GLOBAL columns: {"A": [scipy slabs with global columns],
    "P": [...local...], "part": [[...] per level], "dims": [...]}."""
    import scipy.sparse as sp
    from gcge_amd.lib import CSR
    h = host_lib()

    class MG(C.Structure):
        _fields_ = [("num_levels", C.c_int), ("A", C.POINTER(CSR)), ("B", C.POINTER(CSR)), ("P", C.POINTER(CSR)),
                    ("PT", C.POINTER(CSR)), ("dims", C.POINTER(C.c_int * 3))]
    mg = MG()
    world = len(part) - 1
    parr = (C.c_long * (world + 1))(*[int(v) for v in part])
    d = (C.c_int * 3)(*dims)
    pl = C.POINTER(C.c_long)()
    h.gcge_mg_build_slab.argtypes = [C.POINTER(CSR), C.POINTER(C.c_int * 3), C.POINTER(C.c_long), C.c_int, C.c_int, C.c_int, C.c_double,
                                     C.POINTER(MG), C.POINTER(C.POINTER(C.c_long))]
    rc = h.gcge_mg_build_slab(C.byref(A_slab), C.byref(d), parr, rank, world, max_levels, scale, C.byref(mg), C.byref(pl))
    assert rc == 0, rc

    def to_sp(c):
        rp = np.ctypeslib.as_array(c.rowptr, shape=(c.nrows + 1,)).copy()
        ci = np.ctypeslib.as_array(c.colidx, shape=(max(1, int(c.nnz)),))[:int(c.nnz)].copy()
        va = np.ctypeslib.as_array(c.val, shape=(max(1, int(c.nnz)),))[:int(c.nnz)].copy()
        return sp.csr_matrix((va, ci, rp), shape=(c.nrows, c.ncols))
    L = mg.num_levels
    out = {"A": [to_sp(mg.A[lev]) for lev in range(L)], "P": [to_sp(mg.P[lev]) for lev in range(L - 1)],
           "part": [[pl[lev * (world + 1) + r] for r in range(world + 1)] for lev in range(L)],
           "dims": [tuple(mg.dims[lev]) for lev in range(L)]}
    h.gcge_mg_free.argtypes = [C.POINTER(MG)]
    h.gcge_mg_free(C.byref(mg))
    C.CDLL(None).free(pl)
    return out
