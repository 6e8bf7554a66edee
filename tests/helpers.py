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
