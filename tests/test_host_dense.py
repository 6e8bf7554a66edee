"""Host-side small dense pieces (no GPU): the symmetric eigensolver that replaces dsyevx/dsyev
(ops_eig_sol_gcg.c:1201, ops_orth.c:144) and the host dense table behind ops->lapack_ops."""
import ctypes as C

import numpy as np
import pytest

from gcge_amd.lib import host_lib

DP = C.POINTER(C.c_double)


@pytest.mark.parametrize("n", [1, 2, 3, 10, 64, 257])
def test_symeig_against_numpy(n):
    h = host_lib()
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n)); A = np.asfortranarray(M + M.T)
    if n >= 10:                       # clustered / repeated eigenvalues like the Laplacian spectra
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        d = np.sort(np.concatenate([np.repeat(1.0, 3), np.repeat(2.0, 3), rng.uniform(3, 9, n - 6)]))
        A = np.asfortranarray(Q @ np.diag(d) @ Q.T); A = np.asfortranarray((A + A.T) / 2)
    for uplo in (b"U", b"L"):
        w = np.zeros(n); Z = np.zeros((n, n), order="F"); work = np.zeros(2 * n + 8)
        Ain = A.copy(order="F")
        rc = h.GCGE_SymEig(C.c_char(uplo), n, Ain.ctypes.data_as(DP), n, w.ctypes.data_as(DP), Z.ctypes.data_as(DP), n,
                           work.ctypes.data_as(DP))
        assert rc == 0
        assert np.array_equal(Ain, A), "input must not be modified"
        wr = np.linalg.eigvalsh(A)
        assert np.max(np.abs(w - wr)) < 1e-12 * max(1.0, np.max(np.abs(wr)))
        assert np.max(np.abs(Z.T @ Z - np.eye(n))) < 1e-12 * n
        assert np.max(np.abs(A @ Z - Z * w)) < 1e-11 * max(1.0, np.max(np.abs(wr))) * n


def test_dense_table_orth_and_qtap():
    """DenseMatOrth (app_lapack.c:653-699: project out leading columns, pivoted QR, rank by |r_ii|) and
    DenseMatQtAP with a symmetric A given by its lower triangle (the P^T A P of the RR step)."""
    from gcge_amd.ops_struct import OPS
    h = host_lib()
    ops = C.c_void_p(); h.OPS_Create(C.byref(ops)); h.OPS_DENSE_Set(ops)
    st = C.cast(ops, C.POINTER(OPS)).contents
    rng = np.random.default_rng(7)
    m, n0, n1 = 40, 4, 7
    M = np.zeros((m, n0 + n1), order="F")
    M[:, :n0] = np.linalg.qr(rng.standard_normal((m, n0)))[0]
    M[:, n0:] = rng.standard_normal((m, n1))
    M[:, n0 + 5] = M[:, n0 + 1]              # exact duplicate -> rank n1 - 1
    M[:, n0 + 6] = M[:, 0] + M[:, 2]         # inside the span of the leading columns -> dropped too
    end = C.c_int(n0 + n1)
    dbl = np.zeros(4096); iw = np.zeros(64, dtype=np.int32)
    fn = C.CFUNCTYPE(None, DP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_double, DP, C.c_int, C.POINTER(C.c_int))(st.DenseMatOrth)
    fn(M.ctypes.data_as(DP), m, m, n0, C.byref(end), 1e-10, dbl.ctypes.data_as(DP), 4096, iw.ctypes.data_as(C.POINTER(C.c_int)))
    assert end.value == n0 + n1 - 2
    Q = M[:, :end.value]
    assert np.max(np.abs(Q.T @ Q - np.eye(end.value))) < 1e-12
    # C = Q^T A P with A symmetric ('L' triangle), 'S' output
    k = 9
    S = rng.standard_normal((m, m)); S = S + S.T
    L = np.asfortranarray(np.tril(S))
    P = np.asfortranarray(rng.standard_normal((m, k)))
    Cm = np.zeros((k, k), order="F"); ws = np.zeros(m * k)
    q = C.CFUNCTYPE(None, C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, DP, C.c_int, DP, C.c_int, DP,
                    C.c_int, C.c_double, DP, C.c_int, DP)(st.DenseMatQtAP)
    q(b"L", b"S", m, m, k, k, 1.0, P.ctypes.data_as(DP), m, L.ctypes.data_as(DP), m, P.ctypes.data_as(DP), m, 0.0,
      Cm.ctypes.data_as(DP), k, ws.ctypes.data_as(DP))
    assert np.max(np.abs(Cm - P.T @ S @ P)) < 1e-11
