"""CPU: matrix ingestion formats — PETSc binary Mat files and CCS / MATLAB-style triples."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from gcge_amd.lib import CSR, host_lib, make_problem
from helpers import csr_to_scipy


def _lib():
    h = host_lib()
    h.gcge_load_petsc_binary.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(CSR)]
    h.gcge_save_petsc_binary.argtypes = [C.c_char_p, C.POINTER(CSR)]
    h.gcge_csr_from_ccs.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                    C.c_int, C.POINTER(CSR)]
    return h


def _write_petsc_with_numpy(path, S):
    """Independent writer of the published format: big-endian int32 header/lengths/indices, float64 values."""
    S = S.tocsr()
    S.sort_indices()
    with open(path, "wb") as f:
        np.array([1211216, S.shape[0], S.shape[1], S.nnz], dtype=">i4").tofile(f)
        np.diff(S.indptr).astype(">i4").tofile(f)
        S.indices.astype(">i4").tofile(f)
        S.data.astype(">f8").tofile(f)


def test_petsc_binary_load_matches_numpy_written_file(tmp_path):
    h = _lib()
    rng = np.random.default_rng(3)
    S = sp.random(57, 57, density=0.12, random_state=rng, format="csr") + sp.eye(57) * 3.5
    path = str(tmp_path / "a.petsc").encode()
    _write_petsc_with_numpy(path.decode(), S)
    A = CSR()
    assert h.gcge_load_petsc_binary(path, 0, -1, C.byref(A)) == 0
    got = csr_to_scipy(A)
    assert (got != S.tocsr()).nnz == 0
    # a row slab with global columns, as the row-partitioned path wants it
    B = CSR()
    assert h.gcge_load_petsc_binary(path, 20, 41, C.byref(B)) == 0
    assert (B.nrows, B.ncols, B.row_begin) == (21, 57, 20)
    ip = np.ctypeslib.as_array(B.rowptr, shape=(B.nrows + 1,))
    ci = np.ctypeslib.as_array(B.colidx, shape=(int(B.nnz),))
    va = np.ctypeslib.as_array(B.val, shape=(int(B.nnz),))
    slab = sp.csr_matrix((va, ci, ip), shape=(21, 57))
    assert (slab != S.tocsr()[20:41]).nnz == 0
    h.gcge_csr_free(C.byref(A)); h.gcge_csr_free(C.byref(B))


def test_petsc_binary_round_trip_and_errors(tmp_path):
    h = _lib()
    A, _ = make_problem("sio2", 6, K=4, R0=1.5, R1=2.0, seed=5)
    path = str(tmp_path / "b.petsc").encode()
    assert h.gcge_save_petsc_binary(path, C.byref(A)) == 0
    # numpy reads back exactly the published layout
    raw = np.fromfile(path.decode(), dtype=">i4", count=4)
    assert list(raw) == [1211216, A.nrows, A.ncols, int(A.nnz)]
    Bm = CSR()
    assert h.gcge_load_petsc_binary(path, 0, -1, C.byref(Bm)) == 0
    assert (csr_to_scipy(Bm) != csr_to_scipy(A)).nnz == 0
    h.gcge_csr_free(C.byref(Bm))
    assert h.gcge_load_petsc_binary(b"/nonexistent/file", 0, -1, C.byref(Bm)) == -1
    bad = tmp_path / "bad.petsc"
    bad.write_bytes(b"\\x00" * 64)
    assert h.gcge_load_petsc_binary(str(bad).encode(), 0, -1, C.byref(Bm)) == -2
    short = tmp_path / "short.petsc"
    short.write_bytes(open(path.decode(), "rb").read()[:200])
    assert h.gcge_load_petsc_binary(str(short).encode(), 0, -1, C.byref(Bm)) == -1


@pytest.mark.parametrize("one_based", [0, 1])
def test_csr_from_ccs_general_matrix(one_based):
    h = _lib()
    rng = np.random.default_rng(11)
    S = sp.random(23, 31, density=0.2, random_state=rng, format="csc")
    S.sort_indices()
    jc = (S.indptr + one_based).astype(np.int32)
    ir = (S.indices + one_based).astype(np.int32)
    pr = S.data.astype(np.float64)
    A = CSR()
    rc = h.gcge_csr_from_ccs(23, 31, jc.ctypes.data_as(C.POINTER(C.c_int)), ir.ctypes.data_as(C.POINTER(C.c_int)),
                             pr.ctypes.data_as(C.POINTER(C.c_double)), one_based, C.byref(A))
    assert rc == 0 and (A.nrows, A.ncols, int(A.nnz)) == (23, 31, S.nnz)
    ip = np.ctypeslib.as_array(A.rowptr, shape=(24,))
    ci = np.ctypeslib.as_array(A.colidx, shape=(S.nnz,))
    va = np.ctypeslib.as_array(A.val, shape=(S.nnz,))
    got = sp.csr_matrix((va, ci, ip), shape=(23, 31))
    assert (got != S.tocsr()).nnz == 0
    assert all(np.all(np.diff(ci[ip[r]:ip[r + 1]]) > 0) for r in range(23)), "column indices ascending inside rows"
    h.gcge_csr_free(C.byref(A))


def test_matrix_market_reader(tmp_path):
    """Matrix Market coordinate files (the form SuiteSparse ships SiO2 / Ga41As41H72 ... of the reference's test/submit.sh:9-15
    in): a file written by scipy (symmetric, lower triangle) and hand-written ones (general with comment lines, entries out of
    order, a duplicate; pattern; integer) against scipy; our own writer round-trips a generator matrix bit for bit; bad files."""
    import scipy.io as sio
    h = host_lib()
    h.gcge_load_matrix_market.argtypes = [C.c_char_p, C.POINTER(CSR)]
    h.gcge_save_matrix_market.argtypes = [C.c_char_p, C.POINTER(CSR), C.c_int]
    rng = np.random.default_rng(4)
    S = sp.random(61, 61, density=0.1, random_state=rng, format="csr")
    S = (S + S.T + sp.eye(61) * 2.25).tocsr()
    path = str(tmp_path / "s.mtx")
    sio.mmwrite(path, S, symmetry="symmetric", precision=17)
    A = CSR()
    assert h.gcge_load_matrix_market(path.encode(), C.byref(A)) == 0 and (A.nrows, A.ncols) == (61, 61)
    got = csr_to_scipy(A)
    assert got.has_sorted_indices and abs(got - S).max() == 0.0 and got.nnz == S.nnz
    h.gcge_csr_free(C.byref(A))
    gen = str(tmp_path / "g.mtx")
    with open(gen, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% a comment\n\n3 4 5\n3 1 -1.5\n1 2 2.0\n% another\n1 1 1.0\n1 2 0.25\n2 4 7e-1\n")
    assert h.gcge_load_matrix_market(gen.encode(), C.byref(A)) == 0 and (A.nrows, A.ncols, A.nnz) == (3, 4, 4)
    assert np.array_equal(csr_to_scipy(A).toarray(), np.array([[1.0, 2.25, 0, 0], [0, 0, 0, 0.7], [-1.5, 0, 0, 0]]))
    h.gcge_csr_free(C.byref(A))
    pat = str(tmp_path / "p.mtx")
    with open(pat, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n2 1\n3 3\n3 1\n")
    assert h.gcge_load_matrix_market(pat.encode(), C.byref(A)) == 0
    assert np.array_equal(csr_to_scipy(A).toarray(), np.array([[0, 1.0, 1.0], [1.0, 0, 0], [1.0, 0, 1.0]]))
    h.gcge_csr_free(C.byref(A))
    # our writer -> our reader: bit for bit (both triangles from the lower one)
    G, _ = make_problem("sio2", 7, K=3, R0=1.5, R1=2.0, seed=9)
    out = str(tmp_path / "w.mtx")
    for symmetric in (1, 0):
        assert h.gcge_save_matrix_market(out.encode(), C.byref(G), symmetric) == 0
        assert h.gcge_load_matrix_market(out.encode(), C.byref(A)) == 0
        a, b = csr_to_scipy(A), csr_to_scipy(G)
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)
        assert abs(sio.mmread(out).tocsr() - b).max() == 0.0
        h.gcge_csr_free(C.byref(A))
    bad = str(tmp_path / "bad.mtx")
    for text, rc in (("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n", -2), ("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n", -2),
                     ("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n", -1), ("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n", -2)):
        with open(bad, "w") as f:
            f.write(text)
        assert h.gcge_load_matrix_market(bad.encode(), C.byref(A)) == rc, text
    assert h.gcge_load_matrix_market(str(tmp_path / "missing.mtx").encode(), C.byref(A)) == -1
