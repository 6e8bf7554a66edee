"""CPU (no GPU): the oracle back-end and the host solver stack against the golden vectors made
by the REAL reference (tests/golden/make_golden.py, oracle/_ref)."""
import numpy as np
import pytest

from helpers import gcg_on, lap3d_exact, load_golden
from slot_cases import run_bpcg_case, run_orth_cases, run_slot_cases
from solver_setup import bpcg_setup, orth_setup


def test_oracle_slots_match_reference_vectors(oracle):
    P = run_slot_cases(oracle)
    run_orth_cases(oracle, P, orth_setup(oracle.ops_handle))
    run_bpcg_case(oracle, P, bpcg_setup(oracle.ops_handle))


def test_reference_library_binds_to_itself():
    """oracle/_ref/libgcge_ref.so and our libgcge_host.so export the SAME names (ours mirrors the reference's API) and the tests
    load ours RTLD_GLOBAL: unless the reference library is linked -Bsymbolic the dynamic linker resolves its internal calls
    (OPS_Setup, EigenSolverSetup_GCG, MultiVecOrthSetup_*, DefaultMultiVec*) to OUR functions, and a "reference run" is our solver
    over the reference's back-end.  Checked three ways: the ELF flag, the addresses inside a table the reference built, and
    that the library's own TestEigenSolverGCG is not ours."""
    import ctypes as C
    import os
    import subprocess
    import pyoracle as po
    from gcge_amd.lib import host_lib
    from gcge_amd.ops_struct import OPS
    ref = po.ref_lib()
    if ref is None:
        pytest.skip("oracle/_ref not present on this box")
    h = host_lib()
    for name in ("libgcge_ref.so", "libgcge_ref_omp.so"):
        path = os.path.join(os.path.dirname(po.__file__), "_ref", name)
        if os.path.exists(path):
            dyn = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
            assert "SYMBOLIC" in dyn, "%s is not linked -Bsymbolic" % name
    ref.ref_address_of.restype = C.c_void_p
    ref.ref_address_of.argtypes = [C.c_char_p]
    ref.ref_make_ccs_ops.restype = C.c_void_p
    ops = C.cast(C.c_void_p(ref.ref_make_ccs_ops()), C.POINTER(OPS)).contents      # OPS_Create + OPS_CCS_Set + OPS_Setup, all inside the library
    assert ops.MultiVecQtAP == ref.ref_address_of(b"DefaultMultiVecQtAP")
    assert ops.MultiVecQtAP != C.cast(h.DefaultMultiVecQtAP, C.c_void_p).value
    for nm in ("OPS_Setup", "EigenSolverSetup_GCG", "TestEigenSolverGCG"):
        assert ref.ref_address_of(nm.encode()) != C.cast(getattr(h, nm), C.c_void_p).value, nm


GCG = load_golden("gcg.json")


@pytest.mark.parametrize("key", sorted(GCG.keys()))
def test_gcg_driver_on_oracle_matches_reference_run(oracle, key):
    c = GCG[key]
    args = ["-nevConv", c["nev"]]
    if c["nev_max"]:
        args += ["-nevMax", c["nev_max"]]
    if c["block"]:
        args += ["-blockSize", c["block"]]
    if c.get("nev_init"):
        args += ["-nevInit", c["nev_init"]]
    args += c["extra"]
    ev, res = gcg_on(oracle, c["kind"], c["size"], args, K=6, R0=1.5, R1=2.0, seed=12345)
    assert res.nevConv == c["nevConv"]
    if "autoshift" not in key and "order2" not in key:
        # with the automatic shift the W systems are nearly singular by construction (sigma = -lambda_C + 1% of the gap,
        # ops_eig_sol_gcg.c:483-485) and the iteration count of the reference itself changes with the allocation history
        # of the process (12 or 22 for the same input); only the converged values are pinned there.  Same for the
        # second-order W variant, whose residuals hover around the tolerance for many iterations (28 or 46)
        assert abs(res.numIter - c["numIter"]) <= 1, (res.numIter, c["numIter"])
    ref = np.array(c["eval"])
    rel = np.max(np.abs(ev[:len(ref)] - ref) / np.abs(ref))
    assert rel < 1e-10, "Ritz values differ from the reference: %.3e" % rel


def test_lap3d_closed_form(oracle):
    ev, res = gcg_on(oracle, "lap3d", 14, ["-nevConv", 12])
    exact = lap3d_exact(14, res.nevConv)
    assert res.nevConv >= 12
    assert np.max(np.abs(ev[:res.nevConv] - exact) / exact) < 1e-10


@pytest.mark.parametrize("key", ["lap3d_20_nev20", "fe3d_20_nev20", "sio2_12_nev10"])
def test_cholesky_qr_orth_scheme(oracle, key):
    """Block Cholesky-QR orthonormalisation (method 'chol', the GPU-friendly variant of the MGS scheme)
    reaches the reference's Ritz values with the same number of locked pairs."""
    c = GCG[key]
    ev, res = gcg_on(oracle, c["kind"], c["size"], ["-nevConv", c["nev"], "-gcge_initX_orth_method", "chol",
                                                    "-gcge_compW_orth_method", "chol"], K=6, R0=1.5, R1=2.0, seed=12345)
    assert res.nevConv == c["nevConv"] and abs(res.numIter - c["numIter"]) <= 2
    ref = np.array(c["eval"])
    assert np.max(np.abs(ev[:len(ref)] - ref) / np.abs(ref)) < 1e-10


SHAPES = load_golden("gcg_shapes.json")


@pytest.mark.parametrize("key", ["c2shape_lap3d_24", "c3shape_fe3d_20"])
def test_gcg_driver_at_baseline_solver_shapes(oracle, key):
    """Block size / nevMax of BASELINE configs 2 and 3 on reduced grids (tests/golden/make_golden_shapes.py): workspace
    relations of ops_eig_sol_gcg.c:1275-1280,1641-1645 at block 64 / 128.  A whole block of pairs locks per iteration
    at these widths, so the converged count at exit depends on the iteration the wanted count is reached in; the
    wanted count, the iteration count and every commonly converged value are pinned (the C4 shape runs on the GPU)."""
    c = SHAPES[key]
    ev, res = gcg_on(oracle, c["kind"], c["size"], ["-nevConv", c["nev"], "-nevMax", c["nev_max"], "-blockSize", c["block"]], **c.get("kw", {}))
    assert res.nevConv >= c["nev"] and abs(res.numIter - c["numIter"]) <= 2, (res.nevConv, res.numIter, c["numIter"])
    k = min(res.nevConv, c["nevConv"])
    ref = np.array(c["eval"][:k])
    assert np.max(np.abs(ev[:k] - ref) / np.abs(ref)) < 1e-10
