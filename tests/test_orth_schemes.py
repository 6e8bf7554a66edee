"""CPU: the block Cholesky-QR orthonormalisation scheme (what bench.py runs the device path with) against the
reference's block MGS inside whole eigensolves — same converged count, same Ritz values, iteration counts within
a few — over problem kinds, sizes and block shapes (a 32-case sweep of this kind is clean; this is a subset)."""
import numpy as np
import pytest

from helpers import gcg_on

CASES = [("lap3d", 8, 6, 0), ("lap3d", 10, 20, 8), ("lap3d", 14, 31, 0), ("fe3d", 8, 12, 0), ("fe3d", 12, 20, 8),
         ("fe1d", 300, 31, 0), ("sio2", 8, 20, 8), ("sio2", 11, 12, 0)]


@pytest.mark.parametrize("kind,size,nev,blk", CASES)
def test_cholesky_qr_tracks_block_mgs(oracle, kind, size, nev, blk):
    out = {}
    for orth in ("mgs", "chol"):
        args = ["-nevConv", nev, "-gcge_initX_orth_method", orth, "-gcge_compW_orth_method", orth]
        if blk:
            args += ["-blockSize", blk, "-nevMax", nev + 2 * blk]
        ev, res = gcg_on(oracle, kind, size, args, K=5, R0=1.5, R1=2.0, seed=7)
        out[orth] = (res.nevConv, res.numIter, ev[:nev].copy())
    (ca, ia, ea), (cb, ib, eb) = out["mgs"], out["chol"]
    assert ca >= nev and cb >= nev, (ca, cb)
    assert abs(ia - ib) <= max(3, ia // 5), (ia, ib)
    assert np.max(np.abs(ea - eb) / np.abs(ea)) < 1e-9


def test_cholesky_qr_with_the_reference_absolute_reorth_test(oracle, monkeypatch):
    """GCGE_ORTH_ABSOLUTE_TEST=1: the "chol" scheme decides about a further projection pass by the reference's absolute test
    (|c| < reorth_tol, src/ops_orth.c:315-323) instead of relative to the columns: more passes, the same converged pairs."""
    args = ["-nevConv", 12, "-gcge_initX_orth_method", "chol", "-gcge_compW_orth_method", "chol"]
    ev_rel, res_rel = gcg_on(oracle, "lap3d", 12, args)
    monkeypatch.setenv("GCGE_ORTH_ABSOLUTE_TEST", "1")
    ev_abs, res_abs = gcg_on(oracle, "lap3d", 12, args)
    assert res_rel.nevConv >= 12 and res_abs.nevConv >= 12
    assert abs(res_rel.numIter - res_abs.numIter) <= 2
    k = min(res_rel.nevConv, res_abs.nevConv)
    assert np.max(np.abs(ev_rel[:k] - ev_abs[:k]) / np.abs(ev_abs[:k])) < 1e-10
