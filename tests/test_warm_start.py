"""Warm start: ops->EigenSolver with nevGiven start vectors (reference src/ops_eig_sol_gcg.c:101-158: the given columns
are copied into V, the rest of X is random, everything is orthonormalised).  Fixtures from the compiled reference
(tests/golden/make_golden_warm.py); CPU: host driver over the oracle back-end, GPU: over the HIP back-end."""
import numpy as np
import pytest

from gcge_amd.lib import make_problem, run_gcg
from helpers import WARM_MODES, load_golden, sine_start_block

WARM = load_golden("warm.json")


def _warm_run(be, c, flag=0, extra=()):
    A, B = make_problem(c["kind"], c["size"])
    mA = be.matrix(A)
    mB = be.matrix(B) if B is not None else None
    nev_max = 2 * c["nev"]
    X0 = np.zeros((A.nrows, nev_max))
    X0[:, :c["nevGiven"]] = sine_start_block(c["size"], WARM_MODES, c["eps"], c["seed"])
    evec = be.mv_from_numpy(mA, X0)
    ev, res = run_gcg(be.ops_handle, mA, mB, ["-nevConv", c["nev"]] + list(extra), flag=flag, given=(evec, c["nevGiven"]))
    # the eigenvectors come back in the caller's block: check the first pair against the matrix
    V = be.mv_to_numpy(evec, A.nrows, 0, res.nevConv)
    be.ops.mv_destroy(evec, nev_max)
    return A, B, ev, res, V


def _check(c, A, B, ev, res, V, count_slack):
    from helpers import csr_to_scipy
    assert res.nevConv == c["nevConv"]
    assert abs(res.numIter - c["numIter"]) <= count_slack, (res.numIter, c["numIter"])
    ref = np.array(c["eval"])
    assert np.max(np.abs(ev[:res.nevConv] - ref) / np.abs(ref)) < 1e-10
    S = csr_to_scipy(A)
    BV = csr_to_scipy(B) @ V if B is not None else V
    R = S @ V - BV * ev[:res.nevConv]
    rel = np.linalg.norm(R, axis=0) / (np.abs(ev[:res.nevConv]) * np.linalg.norm(BV, axis=0))
    assert rel.max() < 1e-6, rel.max()      # harness tolerance 1e-8 relative to |lambda| ||x||_B with ||x||_B = 1


@pytest.mark.parametrize("key", sorted(WARM))
def test_warm_start_on_oracle_matches_reference(oracle, key):
    c = WARM[key]
    A, B, ev, res, V = _warm_run(oracle, c)
    _check(c, A, B, ev, res, V, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(WARM))
def test_warm_start_on_hip_matches_reference(hip, key):
    c = WARM[key]
    hip.set_random_mode(0, 0)          # the reference's rand() stream for the random part of X
    A, B, ev, res, V = _warm_run(hip, c)
    _check(c, A, B, ev, res, V, 2)
