"""Drop-in boundary, both directions, on the CPU (needs the compiled reference, oracle/_ref):
  (1) the REFERENCE's solver stack (its GCG, its ModifiedGramSchmidt, its BlockPCG, its OPS_Setup
      defaults) drives a back-end written against OUR operator table — the CPU oracle, which is
      slot-for-slot what the HIP back-end is checked against on the GPU;
  (2) OUR solver stack drives the reference's app_ccs back-end.
Either mix must reproduce the reference-only run."""
import ctypes as C

import numpy as np
import pytest

import pyoracle as po
from helpers import load_golden
from gcge_amd.lib import host_lib, make_problem, run_gcg

pytestmark = pytest.mark.skipif(po.ref_lib() is None, reason="oracle/_ref not built (needs /root/reference)")
GCG = load_golden("gcg.json")


@pytest.mark.parametrize("key", ["lap3d_12_nev10", "fe3d_12_nev10"])
def test_reference_solver_drives_our_backend(key):
    c = GCG[key]
    ref = po.ref_lib(); h = host_lib(); o = po.oracle_lib()
    ops = C.c_void_p()
    h.OPS_Create(C.byref(ops))          # our table ...
    o.OPS_ORACLE_Set(ops)               # ... filled by a back-end written against gcge_ops.h
    ref.ref_use_foreign_backend(ops)
    try:
        A, B = make_problem(c["kind"], c["size"])
        ev, conv, it, sec = po.ref_gcg(A, B, c["nev"])
    finally:
        ref.ref_use_foreign_backend(None)
    assert conv == c["nevConv"] and abs(it - c["numIter"]) <= 1
    refv = np.array(c["eval"])
    assert np.max(np.abs(ev[:len(refv)] - refv) / np.abs(refv)) < 1e-10


@pytest.mark.parametrize("key", ["lap3d_12_nev10", "fe3d_12_nev10"])
def test_our_solver_drives_reference_backend(key):
    c = GCG[key]
    ref = po.ref_lib()
    ref.ref_make_ccs_ops.restype = C.c_void_p
    ops = C.c_void_p(ref.ref_make_ccs_ops())       # OPS_Create + OPS_CCS_Set + OPS_Setup of the reference
    A, B = make_problem(c["kind"], c["size"])
    mA = po.ccs_from_csr(A); mB = po.ccs_from_csr(B) if B is not None else None
    ev, res = run_gcg(ops, C.byref(mA), C.byref(mB) if mB is not None else None, ["-nevConv", c["nev"]])
    assert res.nevConv == c["nevConv"] and abs(res.numIter - c["numIter"]) <= 1
    refv = np.array(c["eval"])
    assert np.max(np.abs(ev[:len(refv)] - refv) / np.abs(refv)) < 1e-10
