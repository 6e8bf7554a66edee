"""Measurement of the LITERAL drop-in: the reference's compiled GCG / ModifiedGramSchmidt / BlockPCG
(oracle/_ref/libgcge_ref.so, built from /root/reference in the build container) over the OPS_HIP_Set slots.

    python tests/refstack_on_hip.py --size 256 --nev 50 --block 64 --nevmax 128 --flag 1

flag 1 = the back-end's fused CG behind ops->MultiLinearSolver (test_app_ccs.c:109-120), flag 0 = the reference's
BlockPCG over the slots.  Prints one JSON line.  Test infrastructure (touches oracle/): lives under tests/."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="lap3d")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--nev", type=int, default=50)
    ap.add_argument("--block", type=int, default=64)
    ap.add_argument("--nevmax", type=int, default=128)
    ap.add_argument("--flag", type=int, default=1)
    ap.add_argument("--rng", type=int, default=1)
    ap.add_argument("--slot-timing", action="store_true", help="synchronise after every slot call and report wall time per slot and width class (stderr)")
    ap.add_argument("--prepare", action="store_true", help="gcge_hip_bpcg_prepare before the solve: the fused CG's blocks are created outside the timed region, "
                    "as the reference creates BlockPCG's (EigenSolverCreateWorkspace_GCG)")
    ap.add_argument("--verbose", action="store_true", help="let the reference print its own log and phase-time table (stdout, before the JSON line)")
    a = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import pyoracle as po
    from gcge_amd import HipBackend, make_problem
    from helpers import lap3d_exact
    ref = po.ref_lib()
    assert ref is not None, "oracle/_ref/libgcge_ref.so missing"
    hip = HipBackend()
    ops = C.c_void_p()
    hip.h.OPS_Create(C.byref(ops))
    hip.g.OPS_HIP_Set(ops)
    if a.flag == 1:
        hip.g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
        hip.g.gcge_hip_bpcg_setup(ops, 30, 1e-2, 1e-14, b"abs")
    hip.set_random_mode(a.rng, 20240601)
    A, B = make_problem(a.kind, a.size)
    mA = hip.matrix(A)
    mB = hip.matrix(B) if B is not None else None
    ref.ref_gcg_solve_foreign.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    if a.prepare and a.flag == 1:
        like = hip.ops.mv_create(a.block, mA)
        hip.g.gcge_hip_bpcg_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        ring = hip.g.gcge_hip_bpcg_prepare(ops, mA, like, a.block)
        assert ring >= 1, ring
        hip.ops.mv_destroy(like, a.block)
    if a.slot_timing:
        hip.g.gcge_hip_slot_timing(1)
    hip.g.gcge_hip_bpcg_time_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_double), C.c_int]
    hip.g.gcge_hip_bpcg_time_stats(None, None, 1)
    if a.verbose:
        ref.ref_set_verbose(1)
    ev = np.zeros(a.nevmax or 2 * a.nev)
    conv, it, sec = C.c_int(), C.c_int(), C.c_double()
    rc = ref.ref_gcg_solve_foreign(ops, mA, mB, a.nev, a.nevmax, a.block, 0, 1e-1, 1e-8, 500, a.flag,
                                   ev.ctypes.data_as(C.POINTER(C.c_double)), C.byref(conv), C.byref(it), C.byref(sec))
    assert rc == 0
    if a.slot_timing:
        buf = C.create_string_buffer(16384)
        hip.g.gcge_hip_slot_timing_report(buf, 16384)
        sys.stderr.write(buf.value.decode())
        hip.g.gcge_hip_slot_timing(0)
    out = {"stack": "reference GCG + ModifiedGramSchmidt + %s over OPS_HIP_Set slots" % ("HIP fused CG (flag 1)" if a.flag else "reference BlockPCG (flag 0)"),
           "kind": a.kind, "size": a.size, "n": int(A.nrows), "nev": a.nev, "block": a.block, "nevMax": a.nevmax,
           "cg_blocks_prepared": bool(a.prepare and a.flag == 1), "nev_converged": conv.value, "gcg_iterations": it.value, "seconds": sec.value,
           "eigenpairs_per_s": conv.value / sec.value}
    if a.flag == 1:   # how the fused CG ran underneath the reference's ComputeW (form, iterations, time inside the solver slot)
        its, cs_ = C.c_long(), C.c_double()
        hip.g.gcge_hip_bpcg_time_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_double), C.c_int]
        hip.g.gcge_hip_bpcg_time_stats(C.byref(its), C.byref(cs_), 0)
        for nm in ("recompute_iters", "device_scalar_iters", "implicit_r_iters", "surplus_iters"):
            getattr(hip.g, "gcge_hip_bpcg_" + nm).restype = C.c_long
        out["cg"] = {"iterations": its.value, "seconds": cs_.value, "recompute_iters": hip.g.gcge_hip_bpcg_recompute_iters(),
                     "device_scalar_iters": hip.g.gcge_hip_bpcg_device_scalar_iters(),
                     "implicit_r_iters": hip.g.gcge_hip_bpcg_implicit_r_iters(), "surplus_iters": hip.g.gcge_hip_bpcg_surplus_iters()}
    if a.kind == "lap3d":
        exact = lap3d_exact(a.size, conv.value) if a.size <= 64 else None
        if exact is None:
            c = np.sort(2.0 * np.cos(np.arange(1, a.size + 1) * np.pi / (a.size + 1)))[::-1][:48]
            exact = np.sort((6.0 - c[:, None, None] - c[None, :, None] - c[None, None, :]).ravel())[:conv.value]
        out["max_rel_err_vs_closed_form"] = float(np.max(np.abs(ev[:conv.value] - exact) / exact))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
