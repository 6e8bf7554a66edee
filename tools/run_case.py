"""Run one eigenproblem through the HIP operator table and check it independently.

    python tools/run_case.py --kind fe3d --size 100 --nev 100 --block 128 --nevmax 256      # BASELINE config 3
    python tests/case_vs_reference.py --kind lap3d --size 50 --nev 20 --block 20 --rng 0   # config 1 next to the CPU reference
    python tools/run_case.py --kind sio2 --size 86 --nev 100 --block 64 --K 60              # config 5 shape on one GPU

Prints one JSON line: timing by phase, converged count, the relative residuals
||A x - lambda B x|| / (lambda ||B x||) recomputed through the slots (not taken from the solver).
A measurement tool, not part of the product path; it never touches oracle/ — tests/case_vs_reference.py wraps it and adds
the run of the compiled reference on the same input."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None, post=None):
    """post(out, a, A, B, ev, k): optional hook that may add entries to the result dictionary before it is printed."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="fe3d")
    ap.add_argument("--size", type=int, default=100)
    ap.add_argument("--nev", type=int, default=100)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--nevmax", type=int, default=0)
    ap.add_argument("--orth", default="chol")
    ap.add_argument("--flag", type=int, default=1, help="1: fused device block CG, 0: BlockPCG over the slots")
    ap.add_argument("--rng", type=int, default=1, help="0: reference rand() stream, 1: device generator")
    ap.add_argument("--K", type=int, default=6)
    ap.add_argument("--R0", type=float, default=1.5)
    ap.add_argument("--R1", type=float, default=2.0)
    ap.add_argument("--petsc", nargs="+", default=None, metavar="FILE",
                    help="PETSc binary Mat file(s): A [B] (the reference's -filename_matA / -filename_matB) instead of a generator")
    ap.add_argument("--mtx", nargs="+", default=None, metavar="FILE",
                    help="Matrix Market coordinate file(s): A [B] (the form SuiteSparse ships the reference's SiO2 / Ga41As41H72 ... in)")
    ap.add_argument("--box", default=None, metavar="NX,NY,NZ",
                    help="7-point Laplacian on an NX x NY x NZ grid instead of --kind/--size (e.g. 512,512,64: the slab one of 8 ranks owns at BASELINE config 4)")
    ap.add_argument("--extra", nargs="*", default=[])
    a = ap.parse_args(argv)

    import numpy as np
    import torch  # noqa: F401  (one libamdhip64 for torch and the extension)
    from gcge_amd import HipBackend, load_matrix_market, load_petsc_binary, make_problem, run_gcg
    t0 = time.perf_counter()
    if a.box:
        from gcge_amd.lib import CSR, host_lib
        nx, ny, nz = (int(t) for t in a.box.split(","))
        A, B = CSR(), None
        if host_lib().gcge_problem_lap3d_box(C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int64(0), C.c_int64(-1), C.byref(A)) != 0:
            raise RuntimeError("gcge_problem_lap3d_box failed")
        a.kind, a.size = "lap3d_box:" + a.box, A.nrows
    elif a.mtx:
        A = load_matrix_market(a.mtx[0])
        B = load_matrix_market(a.mtx[1]) if len(a.mtx) > 1 else None
        a.kind, a.size = "mtx:" + os.path.basename(a.mtx[0]), A.nrows
    elif a.petsc:
        A = load_petsc_binary(a.petsc[0])
        B = load_petsc_binary(a.petsc[1]) if len(a.petsc) > 1 else None
        a.kind, a.size = "petsc:" + os.path.basename(a.petsc[0]), A.nrows
    else:
        A, B = make_problem(a.kind, a.size, K=a.K, R0=a.R0, R1=a.R1, seed=12345)
    t_gen = time.perf_counter() - t0
    hip = HipBackend()
    hip.g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
    mA = hip.matrix(A)
    mB = hip.matrix(B) if B is not None else None
    hip.set_random_mode(a.rng, 20240601)
    if a.flag:
        hip.g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
    args = ["-nevConv", a.nev]
    if a.nevmax:
        args += ["-nevMax", a.nevmax]
    if a.block:
        args += ["-blockSize", a.block]
    args += ["-gcge_initX_orth_method", a.orth, "-gcge_compW_orth_method", a.orth] + list(a.extra)
    ev, res, evec = run_gcg(hip.ops_handle, mA, mB, args, flag=a.flag, keep_evec=True)

    # independent check: R = A X - B X diag(lambda) through the slots
    ops, n, k = hip.ops, A.nrows, res.nevConv
    ax = ops.mv_create(k, mA)
    bx = ops.mv_create(k, mA)
    ops.spmm(mA, evec, ax, (0, 0), (k, k))
    ops.spmm(mB, evec, bx, (0, 0), (k, k))          # mat == NULL copies
    nb = np.sqrt(ops.inner_prod("D", bx, bx, (0, 0), (k, k)))
    coef = np.zeros((k, k))
    coef[np.arange(k), np.arange(k)] = -ev[:k]
    ops.lincomb(bx, ax, (0, 0), (k, k), np.asfortranarray(coef).ravel(order="F"), k, beta=np.ones(1), incb=0)
    nr = np.sqrt(ops.inner_prod("D", ax, ax, (0, 0), (k, k)))
    relres = nr / (np.abs(ev[:k]) * nb)
    ops.mv_destroy(ax, k)
    ops.mv_destroy(bx, k)

    out = {"kind": a.kind, "size": a.size, "n": int(n), "nnz": int(A.nnz), "generalized": B is not None, "nev": a.nev,
           "block": res.block_size, "nevMax": res.nevMax, "orth": a.orth, "fused_cg": bool(a.flag),
           "nev_converged": int(res.nevConv), "gcg_iterations": int(res.numIter), "seconds": res.seconds,
           "eigenpairs_per_s": res.nevConv / res.seconds, "matrix_build_seconds": t_gen,
           "phase_seconds": {q: getattr(res.timing, q) for q in ("initX", "checkconv", "compP", "compRR", "compRV", "compW", "linsol", "total")},
           "lambda_first": ev[:3].tolist(), "lambda_last": float(ev[k - 1]),
           "max_rel_residual_recomputed": float(relres.max())}
    if post is not None:
        post(out, a, A, B, ev, k)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
