#!/usr/bin/env python3
"""bench.py — headline benchmark of the GCG hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is ONE full eigensolve (GCG to convergence) of BASELINE.json config 2:
3-D 7-point Laplacian 256^3 (n = 16 777 216, CSR), nev = 50, block = 64, nevMax = 128,
standard problem, harness-default parameters (test/test_eig_sol_gcg.c:33-49,98-115 of
the reference), fused device block-CG behind ops->MultiLinearSolver, matrix and all
blocks of vectors resident in HBM before the timed region starts.
  value    = converged eigenpairs per second over the K timed solves (whole job)
  roofline = K1 CSR SpMM: algorithmic bytes (12 nnz + 4(n+1) + 16 n m) of the m = block
             launches / their average duration from HIP events inside the timed region
  cpu_baseline = the reference's own CPU path (oracle/_ref, kind "reference") or our C
             restatement (kind "port") on a bounded sample of the same workload.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--size", type=int, default=256, help="grid points per direction (n = size^3)")
    ap.add_argument("--nev", type=int, default=50)
    ap.add_argument("--block", type=int, default=64)
    ap.add_argument("--nevmax", type=int, default=128)
    ap.add_argument("--cpu-size", type=int, default=40, help="grid size of the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--orth", default="chol", help="block orthonormalisation scheme for X and W: chol | mgs | bgs")
    return ap.parse_args()


# fabric bytes per 64-column launch at C2 (256^3, chain2 kernels) from profiles/r01_bench/12_cg_pass_pmc.txt:
# 4 passes x (TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B)
PMC_PASS1 = 4 * (2.33868e7 * 128 + 2048 * 64)
PMC_PASS2 = 4 * (4.16776e7 * 128 + 6.7111e7 * 64)      # 15_cg_pass_pmc_nt_residual_loads.txt (r read non-temporally)


def cpu_baseline(args):
    """Reference CPU path on a bounded sample: same solver configuration, smaller grid."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle as po
    from gcge_amd.lib import make_problem, run_gcg
    N = args.cpu_size
    A, _ = make_problem("lap3d", N)
    sample = "Lap3D %d^3 (n=%d), nev=%d, block=%d, nevMax=%d, same parameters" % (N, A.nrows, args.nev, args.block, args.nevmax)
    ref = po.ref_lib()
    if ref is not None:
        # the reference is serial C over threaded MKL; more than ~16 threads only adds fork/join cost to its
        # skinny BLAS calls, so pin the count and report exactly that as `cores`
        cores = min(16, os.cpu_count() or 1)
        try:
            mkl = C.CDLL("libmkl_rt.so.1", mode=C.RTLD_GLOBAL)
            mkl.MKL_Set_Num_Threads(C.c_int(cores))
            cores = int(mkl.MKL_Get_Max_Threads())
        except (OSError, AttributeError):
            cores = os.cpu_count() or 1
        ev, conv, it, sec = po.ref_gcg(A, None, args.nev, nev_max=args.nevmax, block=args.block)
        return {"value": conv / sec, "unit": "eigenpairs/s", "cores": cores, "kind": "reference",
                "sample": sample + "; stock serial app_ccs build, threaded MKL BLAS/LAPACK; %d GCG its, %.1f s" % (it, sec)}
    ops = po.make_ops()
    po.oracle_lib().oracle_set_threads(os.cpu_count() or 1)
    m = po.ccs_from_csr(A)
    ev, res = run_gcg(ops, C.byref(m), None, ["-nevConv", args.nev, "-nevMax", args.nevmax, "-blockSize", args.block])
    return {"value": res.nevConv / res.seconds, "unit": "eigenpairs/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample": sample + "; oracle/cpu_backend.c, OpenMP over block columns; %d GCG its, %.1f s" % (res.numIter, res.seconds)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GCGE_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo transport staged through the host — a way to run
    # the multi-rank code path on a one-GPU box (numbers from it mean nothing)
    rehearse = world > 1 and os.environ.get("GCGE_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from gcge_amd import HipBackend, make_problem, run_gcg
    from gcge_amd import dist as gdist
    hip = HipBackend(device=local_rank)
    g = hip.g
    g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
    g.gcge_hip_profile_enable.argtypes = [C.c_int]
    g.gcge_hip_profile_spmm.restype = C.c_long
    g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    g.gcge_hip_mat_patterns.argtypes = [C.c_void_p]
    g.gcge_hip_mat_pattern_chain.argtypes = [C.c_void_p]

    N = args.size
    n_global = N ** 3
    # weak scaling: every rank owns size^3 rows of a grid that stays as cube-like as the rank count allows
    # (1: N^3, 2: N x N x 2N, 4: N x 2N x 2N, 8: (2N)^3 = BASELINE config 4), cut into slabs along the last index
    dims = gdist.weak_scaling_box(N, world)
    if world > 1:
        comm = gdist.install(hip, dist, rank, world, stage_through_host=rehearse)
        A, mat = gdist.lap3d_slab(hip, dims, rank, world, comm)
        n_global = dims[0] * dims[1] * dims[2]
    else:
        A, _ = make_problem("lap3d", N)
        mat = hip.matrix(A)
    if world > 1:
        # communicator set-up (RCCL creates its point-to-point channels lazily on first use): one 2-column halo
        # exchange and one tiny all-reduce before anything is timed
        wv, wy = hip.ops.mv_create(2, mat), hip.ops.mv_create(2, mat)
        hip.ops.set_random(wv, 0, 2)
        hip.ops.spmm(mat, wv, wy, (0, 0), (2, 2))
        hip.ops.inner_prod("D", wv, wy, (0, 0), (2, 2))
        hip.ops.mv_destroy(wv, 2)
        hip.ops.mv_destroy(wy, 2)
    # memory set-up, like the matrix upload: hipMalloc of the 17-34 GB work blocks costs 0.2-0.3 s each on a fresh
    # process, so the blocks one solve needs (V, eigenvectors, 3 work blocks, 3 CG blocks) are allocated once here and
    # handed back to the back-end's size-keyed pool, from which the solver's MultiVecCreateByMat calls take them
    cols = [args.nevmax + 2 * args.block, args.nevmax] + [args.block] * 6
    # + the direction ring of the fused CG (up to 15 more blocks of `block` columns, as memory allows: block_pcg.hip)
    free_b, _ = torch.cuda.mem_get_info()
    per_col = 8 * (A.nrows + 2 * dims[0] * dims[1] * (world > 1)) * 1.02
    spare = free_b - per_col * sum(cols) - (14 << 30)
    cols += [args.block] * max(0, min(15, int(spare // (per_col * max(64, args.block)))))
    warm = [hip.ops.mv_create(c, mat) for c in cols]
    for c, w in zip(cols, warm):
        hip.ops.mv_destroy(w, c)
    hip.sync()
    hip.set_random_mode(1, 20240601)          # device generator: 2e9 rand() calls would dominate at this n
    g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
    solver_args = ["-nevConv", args.nev, "-nevMax", args.nevmax, "-blockSize", args.block,
                   "-gcge_initX_orth_method", args.orth, "-gcge_compW_orth_method", args.orth]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_gcg(hip.ops_handle, mat, None, solver_args, flag=1)
    g.gcge_hip_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    conv_total, iters, last = 0, 0, None
    for _ in range(args.steps):
        ev, res = run_gcg(hip.ops_handle, mat, None, solver_args, flag=1)
        conv_total += res.nevConv
        iters += res.numIter
        last = (ev, res)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HIP-event intervals the back-end recorded on its own stream, by kind: 0 = MatDotMultiVec products (K1, plain or
    # with the column sums), 2 / 3 = first / second pass of a block-CG iteration (the product recomputed, never stored)
    g.gcge_hip_profile_kind.restype = C.c_long
    g.gcge_hip_profile_kind.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def prof(kind, ncols):
        ms_, by_ = C.c_double(), C.c_double()
        c_ = g.gcge_hip_profile_kind(kind, ncols, C.byref(ms_), C.byref(by_))
        return int(c_), ms_.value, by_.value

    ci, ai = C.c_long(), C.c_long()
    g.gcge_hip_bpcg_column_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
    g.gcge_hip_bpcg_column_stats(C.byref(ci), C.byref(ai))
    stats = {k: prof(k, args.block) for k in (0, 2, 3)}
    spmm_ms_all = sum(prof(k, 0)[1] for k in (0, 2, 3))
    g.gcge_hip_profile_enable(0)

    if rank == 0:
        ev, res = last
        import numpy as np
        # parity guard inside the bench: converged Ritz values vs the closed-form spectrum
        cs = [np.sort(2.0 * np.cos(np.arange(1, d + 1) * np.pi / (d + 1)))[::-1][:48] for d in dims]
        small = np.sort((6.0 - cs[0][:, None, None] - cs[1][None, :, None] - cs[2][None, None, :]).ravel())
        rel = float(np.max(np.abs(ev[:res.nevConv] - small[:res.nevConv]) / small[:res.nevConv]))
        npat = g.gcge_hip_mat_patterns(mat)
        chain = g.gcge_hip_mat_pattern_chain(mat)
        kbase = ("spmm_pattern", "spmm_pattern_chain", "spmm_pattern_chain2")[chain] if npat > 0 else "spmm_pad8"
        npass = (args.block + 15) // 16 if npat > 0 else 1

        def roof(kind, what, traffic=None, note="no PMC profile for this shape"):
            # `achieved` prices a launch at its ALGORITHMIC bytes: kind 0: SURVEY.md 8(d), 12 B per
            # non-zero + row pointers + X read + Y written; kind 2: matrix + p read; kind 3: matrix + p, r read + r,
            # p_new written.  On the pattern path the matrix itself is streamed as 2 B per row, so the bytes actually
            # moved are lower; the block streams (8 n m each) dominate either way.  One "launch" = the `npass` kernel
            # launches of 16 columns that make up one m-column operation (rocprof lists the 16-column launches).
            c_, ms_, by_ = stats[kind]
            if c_ == 0:
                return None
            ach = (by_ / c_) / (ms_ / c_ * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": "%s<7,%d> x %d passes of 16 columns: %s (%d row patterns, m=%d)"
                                              % (kbase, kind if kind else 0, npass, what, npat, args.block),
                    "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                    "traffic_note": note, "launches": c_, "avg_launch_ms": ms_ / c_, "alg_bytes_per_launch": by_ / c_,
                    "share_of_step": ms_ * 1e-3 / elapsed if elapsed > 0 else None}

        # HBM/fabric bytes per launch from separate rocprofv3 --pmc passes at this shape (not collected live: counters
        # need their own runs): TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B, gfx950 correction of MI355X_MICROARCH.md
        pmc_ok = npat > 0 and N == 256 and args.block == 64 and world == 1 and chain == 2
        r_k1 = roof(0, "Y = A X, K1 (MatDotMultiVec)",
                    4 * (2.35094e7 * 128 + 3.35544e7 * 64) if pmc_ok else None,
                    "4 passes x (2.351e7 x 128 B reads + 3.355e7 x 64 B writes), profiles/r01_spmm_explore/23_chain2_pmc.log"
                    if pmc_ok else "no PMC profile for this shape")
        r_p1 = roof(2, "CG pass 1, p.Ap and |Ap|^2 without storing Ap", PMC_PASS1 if pmc_ok else None,
                    "profiles/r01_bench/12_cg_pass_pmc.txt" if pmc_ok else "no PMC profile for this shape")
        r_p2 = roof(3, "CG pass 2, Ap recomputed + r -= alpha Ap, p' = r + beta p",
                    PMC_PASS2 if pmc_ok and PMC_PASS2 else None,
                    "profiles/r01_bench/15_cg_pass_pmc_nt_residual_loads.txt" if pmc_ok and PMC_PASS2 else "no PMC profile for this shape")
        cands = [r for r in (r_k1, r_p1, r_p2) if r is not None]
        dominant = max(cands, key=lambda r: r["share_of_step"]) if cands else None
        out = {
            "metric": "converged eigenpairs/sec (GCG, 3D Laplacian n=%d, block=%d)" % (n_global, args.block),
            "value": conv_total / elapsed, "unit": "eigenpairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Lap3D %d^3 rows per GPU, grid %dx%dx%d (7-pt, CSR, n=%d global), nev=%d, block=%d, nevMax=%d, B=NULL, "
                                   "tol abs 1e-1 rel 1e-8, fused device block-CG (30 its, rate 1e-2), X/W orthonormalisation '%s', device RNG start block"
                                   % (N, dims[0], dims[1], dims[2], n_global, args.nev, args.block, args.nevmax, args.orth),
                       "gcg_iterations": iters, "nev_converged": conv_total,
                       "max_rel_err_vs_closed_form": rel,
                       "cg_active_column_fraction": (ai.value / ci.value) if ci.value else None,
                       "phase_seconds": {k: getattr(res.timing, k) for k in ("initX", "checkconv", "compP", "compRR", "compRV", "compW", "linsol", "total")}},
            # the dominant kernel of the step; the K1 product alone (the north-star figure) and the other CG pass follow
            "roofline": dominant,
            "roofline_k1_spmm": r_k1, "roofline_cg_pass1": r_p1, "roofline_cg_pass2": r_p2,
            "spmm_share_of_step": spmm_ms_all * 1e-3 / elapsed if elapsed > 0 else None,
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
