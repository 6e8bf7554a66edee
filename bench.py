#!/usr/bin/env python3
"""bench.py — headline benchmark of the GCG hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config c2|c4|c5] [--rehearse]

With --gpus N > 1 and no RANK in the environment bench.py starts the N ranks ITSELF — fresh child processes running
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, spawned before torch or the GPU is touched — relays
rank 0's JSON line and exits with the launcher's code (the reference's scaling runs are one command line too:
test/submit.sh:17-47, `mpirun -np k`).  Ranks started by a launcher (RANK set) run directly.  --rehearse puts every
rank on cuda:0 over gloo (one-GPU boxes; the numbers then mean nothing).

A "step" is ONE full eigensolve (GCG to convergence).  Default workload = BASELINE.json config 2: 3-D 7-point
Laplacian 256^3 (n = 16 777 216, CSR), nev = 50, block = 64, nevMax = 128, standard problem, tolerances and CG
parameters of the reference's harness (test/test_eig_sol_gcg.c:33-49,98-115) but NOT its defaults elsewhere: block
Cholesky-QR for X and W ("chol" instead of "mgs"), device RNG start block, the W systems behind ops->MultiLinearSolver
(flag 1) — all named in config.workload — matrix and all blocks of vectors resident in HBM before the timed region starts.  With
--gpus N every rank owns 256^3 rows of a box that stays as cube-like as N allows (weak scaling; N = 8: 512^3 =
the grid of BASELINE config 4); --config c4 additionally switches the solver shape to config 4's (nev 200,
block 128, nevMax 400); --config c5 is BASELINE config 5: the SiO2-like matrix (12th-order stencil + K = 2000
atom blocks on a 171^3 grid, n = 5 000 211, rows of very different length) at nev 100 / block 64 / nevMax 200
(test/test_eig_sol_SiO2_MAT.c:39-76 of the reference), the SAME matrix for every rank count, rows cut by non-zeros
("scaling": "strong").  Multi-GPU data path: RCCL called from C inside libgcge_hip.so (csrc/hip/rccl_comm.hip);
torch.distributed only hands rank 0's RCCL id to the other ranks and synchronises the timed region.
  value    = converged eigenpairs per second over the K timed solves (whole job)
  roofline = the HBM-bound kernel with the largest share of the step (the second pass of a block-CG iteration);
             roofline_k1_spmm = the plain product K1 (MatDotMultiVec), roofline_cg_pass1/2 = the two CG passes.
             achieved = algorithmic bytes / average launch duration from HIP events on the launch stream inside the
             timed region; frac_required prices the same launches at the bytes the kernel must move on the
             pattern path (2 B per row for the matrix instead of 12 B per non-zero);
             traffic = fabric bytes per launch from rocprofv3 PMC passes (tools/pmc_traffic.py ->
             profiles/pmc_traffic.json), quoted only while the HIP sources hash to what was profiled.
  roofline_gram / roofline_panel_update = the FP64-MFMA kernels (K2 / K3) in the solve: 2 n k m flop per launch / HIP-event
             duration, summed per shape (k, m) over the timed steps; the shape with the largest share of the step, against the
             78.6 TF dense FP64 matrix peak (vendor spec); dense_other_shapes = the next three.  pairs_wanted_per_s = nev x steps / elapsed
             (value counts every pair a solve converged).  upload_seconds = host arrays -> device matrix in all its forms.
  cpu_baseline = the reference's own CPU path, OpenMP build (oracle/_ref/libgcge_ref_omp.so: app_ccs.c:117-131
             under OPS_USE_OMP), all host cores, on BASELINE config 1 (Lap3D 50^3, nev 20, block 20) — with the
             GPU's time on that SAME config beside it (gpu_same_config); kind "port" = our C restatement when the
             compiled reference is absent.
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
    ap.add_argument("--size", type=int, default=0, help="grid points per direction (n = size^3); default 256 (c2, c4) / 171 (c5)")
    ap.add_argument("--nev", type=int, default=50)
    ap.add_argument("--block", type=int, default=64)
    ap.add_argument("--nevmax", type=int, default=128)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 = Lap3D, nev 50 / block 64 / nevMax 128 (BASELINE config 2); c3 = FE stiffness / mass pair (P1 on the Kuhn "
                         "triangulation, --size = interior nodes per direction, default 100: n = 10^6), generalised problem, nev 100 / block 128 / "
                         "nevMax 256 (BASELINE config 3; one rank); c4 = Lap3D, nev 200 / block 128 / nevMax 400; "
                         "c5 = SiO2-like matrix (--size = grid points per direction, default 171), nev 100 / block 64 / nevMax 200, rows split by nnz")
    ap.add_argument("--rehearse", action="store_true",
                    help="every rank on cuda:0, gloo transport staged through the host (multi-rank code path on a one-GPU box)")
    ap.add_argument("--atoms", default="2000,2.0,5.0", help="c5: K,R0,R1 of the SiO2-like generator (SURVEY 8d)")
    ap.add_argument("--cpu-size", type=int, default=50, help="grid size of the CPU baseline (50 = BASELINE config 1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dense-shapes", action="store_true", help="after the timed steps, print the in-solve rate of the Gram / panel-update kernels per shape (k, m) on stderr")
    ap.add_argument("--orth", default="chol", help="block orthonormalisation scheme for X and W: chol | mgs | bgs")
    ap.add_argument("--plain-steps", type=int, default=2,
                    help="with --amg: additional solves with the plain fused block CG (30 iterations) after the timed region -> plain_block_cg "
                         "(the configuration rounds 1-4 reported as value)")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs after the timed region (plain_block_cg, roofline_k1_c5, roofline_k1_permuted, dropin_reference_stack)")
    ap.add_argument("--cpu-like-size", type=int, default=48, help="grid size of the like-for-like CPU sample (config 2's solver shape on the CPU reference)")
    ap.add_argument("--amg", type=int, default=-1,
                    help="levels (>= 2) of the multigrid hierarchy: the W systems are solved by BlockAMG (one V-cycle, fused block CG as the "
                         "smoother; reference src/ops_lin_sol.c:466-715 set up as test/test_eig_sol_SiO2_MAT.c:96-128) instead of 30 block-CG "
                         "iterations; 0 = plain block CG; default: 6 for configs c2 / c3 / c4, 5 for c5 (one GPU or row slabs)")
    ap.add_argument("--amg-cycles", type=int, default=1, help="V-cycles per call of BlockAMG (the reference's SiO2 set-up: 1)")
    ap.add_argument("--amg-scale", type=float, default=0.0, help="coarse operators A_{l+1} = scale P^T A_l P (0: the back-end's default 0.5, include/gcge_multigrid.h)")
    ap.add_argument("--amg-smooth", default=None, help="CG smoothing iterations before and after the coarse correction: finest level, coarser levels "
                                                       "(default 3,4; config c5: 8,24 — the coarse levels carry the projected atoms, 2000 outlying eigenvalues, "
                                                       "and must be smoothed long enough to resolve them: profiles/r05_amg/13_...)")
    a = ap.parse_args()
    if a.config == "c4":
        a.nev, a.block, a.nevmax = 200, 128, 400
    if a.config == "c5":
        a.nev, a.block, a.nevmax = 100, 64, 200
    if a.config == "c3":
        a.nev, a.block, a.nevmax = 100, 128, 256
    if a.size <= 0:
        a.size = 171 if a.config == "c5" else 100 if a.config == "c3" else 256
    if a.amg < 0:
        # (row slabs coarsen by themselves: csrc/hip/multigrid.hip; config 5 on one GPU: 14.0 s with BlockAMG 8 / 24 against 44.7 s with
        #  30 plain CG iterations, profiles/r05_amg/13_...; row slabs of config 5 the same: cut by non-zeros on ANY plane boundary, every
        #  rank pairs its own planes — the rehearsals on two slabs sharing a GPU: 96^3 22.8 s with 5 levels against 66.0 s with 30 plain CG
        #  iterations, 88^3 with an odd cut on level 1 44 outer iterations like the even cuts, profiles/r05_amg/19_..., 22_...)
        a.amg = 6 if (a.config in ("c2", "c3", "c4")) else (5 if a.config == "c5" else 0)
    if a.amg_smooth is None:
        a.amg_smooth = "8,24" if a.config == "c5" else "3,4"
    if a.rehearse:
        os.environ["GCGE_BENCH_REHEARSE"] = "1"
    return a


def self_launch(args):
    """--gpus N > 1 without a launcher: run the N ranks as fresh child processes (nothing in THIS process has touched
    torch or the GPU), relay rank 0's stdout (the one JSON line) and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def host_cores():
    """Cores this process may use: the affinity mask, cut by a cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = max(1, min(n, int(float(q[0]) / float(q[1]) + 0.5)))
    except (OSError, ValueError, IndexError):
        pass
    return n


def hip_source_hash():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gcge_amd", "csrc", "hip")
    # the sources that define the profiled kernels (SpMM / CG passes): an edit to the eigensolver, to another matrix form or
    # to the CG's host loop does not change what a launch of these kernels moves
    for f in sorted(os.listdir(d)):
        if f in ("spmm_pattern.hip", "spmm_ring.hip"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def pmc_traffic(kernel, N, m):
    """(bytes per m-column operation, note) from profiles/pmc_traffic.json, or (None, why not)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    if t.get("source_sha256") != hip_source_hash():
        return None, "profiles/pmc_traffic.json was measured on other kernel sources (hash differs): not quoted"
    sh = t.get("shape", {})
    if sh.get("N") != N or sh.get("m") != m:
        return None, "profiles/pmc_traffic.json holds N=%s m=%s" % (sh.get("N"), sh.get("m"))
    k = t.get("kernels", {}).get(kernel)
    if not k or "fabric_bytes_per_block_operation" not in k:
        return None, "kernel %s not in profiles/pmc_traffic.json" % kernel
    return k["fabric_bytes_per_block_operation"], "profiles/pmc_traffic.json (%s; sources %s)" % (t.get("bytes_rule", ""), t["source_sha256"][:12])


def pmc_traffic_c5(G, atoms, m):
    """Fabric bytes of the plane sweep per m-column product of BASELINE config 5's matrix (profiles/pmc_traffic.json, entry
    "c5"), quoted only while spmm_star.hip hashes to what was profiled."""
    import hashlib
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("c5")
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    if not t:
        return None, "no entry for config 5 in profiles/pmc_traffic.json"
    h = hashlib.sha256()
    for f in sorted(t.get("source_files", [])):
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "gcge_amd", "csrc", "hip", f), "rb").read())
    if h.hexdigest() != t.get("source_sha256"):
        return None, "profiles/pmc_traffic.json (c5) was measured on other kernel sources (hash differs): not quoted"
    sh = t.get("shape", {})
    if sh.get("G") != G or sh.get("atoms") != atoms or sh.get("m") != m:
        return None, "profiles/pmc_traffic.json (c5) holds G=%s atoms=%s m=%s" % (sh.get("G"), sh.get("atoms"), sh.get("m"))
    if "product_fabric_bytes" in t:            # round 5: every kernel of the product (sweep + dense blocks + listed rows)
        return t["product_fabric_bytes"], "all kernels of the product (%s); %s" % (t.get("profile"), t.get("bytes_rule"))
    k = next(iter(t["kernels"].values()))
    return k["fabric_bytes_per_block_operation"], "plane sweep only (%s); %s" % (t.get("profile"), t.get("bytes_rule"))


def cpu_baseline(args, hip):
    """The reference's CPU path on BASELINE config 1 (Lap3D 50^3, nev 20, block 20, nevMax 40), OpenMP build, all
    host cores; and the GPU on that same config."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle as po
    from gcge_amd.lib import make_problem, run_gcg
    N, nev = args.cpu_size, 20
    A, _ = make_problem("lap3d", N)
    cores = host_cores()
    sample = "BASELINE config 1: Lap3D %d^3 (n=%d), nev=20, block=20, nevMax=40, harness parameters" % (N, A.nrows)
    # the GPU on the same config (same start vectors: the reference's rand() stream; chol orthonormalisation + fused CG)
    g = hip.g
    mat = hip.matrix(A)
    hip.set_random_mode(0)
    g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
    sargs = ["-nevConv", nev, "-gcge_initX_orth_method", args.orth, "-gcge_compW_orth_method", args.orth]
    run_gcg(hip.ops_handle, mat, None, sargs, flag=1)
    C.CDLL(None).srand(0)
    ev_g, res_g = run_gcg(hip.ops_handle, mat, None, sargs, flag=1)
    hip.free_matrix(mat)
    gpu = {"value": res_g.nevConv / res_g.seconds, "unit": "eigenpairs/s", "seconds": res_g.seconds,
           "gcg_iterations": res_g.numIter, "nev_converged": res_g.nevConv}
    ref = po.ref_lib(omp=True)
    if ref is not None:
        try:                                   # libgomp was initialised when torch was imported: set the team size now
            C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(cores))
        except OSError:
            pass
        try:
            mkl = C.CDLL("libmkl_rt.so.1", mode=C.RTLD_GLOBAL)
            mkl.MKL_Set_Num_Threads(C.c_int(cores))
        except (OSError, AttributeError):
            pass
        # the host of a GPU box is shared: the faster of two runs (3.5 s and 12.6 s have both been seen for the same run)
        ev, conv, it, sec = po.ref_gcg(A, None, nev, omp=True)
        ev2, conv2, it2, sec2 = po.ref_gcg(A, None, nev, omp=True)
        secs = (sec, sec2)
        if sec2 < sec:
            ev, conv, it, sec = ev2, conv2, it2, sec2
        import numpy as np
        k = min(conv, res_g.nevConv)
        gpu["max_rel_diff_vs_cpu_reference"] = float(np.max(np.abs(ev_g[:k] - ev[:k]) / np.abs(ev[:k]))) if k > 0 else None
        if conv <= 0 or sec <= 0:
            raise RuntimeError("the CPU reference converged %d pairs in %.2f s" % (conv, sec))
        return {"value": conv / sec, "unit": "eigenpairs/s", "cores": cores, "kind": "reference",
                "sample": sample + "; OPS_USE_OMP build of app_ccs/app_lapack (oracle/Makefile ref_omp), OMP_NUM_THREADS = MKL threads = %d, "
                                   "MKL_THREADING_LAYER=GNU; %d GCG its, %d pairs, %.1f s (faster of two runs: %.1f / %.1f s)" % (cores, it, conv, sec, secs[0], secs[1]),
                "gpu_same_config": gpu}
    ops = po.make_ops()
    po.oracle_lib().oracle_set_threads(cores)
    m = po.ccs_from_csr(A)
    ev, res = run_gcg(ops, C.byref(m), None, ["-nevConv", nev])
    return {"value": res.nevConv / res.seconds, "unit": "eigenpairs/s", "cores": cores, "kind": "port",
            "sample": sample + "; oracle/cpu_backend.c, OpenMP over block columns; %d GCG its, %.1f s" % (res.numIter, res.seconds),
            "gpu_same_config": gpu}


def k1_c5_leg(hip, args):
    """roofline_k1_c5: the K1 product (MatDotMultiVec) of BASELINE config 5's matrix — SiO2-like, 171^3 grid, 3.5e8 non-zeros, rows of
    very different length — on 64 columns, outside any solve: generate + upload + 3 untimed + 20 timed products, HIP events on the
    launch stream (the back-end's own, gcge_hip_profile_kind).  Algorithmic bytes: SURVEY 8(d), 12 nnz + 4 (n + 1) + 16 n m."""
    import numpy as np
    from gcge_amd.lib import make_problem
    g = hip.g
    K, R0, R1 = "2000,2.0,5.0".split(",")
    G, m = 171, 64
    t0 = time.perf_counter()
    A, _ = make_problem("sio2", G, K=int(K), R0=float(R0), R1=float(R1), seed=12345)
    t1 = time.perf_counter()
    mat = hip.matrix(A)
    hip.sync()
    t2 = time.perf_counter()
    g.gcge_hip_mat_spmm_form.restype = C.c_char_p
    g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
    form = g.gcge_hip_mat_spmm_form(mat).decode()
    x, y = hip.ops.mv_create(m, mat), hip.ops.mv_create(m, mat)
    hip.set_random_mode(1, 77)
    hip.ops.set_random(x, 0, m)
    for _ in range(3):
        hip.ops.spmm(mat, x, y, (0, 0), (m, m))
    hip.sync()
    g.gcge_hip_profile_enable(1)
    nprod = 20
    for _ in range(nprod):
        hip.ops.spmm(mat, x, y, (0, 0), (m, m))
    hip.sync()
    ms_, by_ = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_kind(0, m, C.byref(ms_), C.byref(by_))
    g.gcge_hip_profile_enable(0)
    # parity inside the leg: 1 . (A x) = (A 1) . x needs every row and every entry once (A symmetric): column sums of y against
    # the row sums of A applied to x, through the slots
    ones = hip.ops.mv_create(2, mat)
    a1 = hip.ops.mv_create(2, mat)
    hip.g.gcge_hip_mv_from_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_long]
    o = np.ones((A.nrows, 2), order="F")
    hip.g.gcge_hip_mv_from_host(ones, 0, 2, o.ctypes.data_as(C.POINTER(C.c_double)), A.nrows)
    hip.ops.spmm(mat, ones, a1, (0, 0), (2, 2))
    lhs = hip.ops.inner_prod("N", ones, y, (0, 0), (1, 2))[0]          # 1 . y_j, j = 0, 1
    rhs = hip.ops.inner_prod("N", a1, x, (0, 0), (1, 2))[0]            # (A 1) . x_j
    sym = float(np.max(np.abs(lhs - rhs) / np.maximum(1e-300, np.abs(rhs))))
    for v, c in ((x, m), (y, m), (ones, 2), (a1, 2)):
        hip.ops.mv_destroy(v, c)
    traffic, note = pmc_traffic_c5(G, "2000,2.0,5.0", m)
    nnz, n = int(A.nnz), int(A.nrows)
    hip.free_matrix(mat)
    hip.h.gcge_csr_free(C.byref(A))
    avg_ms = ms_.value / cnt
    alg = by_.value / cnt
    return {"bound": "hbm", "kernel": form + ": Y = A X, K1 (MatDotMultiVec), m=%d" % m,
            "workload": "SiO2-like matrix of BASELINE config 5: %d^3 grid, 12th-order 37-point stencil + %s atom blocks (R = %s + %s u1 u2), n=%d, nnz=%d"
                        % (G, K, R0, R1, n, nnz),
            "achieved": alg / (avg_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / (avg_ms * 1e-3) / 1e9 / 8000.0,
            "traffic": traffic, "traffic_note": note, "launches": int(cnt), "avg_launch_ms": avg_ms, "alg_bytes_per_launch": alg,
            "generate_seconds": t1 - t0, "upload_seconds": t2 - t1, "symmetry_check_rel": sym}


def k1_plain_leg(hip, mat, A, m, what):
    """K1 alone on a resident matrix: 3 untimed + 20 timed products Y = A X on m columns, HIP events on the launch stream; algorithmic
    bytes per SURVEY 8(d): 12 nnz + 4 (n + 1) + 16 n m."""
    g = hip.g
    g.gcge_hip_mat_spmm_form.restype = C.c_char_p
    g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
    form = g.gcge_hip_mat_spmm_form(mat).decode()
    x, y = hip.ops.mv_create(m, mat), hip.ops.mv_create(m, mat)
    hip.ops.set_random(x, 0, m)
    for _ in range(3):
        hip.ops.spmm(mat, x, y, (0, 0), (m, m))
    hip.sync()
    g.gcge_hip_profile_enable(1)
    for _ in range(20):
        hip.ops.spmm(mat, x, y, (0, 0), (m, m))
    hip.sync()
    ms_, by_ = C.c_double(), C.c_double()
    cnt = g.gcge_hip_profile_kind(0, m, C.byref(ms_), C.byref(by_))
    g.gcge_hip_profile_enable(0)
    hip.ops.mv_destroy(x, m)
    hip.ops.mv_destroy(y, m)
    avg_ms, alg = ms_.value / cnt, by_.value / cnt
    return {"bound": "hbm", "kernel": "%s: Y = A X, K1 (MatDotMultiVec), m=%d" % (form, m), "what": what, "achieved": alg / (avg_ms * 1e-3) / 1e9,
            "peak": 8000.0, "unit": "GB/s", "frac": alg / (avg_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "launches": int(cnt),
            "avg_launch_ms": avg_ms, "alg_bytes_per_launch": alg, "n": int(A.nrows), "nnz": int(A.nnz)}


def k1_permuted_leg(hip, args):
    """roofline_k1_permuted: K1 on a matrix that shows NO grid in the order it arrives in — the SiO2-like matrix (128^3 grid, K = 839:
    config 5's density) under a random symmetric permutation, as a file with its own numbering would deliver it.  The upload recovers
    the grid from the star couplings and keeps P A P^T inside the handle (csrc/hip/reorder.hip); with the re-ordering switched off the
    same arrays take dense blocks + pad-8.  Both measured here (20 products each), checked against scipy on the permuted arrays."""
    import numpy as np
    import scipy.sparse as sp
    from gcge_amd.lib import CSR, make_problem
    g = hip.g
    G, K, m = 128, 839, 64
    A0, _ = make_problem("sio2", G, K=K, R0=2.0, R1=5.0, seed=12345)
    n, nnz = A0.nrows, int(A0.nnz)
    S = sp.csr_matrix((np.ctypeslib.as_array(A0.val, shape=(nnz,)), np.ctypeslib.as_array(A0.colidx, shape=(nnz,)),
                       np.ctypeslib.as_array(A0.rowptr, shape=(n + 1,))), shape=(n, n))
    p = np.random.default_rng(20240601).permutation(n)
    ip = np.empty(n, dtype=np.int32)
    ip[p] = np.arange(n, dtype=np.int32)
    Sp = S[p].tocsr()                      # rows in the new order ...
    Sp.indices = ip[Sp.indices]            # ... columns renamed ...
    Sp.has_sorted_indices = False
    Sp.sort_indices()                      # ... ascending inside every row again
    hip.h.gcge_csr_free(C.byref(A0))
    rp, ci, va = (np.ascontiguousarray(Sp.indptr, dtype=np.int32), np.ascontiguousarray(Sp.indices, dtype=np.int32),
                  np.ascontiguousarray(Sp.data, dtype=np.float64))
    A = CSR(n, n, 0, nnz, rp.ctypes.data_as(C.POINTER(C.c_int)), ci.ctypes.data_as(C.POINTER(C.c_int)), va.ctypes.data_as(C.POINTER(C.c_double)))
    g.gcge_hip_spmm_reorder_mode.argtypes = [C.c_int]
    g.gcge_hip_mat_row_order.restype = C.c_char_p
    g.gcge_hip_mat_row_order.argtypes = [C.c_void_p]
    out = {}
    xh = np.asfortranarray(np.random.default_rng(1).random((n, 2)) - 0.5)
    for tag, mode in (("rows_as_given", -1), ("row_order_chosen_at_upload", 0)):
        g.gcge_hip_spmm_reorder_mode(mode)
        t0 = time.perf_counter()
        mat = hip.matrix(A)
        hip.sync()
        up = time.perf_counter() - t0
        r = k1_plain_leg(hip, mat, A, m, "SiO2-like %d^3 (K = %d, n=%d, nnz=%d) under a random symmetric permutation" % (G, K, n, nnz))
        r["upload_seconds"] = up
        r["row_order"] = g.gcge_hip_mat_row_order(mat).decode()
        x = hip.mv_from_numpy(mat, xh)
        y = hip.ops.mv_create(2, mat)
        hip.ops.spmm(mat, x, y, (0, 0), (2, 2))
        want = Sp @ xh
        r["max_rel_err_vs_scipy"] = float(np.max(np.abs(hip.mv_to_numpy(y, n, 0, 2) - want)) / np.max(np.abs(want)))
        hip.ops.mv_destroy(x, 2)
        hip.ops.mv_destroy(y, 2)
        hip.free_matrix(mat)
        out[tag] = r
    g.gcge_hip_spmm_reorder_mode(0)
    res = out["row_order_chosen_at_upload"]
    res["rows_as_given"] = {k: out["rows_as_given"][k] for k in ("kernel", "avg_launch_ms", "frac", "achieved", "upload_seconds", "max_rel_err_vs_scipy")}
    return res


def dropin_leg(hip, mat, args):
    """dropin_reference_stack: ONE solve of the REFERENCE's own compiled GCG + ModifiedGramSchmidt (oracle/_ref/libgcge_ref.so: its
    OPS_Setup defaults, its LAPACK) over a second table only OPS_HIP_Set touched, fused block CG behind flag 1 — the literal
    drop-in north_star names (TestEigenSolverGCG's parameter flow: oracle/ref_shim.c gcg_solve_core), on the matrix of the timed
    region.  Test infrastructure: the reference library is the thing MEASURED here, beside the product, never part of it."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    ref = po.ref_lib()
    if ref is None:
        return {"error": "oracle/_ref/libgcge_ref.so not present"}
    g = hip.g
    ops = C.c_void_p()
    hip.h.OPS_Create(C.byref(ops))
    g.OPS_HIP_Set(ops)
    g.gcge_hip_bpcg_setup(ops, 30, 1e-2, 1e-14, b"abs")
    ref.ref_gcg_solve_foreign.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    like = hip.ops.mv_create(args.block, mat)
    g.gcge_hip_bpcg_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    g.gcge_hip_bpcg_prepare(ops, mat, like, args.block)        # the CG's blocks before the timer, as EigenSolverCreateWorkspace_GCG does
    hip.ops.mv_destroy(like, args.block)
    ev = np.zeros(args.nevmax)
    conv, it, sec = C.c_int(), C.c_int(), C.c_double()
    rc = ref.ref_gcg_solve_foreign(ops, mat, None, args.nev, args.nevmax, args.block, 0, 1e-1, 1e-8, 500, 1,
                                   ev.ctypes.data_as(C.POINTER(C.c_double)), C.byref(conv), C.byref(it), C.byref(sec))
    if rc != 0 or conv.value <= 0:
        return {"error": "ref_gcg_solve_foreign rc=%d converged %d" % (rc, conv.value)}
    N = args.size
    c = np.sort(2.0 * np.cos(np.arange(1, N + 1) * np.pi / (N + 1)))[::-1][:48]
    exact = np.sort((6.0 - c[:, None, None] - c[None, :, None] - c[None, None, :]).ravel())[:conv.value]
    return {"stack": "the reference's compiled GCG + ModifiedGramSchmidt + OPS_Setup defaults (oracle/_ref/libgcge_ref.so) over OPS_HIP_Set slots, "
                     "fused device block CG behind flag 1 (30 its), device RNG start block",
            "seconds": sec.value, "value": conv.value / sec.value, "unit": "eigenpairs/s", "nev_converged": conv.value,
            "gcg_iterations": it.value, "max_rel_err_vs_closed_form": float(np.max(np.abs(ev[:conv.value] - exact) / exact))}


def cpu_like_for_like(args, hip):
    """Second CPU sample, like for like: the compiled reference (OpenMP build, all cores) at BASELINE config 2's SOLVER SHAPE
    (nev 50, block 64, nevMax 128) on a smaller grid, and the GPU (this product: chol + fused CG, flag 1) on the same input."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from gcge_amd.lib import make_problem, run_gcg
    ref = po.ref_lib(omp=True)
    if ref is None:
        return None
    N = args.cpu_like_size
    A, _ = make_problem("lap3d", N)
    cores = host_cores()
    g = hip.g
    mat = hip.matrix(A)
    hip.set_random_mode(0)
    g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
    sargs = ["-nevConv", 50, "-nevMax", 128, "-blockSize", 64, "-gcge_initX_orth_method", args.orth, "-gcge_compW_orth_method", args.orth]
    run_gcg(hip.ops_handle, mat, None, sargs, flag=1)
    C.CDLL(None).srand(0)
    ev_g, res_g = run_gcg(hip.ops_handle, mat, None, sargs, flag=1)
    hip.free_matrix(mat)
    ev, conv, it, sec = po.ref_gcg(A, None, 50, nev_max=128, block=64, omp=True)
    k = min(conv, res_g.nevConv)
    return {"value": conv / sec, "unit": "eigenpairs/s", "cores": cores, "kind": "reference",
            "sample": "BASELINE config 2's solver shape (nev 50, block 64, nevMax 128, harness parameters) on Lap3D %d^3 (n=%d): OPS_USE_OMP build, "
                      "%d threads; %d GCG its, %d pairs, %.1f s" % (N, A.nrows, cores, it, conv, sec),
            "gpu_same_config": {"value": res_g.nevConv / res_g.seconds, "unit": "eigenpairs/s", "seconds": res_g.seconds,
                                "gcg_iterations": res_g.numIter, "nev_converged": res_g.nevConv,
                                "max_rel_diff_vs_cpu_reference": float(np.max(np.abs(ev_g[:k] - ev[:k]) / np.abs(ev[:k]))) if k > 0 else None}}


def slab_cut_planes(planes, world, levels):
    """Planes per cut unit for row slabs of a grid matrix behind BlockAMG with `levels` levels (1 without it)."""
    unit = 1
    while levels >= 2 and 2 * unit <= (1 << (levels - 1)) and 2 * unit * 4 * world <= planes:
        unit *= 2
    return unit


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))          # before torch / the GPU are touched in this process
    # stdout carries exactly ONE line, the JSON: libraries that greet on stdout (RCCL prints a version banner when a
    # communicator is created) go to stderr for the duration of the run
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GCGE_BENCH_REHEARSE=1 (--rehearse): every rank on cuda:0 with the gloo transport staged through the host — a way
    # to run the multi-rank code path on a one-GPU box (numbers from it mean nothing)
    rehearse = world > 1 and os.environ.get("GCGE_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d (start without RANK in the environment and bench.py launches its ranks itself)" % (world, args.gpus)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not rehearse and torch.cuda.device_count() < world:
            raise SystemExit("bench.py: %d ranks but %d GPUs visible (use --rehearse to share cuda:0 over gloo)" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from gcge_amd import HipBackend, make_problem, run_gcg
    from gcge_amd import dist as gdist
    hip = HipBackend(device=local_rank)
    if os.environ.get("GCGE_BENCH_STAR_MODE"):      # diagnostics: -1 = no matrix takes the plane sweep (gcge_hip_spmm_star_mode)
        hip.g.gcge_hip_spmm_star_mode(int(os.environ["GCGE_BENCH_STAR_MODE"]))
    g = hip.g
    g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
    g.gcge_hip_profile_enable.argtypes = [C.c_int]
    g.gcge_hip_profile_spmm.restype = C.c_long
    g.gcge_hip_profile_spmm.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    g.gcge_hip_mat_patterns.argtypes = [C.c_void_p]
    g.gcge_hip_mat_pattern_chain.argtypes = [C.c_void_p]

    N = args.size
    c5 = args.config == "c5"
    c3 = args.config == "c3"
    matB = None
    if c3 and world > 1:
        raise SystemExit("bench.py: --config c3 is one-rank (BASELINE config 3: 1 x MI355X)")
    t_upload0 = None                           # set right before the matrix goes to the device
    transport = "none (single rank)"
    comm = None
    if world > 1:
        # a rank that hangs in the communicator set-up or the first exchange (a peer that died, a transport that never
        # connects) must end the job rather than sit in a C call for ever: SIGALRM with its default action terminates this
        # rank, torch.distributed.run then stops the others.  Cancelled once the exchange check has passed.
        import signal
        signal.alarm(1200)
        # data path: RCCL inside the back-end (NativeComm); the rehearsal on one shared GPU cannot use RCCL (it refuses
        # two ranks on one device) and takes the torch.distributed callbacks staged through the host
        comm = gdist.install(hip, dist, rank, world, stage_through_host=True) if rehearse else gdist.NativeComm(hip, dist, rank, world)
        transport = "torch-callbacks (gloo, rehearsal on one GPU)" if rehearse else "rccl-native"
    elif os.environ.get("GCGE_BENCH_FORCE_COMM") == "1":
        # one-GPU rehearsal of the production multi-GPU plumbing: a communicator of ONE rank, the matrix built through
        # gcge_hip_mat_create_slab, every reduction routed through ncclAllReduce (numbers from it are one-GPU numbers)
        os.environ["GCGE_COMM_KEEP_SINGLE"] = "1"
        comm = gdist.NativeComm(hip, None, 0, 1)
        transport = "rccl-native (one-rank loop-back)"

    if c3:
        # BASELINE config 3: generalised problem A x = lambda B x, P1 stiffness / consistent mass pair on the Kuhn triangulation of a
        # cube (A 7-point x h, B 15-point x h^3), n = size^3 ~ 10^6
        A, Bc = make_problem("fe3d", N)
        n_global = N ** 3
        dims = (N, N, N)
        t_upload0 = time.perf_counter()
        mat = hip.matrix(A)
        matB = hip.matrix(Bc)
        workload = "P1 FE stiffness / mass pair on a %d^3 grid of interior nodes (A 7-point, B 15-point, n=%d), generalised problem" % (N, n_global)
    elif c5:
        # BASELINE config 5: ONE matrix whatever the rank count (strong scaling), rows cut so that every rank holds the
        # same number of non-zeros (gcge_amd.dist.partition_by_nnz)
        K, R0, R1 = args.atoms.split(",")
        kw = dict(K=int(K), R0=float(R0), R1=float(R1), seed=12345)
        n_global = N ** 3
        dims = (N, N, N)
        if comm is None:
            A, _ = make_problem("sio2", N, **kw)
            t_upload0 = time.perf_counter()
            mat = hip.matrix(A)
            part = [0, n_global]
        else:
            part0 = gdist.row_partition(n_global, world)
            A0, _ = make_problem("sio2", N, row_begin=part0[rank], row_end=part0[rank + 1], **kw)
            # cuts on plane boundaries (a plane of N^2 rows is 1 / N of the matrix): every slab keeps the plane sweep of spmm_star.hip
            # (BlockAMG: every slab of whole planes coarsens by itself, each rank pairing its own planes — gcge_mg_build_slab,
            #  csrc/host/multigrid.c.  Cuts in units of 2^k planes stay even for k levels: those levels are the whole-matrix hierarchy's
            #  rows and keep the grid form of the sweep where it shows; next to an odd cut the cells are the rank's own and the level takes
            #  blocks + tiles or pad-8 rows, profiles/r05_amg/24_...  The largest k <= levels - 1 that leaves every rank four units:
            #  171 planes on 2 / 4 / 8 ranks: 16 / 8 / 4 planes.  GCGE_BENCH_CUT_PLANES overrides: 1 = any plane boundary.)
            cut_planes = int(os.environ.get("GCGE_BENCH_CUT_PLANES", "0")) or slab_cut_planes(N, world, args.amg)
            part = gdist.partition_by_nnz(dist, A0, part0, align=cut_planes * N * N) if world > 1 else part0
            A, _ = make_problem("sio2", N, row_begin=part[rank], row_end=part[rank + 1], **kw)
            t_upload0 = time.perf_counter()
            mat = comm.slab_matrix(A, part) if isinstance(comm, gdist.NativeComm) else gdist.hip_slab_matrix(hip, comm, A, n_global, part)
        workload = ("SiO2-like matrix on a %d^3 grid (12th-order 37-point stencil + %s atom blocks, R = %s + %s u1 u2; n=%d global), "
                    "rows split by non-zeros" % (N, K, R0, R1, n_global))
    else:
        # weak scaling: every rank owns size^3 rows of a grid that stays as cube-like as the rank count allows
        # (1: N^3, 2: N x N x 2N, 4: N x 2N x 2N, 8: (2N)^3 = BASELINE config 4), cut into slabs along the last index
        dims = gdist.weak_scaling_box(N, world)
        n_global = dims[0] * dims[1] * dims[2]
        if comm is not None:
            t_upload0 = time.perf_counter()     # (the slab generator runs inside: a few tenths of a second of it are not upload)
            A, mat = gdist.lap3d_slab(hip, dims, rank, world, comm)
        else:
            A, _ = make_problem("lap3d", N)
            t_upload0 = time.perf_counter()
            mat = hip.matrix(A)
        workload = "Lap3D %d^3 rows per GPU, grid %dx%dx%d (7-pt, CSR, n=%d global)" % (N, dims[0], dims[1], dims[2], n_global)
    hip.sync()
    upload_seconds = time.perf_counter() - t_upload0      # host arrays -> device matrix in all its forms (analysis included)
    nnz_local = int(A.nnz)

    exchange = None
    if comm is not None:
        # communicator set-up (RCCL creates its point-to-point channels lazily on first use) and a check of the whole
        # exchange machinery before anything is timed: with x = 1, sum(x . A x) over all ranks is the sum of all matrix
        # entries — it needs the halo rows and the all-reduce to be right.  A wrong result ENDS the run: the transport
        # named in the JSON line is the one that ran, there is no silent switch to another one.
        vals = np.ctypeslib.as_array(A.val, shape=(max(1, nnz_local),))[:nnz_local]
        mine = float(np.sum(vals))
        if world > 1:
            allv = [None] * world
            dist.all_gather_object(allv, mine)
            want = float(sum(allv))
        else:
            want = mine
        wv, wy = hip.ops.mv_create(2, mat), hip.ops.mv_create(2, mat)
        ones = np.ones((A.nrows, 2), order="F")
        g.gcge_hip_mv_from_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_long]
        g.gcge_hip_mv_from_host(wv, 0, 2, ones.ctypes.data_as(C.POINTER(C.c_double)), A.nrows)
        hip.ops.spmm(mat, wv, wy, (0, 0), (2, 2))
        got = hip.ops.inner_prod("D", wv, wy, (0, 0), (2, 2))
        hip.ops.mv_destroy(wv, 2)
        hip.ops.mv_destroy(wy, 2)
        ok = bool(np.all(np.abs(got - want) <= 1e-9 * abs(want)))
        exchange = {"ok": ok, "got": float(got[0]), "want": want, "what": "sum over ranks of 1 . A 1 (halo rows + all-reduce) vs the sum of all matrix entries"}
        if not ok:
            raise SystemExit("bench.py rank %d: exchange check FAILED over transport '%s': %.12g != %.12g" % (rank, transport, got[0], want))
    if world > 1:
        signal.alarm(0)
    # the multigrid hierarchy behind BlockAMG (--amg L): set-up like the matrix upload (the reference's SiO2 driver builds it before
    # EigenSolverSetup_GCG, test/test_eig_sol_SiO2_MAT.c:96-128); created BEFORE the blocks below so that the fused CG sizes its
    # direction rings from what is left
    amg, amg_setup_seconds, amg_levels = None, None, None
    if args.amg >= 2:
        if world > 1 and rehearse:
            gdist.install_slab_factory(hip, comm)         # coarse slabs through the rehearsal's transport (default: RCCL from C)
        s0, s1 = (int(v) for v in args.amg_smooth.split(","))
        hip.h.GCGE_AMGCreate.restype = C.c_void_p
        hip.h.GCGE_AMGCreate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        hip.h.GCGE_AMGInstall.argtypes = [C.c_void_p, C.c_void_p]
        if args.amg_scale > 0.0:
            hip.h.gcge_mg_set_defaults.argtypes = [C.c_double, C.c_int, C.c_double]
            hip.h.gcge_mg_set_defaults(args.amg_scale, 0, -1.0)
        t_a = time.perf_counter()
        amg = C.c_void_p(hip.h.GCGE_AMGCreate(mat, None, args.amg, args.block, args.amg_cycles, s0, s1, 1e-2, hip.ops_handle))
        hip.sync()
        amg_setup_seconds = time.perf_counter() - t_a
        amg_levels = C.cast(amg, C.POINTER(C.c_int * 8)).contents[6]     # GCGE_AMG: three pointers, then num_levels
    # memory set-up, like the matrix upload: hipMalloc of the 17-34 GB work blocks costs 0.2-0.3 s each on a fresh
    # process, so the blocks one solve needs (V, eigenvectors, 3 work blocks, 3 CG blocks) are allocated once here and
    # handed back to the back-end's size-keyed pool, from which the solver's MultiVecCreateByMat calls take them
    cols = [args.nevmax + 2 * args.block, args.nevmax] + [args.block] * 6
    # + the direction ring of the fused CG (up to 15 more blocks of `block` columns, as memory allows: block_pcg.hip)
    free_b, _ = torch.cuda.mem_get_info()
    halo_rows = (2 * dims[0] * dims[1] if not c5 else 12 * dims[0] * dims[1]) * (world > 1)
    per_col = 8 * (A.nrows + halo_rows) * 1.02
    spare = free_b - per_col * sum(cols) - (14 << 30)
    if rehearse:
        spare = spare / world - (8 << 30)
    cols += [args.block] * max(0, min(15, int(spare // (per_col * max(64, args.block)))))
    warm = [hip.ops.mv_create(c, mat) for c in cols]
    for c, w in zip(cols, warm):
        hip.ops.mv_destroy(w, c)
    hip.sync()
    hip.set_random_mode(1, 20240601)          # device generator: 2e9 rand() calls would dominate at this n
    g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
    if amg is not None:
        hip.h.GCGE_AMGInstall(amg, hip.ops_handle)    # ops->MultiLinearSolver = BlockAMG (its smoother: the fused CG, per level)
    solver_args = ["-nevConv", args.nev, "-nevMax", args.nevmax, "-blockSize", args.block,
                   "-gcge_initX_orth_method", args.orth, "-gcge_compW_orth_method", args.orth]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_w0 = time.perf_counter()
    for _ in range(args.warmup):
        run_gcg(hip.ops_handle, mat, matB, solver_args, flag=1)
    hip.sync()
    warm_solve = (time.perf_counter() - t_w0) / args.warmup if args.warmup > 0 else None
    # A collective that hangs inside the timed region (a peer that died) must end THIS process with a non-zero code well
    # before the driver's limit: SIGALRM (default action: terminate) stays armed over the timed solves with a generous
    # budget — three times what the warm-up solve took per step, an hour when there was no warm-up solve to go by.
    import signal
    alarm_budget = int(3.0 * warm_solve * args.steps) + 300 if warm_solve is not None else 3600 + 600 * args.steps
    signal.alarm(alarm_budget)
    g.gcge_hip_profile_enable(1)
    g.gcge_hip_dense_profile.argtypes = [C.c_int]
    g.gcge_hip_dense_profile(1)                 # HIP events round every Gram (K2) / panel update (K3) launch: no synchronisation added
    g.gcge_hip_bpcg_time_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_double), C.c_int]
    g.gcge_hip_bpcg_time_stats(None, None, 1)
    barrier()
    t0 = time.perf_counter()
    conv_total, iters, last = 0, 0, None
    for step in range(args.steps):
        if (c5 or c3) and step == args.steps - 1:      # keep the last solve's vectors: the parity guard recomputes their residuals
            if last is not None and len(last) > 2:
                hip.ops.mv_destroy(last[2], args.nevmax)
            last = run_gcg(hip.ops_handle, mat, matB, solver_args, flag=1, keep_evec=True)
        else:
            last = run_gcg(hip.ops_handle, mat, matB, solver_args, flag=1)
        conv_total += last[1].nevConv
        iters += last[1].numIter
    barrier()
    elapsed = time.perf_counter() - t0
    signal.alarm(0)
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HIP-event intervals the back-end recorded on its own stream, by kind: 0 = MatDotMultiVec products (K1, plain or
    # with the column sums), 2 / 3 = first / second pass of a block-CG iteration (the product recomputed, never stored)
    g.gcge_hip_profile_kind.restype = C.c_long
    g.gcge_hip_profile_kind.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]

    g.gcge_hip_profile_kind_rows.restype = C.c_long
    g.gcge_hip_profile_kind_rows.argtypes = [C.c_int, C.c_int, C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def prof(kind, ncols, all_levels=False):
        # the launches on THIS matrix (A.nrows local rows): with BlockAMG the fused CG runs the same kernels on every level of the
        # hierarchy — a roofline figure belongs to one problem size, the finest level's; all_levels: every launch (time shares)
        ms_, by_ = C.c_double(), C.c_double()
        c_ = g.gcge_hip_profile_kind_rows(kind, ncols, 0 if all_levels else A.nrows, C.byref(ms_), C.byref(by_))
        return int(c_), ms_.value, by_.value

    ci, ai = C.c_long(), C.c_long()
    g.gcge_hip_bpcg_column_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
    g.gcge_hip_bpcg_column_stats(C.byref(ci), C.byref(ai))
    cg_its, cg_sec = C.c_long(), C.c_double()
    g.gcge_hip_bpcg_time_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_double), C.c_int]
    g.gcge_hip_bpcg_time_stats(C.byref(cg_its), C.byref(cg_sec), 0)
    stats = {k: prof(k, args.block) for k in (0, 2, 3)}
    spmm_ms_all = sum(prof(k, 0, all_levels=True)[1] for k in (0, 2, 3))
    g.gcge_hip_profile_enable(0)
    # in-solve rate of the dense kernels per shape (north_star: "MFMA utilisation for the TSQR/Gram kernels reported against
    # gfx950 peak"): 2 n k m flop of a launch / its HIP-event duration, summed per (kernel, k, m) over the timed steps
    g.gcge_hip_dense_profile_shapes.argtypes = [C.POINTER(C.c_double), C.c_int]
    dbuf = (C.c_double * (7 * 64))()
    nshape = min(64, g.gcge_hip_dense_profile_shapes(dbuf, 64))
    dense_rows = [tuple(dbuf[7 * i + j] for j in range(7)) for i in range(nshape)]
    if args.dense_shapes and rank == 0:
        buf = C.create_string_buffer(1 << 16)
        g.gcge_hip_dense_profile_report.argtypes = [C.c_char_p, C.c_int]
        g.gcge_hip_dense_profile_report(buf, 1 << 16)
        sys.stderr.write("in-solve rate of the dense kernels over the %d timed steps (n = %d rows per rank):\n%s" % (args.steps, A.nrows, buf.value.decode()))
    g.gcge_hip_dense_profile(0)

    # parity guard inside the bench (all ranks take part: the slots are collective)
    ev, res = last[0], last[1]
    if c5 or c3:
        # no closed form: relative residuals ||A x - lambda B x|| / (lambda ||B x||) of the converged pairs, recomputed
        # through the slots from the eigenvectors the solver returned
        k = int(res.nevConv)
        evec = last[2]
        ax = hip.ops.mv_create(k, mat)
        hip.ops.spmm(mat, evec, ax, (0, 0), (k, k))
        if matB is not None:
            # generalised problem: the reference's own criterion (src/ops_eig_sol_gcg.c:240-259) — ||A x - lambda B x||_2 against
            # lambda for B-normalised x — so the norm below is sqrt(x'Bx)
            bx = hip.ops.mv_create(k, mat)
            hip.ops.spmm(matB, evec, bx, (0, 0), (k, k))
            nx = np.sqrt(hip.ops.inner_prod("D", evec, bx, (0, 0), (k, k)))
            hip.ops.mv_destroy(evec, args.nevmax)
            evec = bx
        else:
            nx = np.sqrt(hip.ops.inner_prod("D", evec, evec, (0, 0), (k, k)))
        coef = np.zeros((k, k))
        coef[np.arange(k), np.arange(k)] = -ev[:k]
        hip.ops.lincomb(evec, ax, (0, 0), (k, k), np.asfortranarray(coef).ravel(order="F"), k, beta=np.ones(1), incb=0)
        nr = np.sqrt(hip.ops.inner_prod("D", ax, ax, (0, 0), (k, k)))
        hip.ops.mv_destroy(ax, k)
        hip.ops.mv_destroy(evec, k if matB is not None else args.nevmax)
        parity = {"max_rel_residual_recomputed": float(np.max(nr / (np.abs(ev[:k]) * nx))) if k else None}
    else:
        cs = [np.sort(2.0 * np.cos(np.arange(1, d + 1) * np.pi / (d + 1)))[::-1][:48] for d in dims]
        small = np.sort((6.0 - cs[0][:, None, None] - cs[1][None, :, None] - cs[2][None, None, :]).ravel())
        parity = {"max_rel_err_vs_closed_form": float(np.max(np.abs(ev[:res.nevConv] - small[:res.nevConv]) / small[:res.nevConv])) if res.nevConv else None}

    if rank == 0:
        npat = g.gcge_hip_mat_patterns(mat)
        chain = g.gcge_hip_mat_pattern_chain(mat)
        g.gcge_hip_mat_spmm_form.restype = C.c_char_p
        g.gcge_hip_mat_spmm_form.argtypes = [C.c_void_p]
        form = g.gcge_hip_mat_spmm_form(mat).decode()
        kbase = ("spmm_pattern", "spmm_pattern_chain", "spmm_pattern_chain2")[chain] if npat > 0 else form
        npass = (args.block + 15) // 16 if npat > 0 else 1
        g.gcge_hip_spmm_ring_launches.restype = C.c_long
        ring_launches = g.gcge_hip_spmm_ring_launches()
        g.gcge_hip_bpcg_implicit_r_iters.restype = C.c_long
        implicit_r = g.gcge_hip_bpcg_implicit_r_iters()

        def roof(kind, what, streams):
            # `achieved` prices a launch at its ALGORITHMIC bytes: kind 0: SURVEY.md 8(d), 12 B per
            # non-zero + row pointers + X read + Y written; kind 2: matrix + p read; kind 3: matrix + p, r read + r,
            # p_new written — or, without a stored residual (kernel MODE 7), matrix + p, p_prev read + p_new written.  `required` is what the pattern kernels must move at least: `streams` block streams of
            # 8 n m bytes + 2 B of pattern id per row and 16-column pass (the matrix is not read as CSR there).
            # One "launch" = the `npass` kernel launches of 16 columns that make up one m-column operation (rocprof
            # lists the 16-column launches).
            c_, ms_, by_ = stats[kind]
            if c_ == 0:
                return None
            ach = (by_ / c_) / (ms_ / c_ * 1e-3) / 1e9
            kname = "%s<7,%d,16,false>" % (kbase, kind) if npat > 0 else kbase
            if kind == 2 and ring_launches > 0:   # the read-only pass ran the LDS-ring sweep (spmm_ring.hip), 3 planes ahead
                kname = "spmm_ring<2,16,3,false>"
            if kind == 3 and implicit_r > 0:      # second pass without a stored residual (kernel MODE 7): 3 block streams
                kname = "%s<7,7,16,false>" % kbase
                streams = 3
            traffic, note = pmc_traffic(kname, N, args.block) if (npat > 0 and world == 1 and not c5) else (None, "no PMC profile for this shape")
            if c5 and world == 1 and kind == 0 and form.startswith("spmm_star"):
                traffic, note = pmc_traffic_c5(N, args.atoms, args.block)
            req = (8.0 * streams * args.block + 2.0 * npass) * A.nrows if npat > 0 else by_ / c_
            return {"bound": "hbm", "kernel": "%s x %d passes of 16 columns: %s (%d row patterns, m=%d)"
                                              % (kname, npass, what, npat, args.block) if npat > 0 else "%s: %s (m=%d)" % (kname, what, args.block),
                    "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                    "traffic_note": note, "launches": c_, "avg_launch_ms": ms_ / c_, "alg_bytes_per_launch": by_ / c_,
                    "required_bytes_per_launch": req, "frac_required": req / (ms_ / c_ * 1e-3) / 1e9 / 8000.0,
                    "share_of_step": ms_ * 1e-3 / elapsed if elapsed > 0 else None}

        r_k1 = roof(0, "Y = A X, K1 (MatDotMultiVec)", 2)
        r_p1 = roof(2, "CG pass 1, p.Ap and |Ap|^2 without storing Ap", 1)
        r_p2 = roof(3, "CG pass 2, Ap recomputed + r = p - beta_prev p_prev - alpha Ap (no stored residual), p' = r + beta p" if implicit_r > 0
                    else "CG pass 2, Ap recomputed + r -= alpha Ap, p' = r + beta p", 4)
        # dense FP64 matrix peak: the VENDOR SPEC figure for MI355X (78.6 TF; /opt/skills/guides/MI355X_MICROARCH.md has no FP64 line).
        # A register-only loop of v_mfma_f64_16x16x4_f64 measures 76.9 TF on this chip (tools/dense_bench.hip).
        FP64_MFMA_PEAK_TF = 78.6

        def dense_roof(kind, kernel, what):
            rows_ = [r for r in dense_rows if int(r[0]) == kind and r[3] > 0 and r[4] > 0]
            if not rows_:
                return None, []
            def entry(r):
                _, k_, m_, calls_, ms_, fl_, by = r
                tf = fl_ / (ms_ * 1e-3) * 1e-12
                # `by`: bytes the launches had to move, summed by the back-end per call — operands once, the panel written, and
                # read only where beta != 0 and the update is not in place
                return {"shape": {"k": int(k_), "m": int(m_), "n": int(A.nrows)}, "calls": int(calls_), "avg_launch_ms": ms_ / calls_,
                        "achieved": tf, "frac": tf / FP64_MFMA_PEAK_TF, "hbm_GBs": by / (ms_ * 1e-3) / 1e9,
                        "share_of_step": ms_ * 1e-3 / elapsed if elapsed > 0 else None}
            top = entry(rows_[0])                # rows come sorted by total time: the shape with the largest share of the step
            top.update({"bound": "mfma", "kernel": kernel, "what": what, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "share_of_step_all_shapes": sum(r[4] for r in rows_) * 1e-3 / elapsed if elapsed > 0 else None})
            return top, [entry(r) for r in rows_[1:4]]

        r_gram, gram_more = dense_roof(0, "gram_mfma (v_mfma_f64_16x16x4_f64)", "K2 Gram G = Q^T P (k x m) of MultiVecInnerProd / QtAP")
        r_upd, upd_more = dense_roof(1, "lincomb_mfma (v_mfma_f64_16x16x4_f64)", "K3 panel update Y = X C + Y diag(beta) (k x m coefficients) of MultiVecLinearComb")
        cands = [r for r in (r_k1, r_p1, r_p2) if r is not None]
        dominant = max(cands, key=lambda r: r["share_of_step"]) if cands else None
        wsolver = ("fused device block-CG (30 its, rate 1e-2)" if amg is None else
                   "BlockAMG (%d levels of 2x2x2 aggregates, %d V-cycle%s, %s fused-CG smoothing its before and after the coarse correction, rate 1e-2)"
                   % (amg_levels, args.amg_cycles, "" if args.amg_cycles == 1 else "s", args.amg_smooth.replace(",", " / ")))
        cfg = {"workload": "%s, nev=%d, block=%d, nevMax=%d, %s, tol abs 1e-1 rel 1e-8, %s, "
                           "X/W orthonormalisation '%s', device RNG start block" % (workload, args.nev, args.block, args.nevmax, "B = mass matrix" if c3 else "B=NULL", wsolver, args.orth),
               "amg_levels": amg_levels, "amg_setup_seconds": amg_setup_seconds,
               "gcg_iterations": iters, "nev_converged": conv_total,
               "cg_active_column_fraction": (ai.value / ci.value) if ci.value else None,
               "cg_iterations": cg_its.value, "ms_per_cg_iteration": 1e3 * cg_sec.value / cg_its.value if cg_its.value else None,
               "phase_seconds": {k: getattr(res.timing, k) for k in ("initX", "checkconv", "compP", "compRR", "compRV", "compW", "linsol", "total")},
               # which transport moved the halo rows and the sums, and the check it passed before the timed region
               "transport": transport, "exchange_check": exchange, "spmm_form": form, "nnz_rank0": nnz_local}
        cfg.update(parity)
        out = {
            "metric": ("converged eigenpairs/sec (GCG, SiO2-like irregular CSR n=%d, block=%d)" if c5 else
                       "converged eigenpairs/sec (GCG, FE stiffness/mass pair n=%d, block=%d, generalised)" if c3 else
                       "converged eigenpairs/sec (GCG, 3D Laplacian n=%d, block=%d)") % (n_global, args.block),
            "value": conv_total / elapsed, "unit": "eigenpairs/s", "n_gpus": world, "steps": args.steps,
            # `value` counts every pair a solve converged (a whole block locks at once: 56 for 50 wanted at config 2);
            # the pairs that were ASKED for, per second:
            "pairs_wanted_per_s": args.nev * args.steps / elapsed,
            "upload_seconds": upload_seconds, "warmup_solve_seconds": warm_solve, "alarm_budget_seconds": alarm_budget,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if c5 else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": cfg,
            # the dominant kernel of the step; the K1 product alone (the north-star figure) and the other CG pass follow
            "roofline": dominant,
            "roofline_k1_spmm": r_k1, "roofline_cg_pass1": r_p1, "roofline_cg_pass2": r_p2,
            # MFMA utilisation of the dense kernels in the solve: the shape with the largest share of the step, then the next three
            "roofline_gram": r_gram, "roofline_panel_update": r_upd,
            "dense_other_shapes": {"gram": gram_more, "panel_update": upd_more},
            "spmm_share_of_step": spmm_ms_all * 1e-3 / elapsed if elapsed > 0 else None,
        }
        import traceback

        def leg(name, fn):                    # the finished measurement must reach the driver whatever a later leg does
            t_l = time.perf_counter()
            try:
                r = fn()
            except Exception as exc:          # noqa: BLE001
                traceback.print_exc()
                r = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if isinstance(r, dict):
                r["leg_seconds"] = time.perf_counter() - t_l
            if r is not None:
                out[name] = r

        extra = world == 1 and not args.no_extra and args.config == "c2"
        if extra and amg is not None and args.plain_steps > 0:
            # the configuration rounds 1-4 reported as `value`: the same solves with the W systems through 30 iterations of the
            # fused block CG (no multigrid), so that the round-over-round record stays comparable
            def plain():
                g.gcge_hip_bpcg_setup(hip.ops_handle, 30, 1e-2, 1e-14, b"abs")
                g.gcge_hip_bpcg_time_stats(None, None, 1)
                run_gcg(hip.ops_handle, mat, None, solver_args, flag=1)       # (untimed: the ring of 15 direction blocks is created here)
                hip.sync()
                g.gcge_hip_bpcg_time_stats(None, None, 1)
                t_p = time.perf_counter()
                cv, itn = 0, 0
                for _ in range(args.plain_steps):
                    e_, r_ = run_gcg(hip.ops_handle, mat, None, solver_args, flag=1)
                    cv += r_.nevConv
                    itn += r_.numIter
                hip.sync()
                el = time.perf_counter() - t_p
                ci_, cs_ = C.c_long(), C.c_double()
                g.gcge_hip_bpcg_time_stats(C.byref(ci_), C.byref(cs_), 0)
                return {"what": "the same solves with the W systems through 30 iterations of the fused block CG instead of BlockAMG "
                                "(the configuration BENCH_r01-r04 report as value)",
                        "value": cv / el, "unit": "eigenpairs/s", "steps": args.plain_steps, "ms_per_step": 1e3 * el / args.plain_steps,
                        "gcg_iterations": itn, "nev_converged": cv, "cg_iterations": ci_.value,
                        "ms_per_cg_iteration": 1e3 * cs_.value / ci_.value if ci_.value else None}
            leg("plain_block_cg", plain)
        if extra and not args.no_cpu:
            leg("dropin_reference_stack", lambda: dropin_leg(hip, mat, args))
        if extra:
            def k1c5():
                if amg is not None:
                    hip.h.GCGE_AMGDestroy.argtypes = [C.c_void_p, C.c_void_p]
                    hip.h.GCGE_AMGDestroy(C.byref(amg), hip.ops_handle)
                g.gcge_hip_bpcg_release.argtypes = [C.c_void_p]
                g.gcge_hip_bpcg_release(hip.ops_handle)
                hip.free_matrix(mat)
                g.gcge_hip_pool_release()
                return k1_c5_leg(hip, args)
            leg("roofline_k1_c5", k1c5)
            leg("roofline_k1_permuted", lambda: k1_permuted_leg(hip, args))
        if c3 and world == 1 and not args.no_extra:
            leg("roofline_k1_spmm_A", lambda: k1_plain_leg(hip, mat, A, args.block, "stiffness matrix A (7-point stencil x h), config 3"))
            leg("roofline_k1_spmm_B", lambda: k1_plain_leg(hip, matB, Bc, args.block, "consistent mass matrix B (15-point stencil x h^3), config 3"))
        if not args.no_cpu and world == 1:
            leg("cpu_baseline", lambda: cpu_baseline(args, hip))
            if extra:
                leg("cpu_baseline_like_for_like", lambda: cpu_like_for_like(args, hip))
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1:
        if isinstance(comm, gdist.NativeComm):
            comm.finalize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
