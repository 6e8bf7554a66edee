"""Row-partitioned (N > 1 ranks) path: planning, halo exchange, Gram all-reduce and a whole SPMD
eigensolve with world_size = 2 over gloo.  CPU: oracle back-end.  GPU: HIP back-end, both ranks on
the one GPU of the test box (≤ 6 processes allowed), gloo transport staged through the host."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(mode, world=2, timeout=600, dims=None, spec=None, rank_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update((rank_env or {}).get(r, {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode]
                                      + ([",".join(str(d) for d in dims)] if dims else ([spec] if spec else [])),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "rank %d ok" % r in o, o[-2000:]


def test_two_ranks_gloo_cpu_oracle():
    _run("oracle")


def test_eight_ranks_gloo_cpu_oracle():
    """The 8-rank shape of bench.py's weak-scaling grid ((2N)^3 cut into 8 slabs, here N = 4: one plane per rank,
    every inner rank exchanges with two neighbours)."""
    sys.path.insert(0, ROOT)
    from gcge_amd import dist as gdist
    assert gdist.weak_scaling_box(4, 8) == (8, 8, 8) and gdist.weak_scaling_box(4, 2) == (4, 4, 8)
    assert gdist.weak_scaling_box(4, 4) == (4, 8, 8) and gdist.weak_scaling_box(4, 3) == (4, 4, 12)
    _run("oracle", world=8, dims=gdist.weak_scaling_box(4, 8))


def test_three_ranks_gloo_sio2_rows_split_by_nnz():
    """Load-imbalanced matrix (SiO2-like, rows of 19-200 non-zeros): partition_by_nnz, arbitrary halos, whole solve."""
    _run("oracle", world=3, spec="sio2:14")


def test_two_ranks_gloo_ball_cuts_between_grid_lines():
    """partition_lines on the ball matrix (a masked grid: cuts between grid lines), slabs generated per rank, whole solve on the CPU oracle."""
    _run("oracle", world=2, spec="sio2ball:14")


def test_two_ranks_gloo_sio2_cuts_on_plane_boundaries():
    """partition_by_nnz(align = plane of the grid read off a slab) over gloo, then the whole solve on the CPU oracle."""
    _run("oracle", world=2, spec="sio2star:16")


def test_row_partition_helpers():
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    from gcge_amd import dist as gdist
    from gcge_amd.lib import CSR, host_lib
    assert gdist.row_partition(10, 3) == [0, 4, 7, 10]
    h = host_lib()
    A = CSR(); h.gcge_problem_lap3d_box(4, 4, 6, C.c_int64(32), C.c_int64(64), C.byref(A))
    ghosts = gdist.localize_slab(A)
    assert list(ghosts) == list(range(16, 32)) + list(range(64, 80))      # one plane on each side
    ci = np.ctypeslib.as_array(A.colidx, shape=(int(A.nnz),))
    assert ci.min() == 0 and ci.max() == 32 + 32 - 1 and A.ncols == 64
    # cuts in units that do not divide the row count (2^(L-1) planes of a grid with an odd plane count: what a slab hierarchy of L levels
    # wants, bench.py --config c5 --gpus N): every cut a multiple of the unit, the short last unit stays with the last rank
    N = 171
    for world, L in [(2, 5), (4, 4), (8, 3)]:
        unit = (1 << (L - 1)) * N * N
        nb = (N ** 3 + unit - 1) // unit
        w = np.full(nb, float(unit)); w[-1] = N ** 3 - (nb - 1) * unit
        cuts = gdist.cuts_by_weight(w, world, unit, N ** 3)
        assert cuts[0] == 0 and cuts[-1] == N ** 3 and all(c % unit == 0 for c in cuts[:-1]) and all(a < b for a, b in zip(cuts, cuts[1:])), cuts
        share = np.diff(cuts) / (N ** 3 / world)
        assert share.max() <= 1.15, (world, L, share)
    # bench.py's unit for slabs behind BlockAMG: the largest 2^k planes (k < levels) that leaves every rank four units
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_cuts", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert [b.slab_cut_planes(171, w, 5) for w in (2, 4, 8)] == [16, 8, 4]
    assert b.slab_cut_planes(171, 8, 0) == 1 and b.slab_cut_planes(171, 2, 2) == 2 and b.slab_cut_planes(24, 2, 3) == 2 and b.slab_cut_planes(7, 8, 5) == 1


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_hip():
    _run("hip")


@pytest.mark.gpu
@pytest.mark.parametrize("dims", ["8,8,10", "32,32,40"])
def test_rccl_native_loopback_one_gpu(dims):
    """The production multi-GPU path: RCCL called from C inside libgcge_hip.so (gcge_hip_comm_init, grouped
    ncclSend/ncclRecv halo exchange on the back-end's streams, ncclAllReduce behind GCGE_COMM), one rank exchanging
    its halo with itself.  No torch.distributed in the process."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_loopback_worker.py"), dims, "native"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "rccl loop-back ok" in p.stdout, p.stdout[-3000:]
    print(p.stdout[-300:])


@pytest.mark.gpu
@pytest.mark.parametrize("G", [28])
def test_rccl_native_loopback_sweep_on_a_slab_of_a_masked_grid(G):
    """The same on the ball (a masked grid, cut between two grid lines): the slab's halo rows — lines of the planes above, found through
    the line table — arrive by grouped ncclSend/ncclRecv from C; products, products with sums, the fused CG with narrow halo buffers."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_loopback_worker.py"), str(G), "native_ball"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "rccl loop-back ok" in p.stdout, p.stdout[-3000:]
    print(p.stdout[-300:])


@pytest.mark.gpu
@pytest.mark.parametrize("G", [24, 40])
def test_rccl_native_loopback_star_sweep_on_a_slab(G):
    """The plane sweep of spmm_star.hip on a row slab over the production transport: slab 0 of a two-slab SiO2-like matrix
    exchanging its halo planes with itself by grouped ncclSend/ncclRecv from C (tests/rccl_loopback_worker.py: star_loopback)."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_loopback_worker.py"), str(G), "native_star"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "rccl loop-back ok" in p.stdout, p.stdout[-3000:]
    print(p.stdout[-300:])


@pytest.mark.gpu
def test_rccl_native_loopback_star_sweep_full_size_row_sums():
    """BASELINE config 5's matrix at full size (171^3, K = 2000), both slabs of a two-way plane-aligned cut, halo looped back over
    RCCL from C, the interior swept while it travels: every row of A 1 against the row sums of the host arrays."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_loopback_worker.py"), "171", "native_star_full", "2000"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "rccl loop-back ok" in p.stdout, p.stdout[-3000:]
    print(p.stdout[-400:])


@pytest.mark.gpu
@pytest.mark.parametrize("dims", ["8,8,10", "32,32,40"])
def test_rccl_loopback_one_gpu(dims):
    """The production transport (backend nccl == RCCL) on device buffers, one rank exchanging its halo with itself:
    see tests/rccl_loopback_worker.py.  The second size makes the split exchange overlap a real interior product."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_loopback_worker.py"), dims], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "rccl loop-back ok" in p.stdout, p.stdout[-3000:]


@pytest.mark.gpu
def test_native_worker_as_one_rank():
    """The worker of the two-GPU test below, as a world of one rank on the one GPU (communicator created, slab built by
    gcge_hip_mat_create_slab, whole solves): keeps that script running on the one-GPU box."""
    _run("hip_native", world=1)


@pytest.mark.gpu
@pytest.mark.parametrize("spec", [None, "sio2:16", "sio2star:24", "sio2ball:24"])
def test_two_ranks_two_gpus_rccl_native(spec):
    """the world > 1 branches of gcge_hip_mat_create_slab (all-gather of the per-slab counts, grouped send/recv of the
    index lists), the split halo exchange next to the all-reduces on one communicator and the device-scalar CG — two ranks on
    two devices, SpMM / Gram / whole solves (ours and the reference's stack) against the global matrix.  Needs two GPUs: skipped
    on the one-GPU test box, where this path runs as a one-rank loop-back only (test_rccl_native_loopback_one_gpu)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (%d visible)" % torch.cuda.device_count())
    _run("hip_native", spec=spec)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_hip_sio2_rows_split_by_nnz():
    _run("hip", spec="sio2:16")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_one_gpu_hip_sio2_star_sweep_on_plane_aligned_slabs(world):
    """BASELINE config 5 as specified (row slabs): partition_by_nnz(align = plane) cuts on plane boundaries — inside atom
    blocks — and every slab keeps the plane sweep of spmm_star.hip, its first / last 6 planes reading their z-neighbours from
    the halo rows; products with and without the interior / boundary split, odd column ranges, the product with its column
    sums, a whole SPMD solve, all against the global matrix."""
    _run("hip", world=world, spec="sio2star:24")


@pytest.mark.gpu
@pytest.mark.parametrize("world,G", [(2, 24), (3, 28)])
def test_ranks_on_one_gpu_hip_slabs_of_a_masked_grid_keep_the_sweep(world, G):
    """Row slabs of a MASKED grid (the PARSEC matrices behind BASELINE config 5 on more than one device): the
    SiO2-like operator on the ball, partition cut between grid lines (gdist.partition_lines), geometry named to the slab constructor —
    every slab keeps the plane sweep (third form: own AND halo rows found through one line table), its products (plain, odd column
    ranges, with the column sums) and a whole SPMD solve against the global matrix."""
    _run("hip", world=world, spec="sio2ball:%d" % G)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_masked_grid_from_a_file_geometry_recovered():
    """The flow for the reference's real SiO2 file on two devices: a Matrix-Market file (no grid named), every rank recovers the
    geometry from the file's rows (gcge_hip_star_infer_grid), the rows are cut between the recovered grid lines, each rank uploads its
    slab with that geometry — the slabs keep the plane sweep (third form), products and the SPMD solve match the global matrix."""
    _run("hip", world=2, spec="ballfile:24")


@pytest.mark.gpu
def test_native_worker_one_rank_slab_of_a_masked_grid():
    """gcge_hip_mat_create_slab_grid (RCCL constructor with the geometry named) as a world of one rank."""
    _run("hip_native", world=1, spec="sio2ball:24")


@pytest.mark.gpu
@pytest.mark.parametrize("world,dims", [(2, (8, 8, 16)), (3, (8, 8, 24))])
def test_ranks_on_one_gpu_block_amg_on_slabs(world, dims):
    """BlockAMG with a row-partitioned hierarchy (round 5): ops->MultiGridCreate coarsens every rank's slab by itself
    (gcge_mg_build_slab: whole planes, even cuts), coarse slabs through the transport's slab constructor, the fused CG as the
    smoother of every level with its sums reduced over the ranks — GCG with -gcge_amg_levels against the closed form and the plain
    solver, two and three ranks sharing the one GPU over gloo."""
    _run("hip", world=world, dims=dims, rank_env={r: {"GCGE_TEST_AMG": "3"} for r in range(world)})


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_block_amg_on_slabs_of_the_sio2_like_matrix():
    """The same on the SiO2-like matrix (12th-order stencil + atom blocks that cross the cut): slabs in grid form on every level that
    still shows the star, coarse slabs through the slab constructor, smoothing through the stored-product CG with halo exchange."""
    _run("hip", world=2, spec="sio2star:24", rank_env={r: {"GCGE_TEST_AMG": "3"} for r in range(2)})


@pytest.mark.gpu
@pytest.mark.parametrize("size,atoms,levels,unit", [(48, "60,2.0,5.0", 4, None), (96, "350,2.0,5.0", 5, None), (48, "60,2.0,5.0", 5, 1)])
def test_two_ranks_on_one_gpu_every_level_of_the_slab_hierarchy_multiplies_right(size, atoms, levels, unit):
    """Every level of the slab hierarchy of the SiO2-like matrix (tests/slab_level_worker.py): uploaded through the slab constructor,
    product and product-with-column-sums (the fused CG's entry point, gcge_hip_spmm_dot2_mv) against the slab's rows on the host, with
    and without the interior rows swept while the halo travels.  The coarse levels leave the grid form (pad-8 rows, then repeated
    patterns): the 96^3 case is where the fused product + x.y sums over a ROW STRIP read the strip's own rows of x from row 0 of the
    block (round 5: column sums off by 6e-5 on levels 2 and 3 with the split exchange — BlockAMG on slabs deeper than two levels
    did not converge).  unit = 1: cuts on any plane boundary, an odd one included — every rank pairs its own planes (csrc/host/multigrid.c)."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "slab_level_worker.py"), str(size), atoms, str(levels)] + ([str(unit)] if unit else []),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "rank %d: worst relative error" % r in o and "products OK on every level" in o, o[-2000:]
    assert "PASS levels=%d" % levels in outs[0], outs[0][-1500:]


@pytest.mark.gpu
def test_native_worker_as_one_rank_block_amg():
    """The same through the production constructor (gcge_hip_mat_create_slab, RCCL from C) as a world of one rank."""
    _run("hip_native", world=1, dims=(8, 8, 16), rank_env={0: {"GCGE_TEST_AMG": "3"}})


@pytest.mark.gpu
def test_two_ranks_disagree_on_the_cg_ring_length():
    """The fused CG's direction ring is sized from each rank's own free memory (or GCGE_CG_RING); the length decides
    the column window and with it the length of the per-iteration all-reduces, so the ranks must settle on one
    value (block_pcg.hip: vote over GCGE_COMM).  Rank 0 may take 15 extra slots, rank 1 none: both must run without a
    ring, with matching all-reduce counts, to the same Ritz values."""
    _run("hip", rank_env={0: {"GCGE_CG_RING": "16"}, 1: {"GCGE_CG_RING": "3"}})


@pytest.mark.gpu
def test_c_host_drives_the_backend_with_rccl(tmp_path):
    """tools/test_app_hip_multi.c: a plain-C program (no Python, no torch, no MPI) that initialises RCCL through the
    C ABI, builds its slab with gcge_hip_mat_create_slab and solves — here as a world of one rank on the one GPU."""
    exe = str(tmp_path / "test_app_hip_multi")
    lib = os.path.join(ROOT, "gcge_amd", "lib")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "test_app_hip_multi.c"),
                    "-o", exe, "-L" + lib, "-lgcge_hip", "-lgcge_host", "-Wl,-rpath," + lib, "-lm"], check=True)
    env = dict(os.environ, GCGE_RANK="0", GCGE_WORLD="1", GCGE_LOCAL_DEVICE="0", GCGE_ID_FILE=str(tmp_path / "id"),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "16", "10"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and " OK" in p.stdout, p.stdout[-2000:]
