"""(script, not collected by pytest)  Cost of the multi-GPU machinery at bench size, measured on ONE GPU: the lower
slab of a 256 x 256 x 512 Laplacian (16.8 M rows, the rank-0 share of `bench.py --gpus 2`) whose halo neighbour is the
rank itself over the real transport (backend nccl == RCCL, see rccl_loopback_worker.py).  The transfers are local
copies, so what shows is everything else an iteration pays on a row slab: pack / unpack kernels, the callbacks into
torch.distributed, their synchronisations, the split product (interior + two boundary strips) and the device round
trip of every small all-reduce.  Prints one JSON line.
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 python tests/rccl_loopback_bench.py [N]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from gcge_amd import HipBackend
    from gcge_amd import dist as gdist
    from gcge_amd.lib import CSR, host_lib, run_gcg
    h = host_lib()
    nx, ny, nz = N, N, 2 * N
    plane, n_global = nx * ny, nx * ny * nz
    part = gdist.row_partition(n_global, 2)
    n_loc = part[1]

    class LoopbackComm(gdist.Comm):
        def _peer(self, q):
            return 0

        def plan_halo(self, ghosts, part_):
            gh = np.asarray(ghosts, dtype=np.int64)
            assert gh.size == plane and gh[0] == n_loc
            return np.ascontiguousarray((gh - plane).astype(np.int32)), [0, plane], [0, plane]

        def install(self, with_allreduce=False):
            # a communicator of one rank installs no all-reduce; claiming two makes every Gram / dot result take the
            # device round trip through RCCL (sum over the one real rank = identity, the numbers stay right)
            if not with_allreduce:
                return gdist.Comm.install(self)
            c = gdist.GcgeComm(0, 2, C.cast(self._allreduce_cb, C.c_void_p), None)
            self._keep.append(c)
            h.GCGE_SetComm(C.byref(c))

    be = HipBackend(device=0)
    out = {}
    for tag in ("slab_with_loopback_halo", "slab_with_loopback_halo_and_allreduce", "same_rows_without_halo"):
        A = CSR()
        if tag.startswith("slab_with_loopback_halo"):
            comm = LoopbackComm(dist, 0, 1, device=torch.device("cuda", 0))
            comm.install(with_allreduce=tag.endswith("allreduce"))
            h.gcge_problem_lap3d_box(nx, ny, nz, C.c_int64(0), C.c_int64(n_loc), C.byref(A))
            mat = gdist.hip_slab_matrix(be, comm, A, n_global, part, cap_cols=128)
        else:
            comm.uninstall()
            h.gcge_problem_lap3d_box(nx, ny, nz // 2, C.c_int64(0), C.c_int64(-1), C.byref(A))     # Dirichlet box, no halo
            mat = be.matrix(A)
        be.set_random_mode(1, 20240601)
        be.g.gcge_hip_bpcg_setup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_char_p]
        be.g.gcge_hip_bpcg_setup(be.ops_handle, 30, 1e-2, 1e-14, b"abs")
        args = ["-nevConv", 50, "-nevMax", 128, "-blockSize", 64, "-gcge_initX_orth_method", "chol", "-gcge_compW_orth_method", "chol"]
        run_gcg(be.ops_handle, mat, None, args + ["-gcge_max_niter", 2], flag=1)             # warm the pool / communicator
        n_ar0 = comm.n_allreduce
        ev, res = run_gcg(be.ops_handle, mat, None, args, flag=1)
        out[tag] = {"rows": int(A.nrows), "seconds": res.seconds, "gcg_iterations": int(res.numIter), "nev_converged": int(res.nevConv),
                    "linsol_seconds": res.timing.linsol, "ms_per_cg_iteration": 1e3 * res.timing.linsol / (30.0 * res.numIter),
                    "seconds_per_gcg_iteration": res.seconds / res.numIter,
                    "allreduce_calls": comm.n_allreduce - n_ar0 if tag.startswith("slab") else 0}
        be.free_matrix(mat)
    print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
