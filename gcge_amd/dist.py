"""Row-partitioned execution: one process per GPU, every block of vectors and the matrix split by
contiguous row slabs (the reference's MPI back-ends do the same: app/app_slepc.c:49-60,
src/ops_multi_vec.c:206-228).  Two exchanges exist on the hot path:

  * all-reduce (sum, f64) of every small Gram / dot result — GCGE_COMM.allreduce_sum, installed
    here over torch.distributed (backend "nccl" == RCCL over xGMI on the GPUs, "gloo" on CPU);
  * the SpMM halo: rows of X owned by other ranks — point-to-point batch_isend_irecv of packed
    row blocks (RCCL send/recv), planned once per matrix by plan_halo().

Python is plumbing: planning happens once, the callbacks only move buffers.

Production path on the GPUs (NativeComm below): both exchanges live in libgcge_hip.so (csrc/hip/rccl_comm.hip —
ncclAllReduce and grouped ncclSend/ncclRecv on the back-end's own streams, planned in C); Python only hands rank 0's
RCCL id to the other ranks.  The callback classes further down remain for what RCCL cannot do: the CPU oracle over
gloo and several ranks sharing one GPU (the rehearsals of tests/test_dist.py).
"""
import ctypes as C

import numpy as np

from .lib import CSR, host_lib


class GcgeComm(C.Structure):
    """GCGE_COMM (include/gcge_ops.h)."""
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("allreduce_sum", C.c_void_p), ("ctx", C.c_void_p)]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_void_p)
EXCHANGE_END_FN = C.CFUNCTYPE(None, C.c_void_p)


def row_partition(n_global, world):
    """Contiguous, as even as possible: part[r] .. part[r+1] are rank r's rows."""
    base, rem = divmod(n_global, world)
    part = [0]
    for r in range(world):
        part.append(part[-1] + base + (1 if r < rem else 0))
    return part


def grid_of(A):
    """(nx, ny, nz, arm) of the lexicographic grid behind a matrix whose rows are mostly one star stencil (the real-space
    DFT Hamiltonians of BASELINE config 5), read off a CSR slab with GLOBAL columns; None: no such grid.  Host only."""
    from .lib import hip_lib
    g = hip_lib()
    g.gcge_hip_star_grid.argtypes = [C.c_int, C.c_long, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_double), C.POINTER(C.c_long)]
    out = (C.c_long * 4)()
    if not g.gcge_hip_star_grid(A.nrows, A.row_begin, A.ncols, A.rowptr, A.colidx, A.val, out):
        return None
    return tuple(int(v) for v in out)


def partition_by_nnz(dist, A, part, bucket=None, align=None):
    """Contiguous row partition with (nearly) equal numbers of non-zeros per rank — what SURVEY §8e asks for on the
    load-imbalanced matrices (rows of very different length).  A: this rank's CSR slab under the current partition
    `part` (any contiguous one, e.g. row_partition).  Every rank sums its row lengths over buckets of `bucket` global
    rows, the sums are gathered, and the cuts are put at the bucket boundaries nearest to k/world of the total.
    align: cuts only at multiples of `align` rows — the plane size nx * ny of a grid matrix (grid_of), so that every slab is
    whole planes and keeps the plane sweep of spmm_star.hip; a plane of 171^2 rows is 0.6 % of BASELINE config 5's matrix.  A multiple
    of the plane size (2^(L-1) planes: cuts that stay on even planes down L levels of a slab hierarchy, which then equals the whole
    matrix's, csrc/host/multigrid.c) need not divide the row count: the last unit is then a short one (171 planes = 10 units of 16
    planes + 11 planes)."""
    world = len(part) - 1
    n_global = part[-1]
    rank = dist.get_rank()
    if align is not None:
        if n_global // align < world:
            align = None                                  # fewer planes than ranks: free cuts
        else:
            bucket = align
    if bucket is None:
        bucket = max(1, n_global // (1024 * world))      # cuts resolved to ~0.1 % of a rank's share
    rp = np.ctypeslib.as_array(A.rowptr, shape=(A.nrows + 1,)).astype(np.int64)
    rows = np.arange(part[rank], part[rank + 1])
    nb = (n_global + bucket - 1) // bucket
    mine = np.bincount(rows // bucket, weights=np.diff(rp).astype(np.float64), minlength=nb)
    allb = [None] * world
    dist.all_gather_object(allb, mine)
    return cuts_by_weight(np.sum(allb, axis=0), world, bucket, n_global)


def cuts_by_weight(weight, world, bucket, n_global):
    """weight[b]: non-zeros of global rows [b * bucket, (b + 1) * bucket).  Cuts at the bucket boundaries nearest to k / world of
    the total; every rank keeps at least one bucket (so at least one row)."""
    cum = np.concatenate([[0.0], np.cumsum(weight)])
    nb = len(weight)
    new = [0]
    for k in range(1, world):
        cut = int(np.searchsorted(cum, cum[-1] * k / world))
        if cut > 0 and abs(cum[cut - 1] - cum[-1] * k / world) < abs(cum[cut] - cum[-1] * k / world):
            cut -= 1
        cut = min(cut, nb - (world - k))                   # leave a bucket for every rank that follows
        new.append(min(n_global, max(new[-1] + bucket, cut * bucket)))
    new.append(n_global)
    return new


def _loud(fn):
    """ctypes swallows exceptions raised inside callbacks (the C caller would carry on with stale halo rows or
    un-reduced sums): print the traceback and take the process down instead."""
    def wrapped(*a):
        try:
            return fn(*a)
        except BaseException:      # noqa: BLE001 - nothing may escape into the C caller
            import os
            import sys
            import traceback
            traceback.print_exc()
            sys.stderr.write("gcge_amd.dist: exception inside a communication callback - aborting\n")
            sys.stderr.flush()
            os._exit(70)
    return wrapped


class Comm:
    """Owns the torch.distributed plumbing of one rank and the ctypes callbacks built on it."""

    def __init__(self, dist, rank, world, device=None, stage_through_host=False):
        import torch
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.device = device                  # torch.device for device-resident exchange buffers, None = host
        self.stage = stage_through_host       # device buffers but a host-only transport (gloo): copy through host
        self._allreduce_cb = ALLREDUCE_FN(_loud(self._allreduce))
        self._keep = []
        self.n_allreduce = 0
        self._ar_cap, self._ar_pin, self._ar_dev = 0, None, None

    # ---- small-result all-reduce (host buffer in, host buffer out) -------------------------------
    def _allreduce(self, buf, n, ctx):
        torch = self.torch
        arr = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(arr)
        if self.device is not None and not self.stage:
            # persistent pinned + device staging: two asynchronous copies and one stream synchronisation per call
            if self._ar_cap < n:
                self._ar_cap = max(1024, 2 * n)
                self._ar_pin = torch.empty(self._ar_cap, dtype=torch.float64).pin_memory()
                self._ar_dev = torch.empty(self._ar_cap, dtype=torch.float64, device=self.device)
            pin, dev = self._ar_pin[:n], self._ar_dev[:n]
            pin.copy_(t)
            dev.copy_(pin, non_blocking=True)
            self.dist.all_reduce(dev)
            pin.copy_(dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            t.copy_(pin)
        else:
            self.dist.all_reduce(t)
        self.n_allreduce += 1

    def _peer(self, q):
        """torch.distributed rank that owns partition q (identity; the single-GPU RCCL loop-back test maps every
        partition to rank 0 so that the real transport runs against itself)."""
        return q

    def install(self):
        """GCGE_SetComm: from now on inner products / CG scalars are summed over the ranks."""
        h = host_lib()
        c = GcgeComm(self.rank, self.world, C.cast(self._allreduce_cb, C.c_void_p), None)
        self._keep.append(c)
        h.GCGE_SetComm(C.byref(c))

    def uninstall(self):
        host_lib().GCGE_SetComm(None)

    # ---- halo planning ------------------------------------------------------------------------
    def plan_halo(self, ghosts, part):
        """ghosts: ascending global row ids this rank needs.  Returns (send_rows, send_cnt, recv_cnt): local rows to ship grouped by
        destination rank, and per-peer counts — from gcge_dist_plan_halo (csrc/host/problems.c), the planner
        gcge_hip_mat_create_slab runs over RCCL; here its two int transports are torch.distributed collectives."""
        h = host_lib()
        world, rank, dist = self.world, self.rank, self.dist

        def allgather_int(send, n, recv_all, ctx):
            mine = np.ctypeslib.as_array(send, shape=(max(1, n),))[:n].copy()
            got = [None] * world
            dist.all_gather_object(got, mine)
            out = np.ctypeslib.as_array(recv_all, shape=(max(1, n * world),))
            for r in range(world):
                out[r * n:(r + 1) * n] = got[r]

        def exchange_int(sendbuf, send_cnt, recvbuf, recv_cnt, ctx):
            sc = np.ctypeslib.as_array(send_cnt, shape=(world,)).copy()
            rc = np.ctypeslib.as_array(recv_cnt, shape=(world,)).copy()
            ns, nr = int(sc.sum()), int(rc.sum())
            sb = np.ctypeslib.as_array(sendbuf, shape=(max(1, ns),))[:ns].copy()
            off = np.concatenate([[0], np.cumsum(sc)])
            groups = [sb[off[q]:off[q + 1]] for q in range(world)]
            got = [None] * world
            dist.all_gather_object(got, groups)             # (set-up only: every rank sees every list and keeps its own)
            out = np.ctypeslib.as_array(recvbuf, shape=(max(1, nr),))
            o = 0
            for q in range(world):
                seg = got[q][rank]
                assert len(seg) == rc[q], ("halo plan: rank %d sends %d ints, %d expected" % (q, len(seg), rc[q]))
                out[o:o + rc[q]] = seg
                o += rc[q]

        AG = C.CFUNCTYPE(None, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_void_p)
        EX = C.CFUNCTYPE(None, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p)

        class Transport(C.Structure):
            _fields_ = [("rank", C.c_int), ("size", C.c_int), ("allgather_int", AG), ("exchange_int", EX), ("ctx", C.c_void_p)]
        tr = Transport(rank, world, AG(_loud(allgather_int)), EX(_loud(exchange_int)), None)
        gh = np.ascontiguousarray(ghosts, dtype=np.int32)
        parr = (C.c_long * (world + 1))(*[int(v) for v in part])
        recv_cnt = (C.c_int * world)()
        send_cnt = (C.c_int * world)()
        rows = C.POINTER(C.c_int)()
        ns = C.c_int()
        h.gcge_dist_plan_halo.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_int), C.c_int, C.POINTER(Transport), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_int)]
        rc = h.gcge_dist_plan_halo(parr, gh.ctypes.data_as(C.POINTER(C.c_int)), int(gh.size), C.byref(tr), recv_cnt, send_cnt,
                                   C.byref(rows), C.byref(ns))
        if rc != 0:
            raise RuntimeError("gcge_dist_plan_halo failed: %d" % rc)
        send_rows = np.ctypeslib.as_array(rows, shape=(max(1, ns.value),))[:ns.value].astype(np.int32).copy()
        h.gcge_free_ints(rows)
        return np.ascontiguousarray(send_rows, dtype=np.int32), [int(v) for v in send_cnt], [int(v) for v in recv_cnt]

    def make_exchange(self, send_cnt, recv_cnt, cap_cols):
        """Allocates the exchange buffers and returns (callback, send_ptr, recv_ptr, keepalive)."""
        torch = self.torch
        nsend, nrecv = sum(send_cnt), sum(recv_cnt)
        dev = self.device if self.device is not None else torch.device("cpu")
        send_t = torch.zeros(max(1, nsend * cap_cols), dtype=torch.float64, device=dev)
        recv_t = torch.zeros(max(1, nrecv * cap_cols), dtype=torch.float64, device=dev)
        soff = np.concatenate([[0], np.cumsum(send_cnt)]).astype(int)
        roff = np.concatenate([[0], np.cumsum(recv_cnt)]).astype(int)
        host_buffers = self.device is None

        def exchange(sendbuf, recvbuf, ncols, ctx):
            dist = self.dist
            if host_buffers:     # the caller owns plain host arrays of exactly nsend/nrecv x ncols
                s_all = torch.from_numpy(np.ctypeslib.as_array(sendbuf, shape=(max(1, nsend * ncols),)))
                r_all = torch.from_numpy(np.ctypeslib.as_array(recvbuf, shape=(max(1, nrecv * ncols),)))
            elif self.stage:
                torch.cuda.synchronize()
                s_all = send_t[:max(1, nsend * ncols)].cpu()
                r_all = torch.zeros(max(1, nrecv * ncols), dtype=torch.float64)
            else:
                # The pack/unpack kernels run on the back-end's own HIP stream, RCCL on torch's: order them through
                # the host (device-wide sync before the sends are posted and after the receives have completed).
                torch.cuda.synchronize()
                s_all, r_all = send_t, recv_t
            ops = []
            for q in range(len(send_cnt)):          # the own rank has zero counts
                if send_cnt[q]:
                    ops.append(dist.P2POp(dist.isend, s_all[soff[q] * ncols:soff[q + 1] * ncols], self._peer(q)))
                if recv_cnt[q]:
                    ops.append(dist.P2POp(dist.irecv, r_all[roff[q] * ncols:roff[q + 1] * ncols], self._peer(q)))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if self.stage and not host_buffers:
                recv_t[:max(1, nrecv * ncols)].copy_(r_all)
                torch.cuda.synchronize()
            elif not host_buffers:
                torch.cuda.synchronize()

        # split form for device buffers (gcge_hip_mat_set_halo_async): begin posts, end completes; the back-end multiplies
        # the interior rows in between.  Over a host-staged transport (gloo rehearsal) begin does everything.
        pending = []

        def begin(sendbuf, recvbuf, ncols, ctx):
            if host_buffers or self.stage:
                exchange(sendbuf, recvbuf, ncols, ctx)
                return
            torch.cuda.synchronize()                      # the pack kernel ran on the back-end's stream
            ops = []
            for q in range(len(send_cnt)):
                if send_cnt[q]:
                    ops.append(self.dist.P2POp(self.dist.isend, send_t[soff[q] * ncols:soff[q + 1] * ncols], self._peer(q)))
                if recv_cnt[q]:
                    ops.append(self.dist.P2POp(self.dist.irecv, recv_t[roff[q] * ncols:roff[q + 1] * ncols], self._peer(q)))
            pending[:] = self.dist.batch_isend_irecv(ops) if ops else []

        def end(ctx):
            if not pending:
                return
            for w in pending:
                w.wait()                                   # torch's current stream now waits for the transfers ...
            torch.cuda.current_stream().synchronize()      # ... and the host for that stream (not for the interior product)
            pending[:] = []

        cb = EXCHANGE_FN(_loud(exchange))
        self._begin_cb, self._end_cb = EXCHANGE_FN(_loud(begin)), EXCHANGE_END_FN(_loud(end))
        self._keep += [cb, self._begin_cb, self._end_cb, send_t, recv_t]
        sp = C.cast(send_t.data_ptr(), C.POINTER(C.c_double))
        rp = C.cast(recv_t.data_ptr(), C.POINTER(C.c_double))
        return cb, sp, rp


class NativeComm:
    """RCCL inside the back-end (include/gcge_hip.h "multi-GPU from C").  `dist` (any initialised torch.distributed
    backend) is used once, to broadcast the 128-byte communicator id."""

    def __init__(self, hip, dist, rank, world):
        self.hip, self.rank, self.world = hip, rank, world
        g = hip.g
        g.gcge_hip_comm_unique_id.argtypes = [C.c_void_p]
        g.gcge_hip_comm_init.argtypes = [C.c_int, C.c_int, C.c_void_p]
        g.gcge_hip_mat_create_slab.restype = C.c_void_p
        g.gcge_hip_mat_create_slab.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_double), C.c_int]
        g.gcge_hip_comm_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        ident = C.create_string_buffer(128)
        if rank == 0 and g.gcge_hip_comm_unique_id(ident) != 0:
            raise RuntimeError("gcge_hip_comm_unique_id failed")
        box = [ident.raw]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        rc = g.gcge_hip_comm_init(rank, world, box[0])
        if rc != 0:
            raise RuntimeError("gcge_hip_comm_init failed: %d" % rc)

    @property
    def n_allreduce(self):
        a, e = C.c_long(), C.c_long()
        self.hip.g.gcge_hip_comm_stats(C.byref(a), C.byref(e))
        return a.value

    def slab_matrix(self, A, part, cap_cols=128, geometry=None):
        """A: CSR slab with GLOBAL column indices (rows part[rank] .. part[rank+1]).  Collective.  geometry = (dims, box_global): a
        matrix on a masked grid, cuts between grid lines (partition_lines): gcge_hip_mat_create_slab_grid."""
        parr = (C.c_long * (self.world + 1))(*part)
        if geometry is not None:
            dims, box_global = geometry
            bg = np.ascontiguousarray(box_global, dtype=np.int32)
            f = self.hip.g.gcge_hip_mat_create_slab_grid
            f.restype = C.c_void_p
            f.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int,
                          C.POINTER(C.c_int)]
            m = f(parr, A.rowptr, A.colidx, A.val, cap_cols, int(dims[0]), int(dims[1]), int(dims[2]), bg.ctypes.data_as(C.POINTER(C.c_int)))
            if not m:
                raise RuntimeError("gcge_hip_mat_create_slab_grid failed")
            return C.c_void_p(m)
        m = self.hip.g.gcge_hip_mat_create_slab(parr, A.rowptr, A.colidx, A.val, cap_cols)
        if not m:
            raise RuntimeError("gcge_hip_mat_create_slab failed")
        return C.c_void_p(m)

    def finalize(self):
        self.hip.g.gcge_hip_comm_finalize()


def localize_slab(A):
    """A: CSR slab with GLOBAL columns.  Returns the ascending ghost list; A is rewritten to local numbering."""
    h = host_lib()
    gp = C.POINTER(C.c_int)()
    ng = C.c_int()
    rc = h.gcge_dist_ghosts(C.byref(A), C.byref(gp), C.byref(ng))
    if rc != 0:
        raise RuntimeError("gcge_dist_ghosts failed")
    ghosts = np.ctypeslib.as_array(gp, shape=(ng.value,)).copy() if ng.value else np.zeros(0, dtype=np.int32)
    rc = h.gcge_dist_localize(C.byref(A), gp, ng)
    h.gcge_free_ints(gp)
    if rc != 0:
        raise RuntimeError("gcge_dist_localize failed")
    return ghosts


def install(hip, dist, rank, world, stage_through_host=False):
    """Comm for the HIP back-end of this rank (device-resident exchange buffers)."""
    import torch
    comm = Comm(dist, rank, world, device=torch.device("cuda", torch.cuda.current_device()),
                stage_through_host=stage_through_host)
    comm.install()
    return comm


def partition_lines(box_global, nx, world):
    """Row partition of a matrix on a MASKED grid (box_global[r] = x + nx (y + ny z), rows in scan order) with every cut between two
    grid LINES, as even in rows as that allows: what gcge_hip_mat_create_slab_grid / hip_slab_matrix(geometry=) need to keep the
    plane sweep on every slab (a line is one run of rows of ONE rank)."""
    box = np.asarray(box_global, dtype=np.int64)
    n = int(box.size)
    starts = np.concatenate([[0], np.nonzero(box[1:] // nx != box[:-1] // nx)[0] + 1, [n]])     # first row of every line, and n
    part = [0]
    for q in range(1, world):
        want = q * n // world
        k = int(np.searchsorted(starts, want))
        cand = [int(starts[j]) for j in (k - 1, k) if 0 <= j < len(starts)]
        cut = min(cand, key=lambda c: abs(c - want))
        part.append(max(cut, part[-1]))
    part.append(n)
    return part


def hip_slab_matrix(hip, comm, A, n_global, part, cap_cols=128, geometry=None):
    """Upload a CSR slab (GLOBAL columns on entry) and install its halo plan.  geometry = (dims, box_global): the matrix lives on a
    masked grid (box index of every GLOBAL row); with cuts between grid lines (partition_lines) the slab keeps the plane sweep."""
    g = hip.g
    ghosts = localize_slab(A)
    send_rows, send_cnt, recv_cnt = comm.plan_halo(ghosts, part)
    if geometry is not None:
        dims, box_global = geometry
        bg = np.asarray(box_global, dtype=np.int32)
        box_local = np.ascontiguousarray(np.concatenate([bg[A.row_begin:A.row_begin + A.nrows], bg[np.asarray(ghosts, dtype=np.int64)]]), dtype=np.int32)
        g.gcge_hip_star_next_geometry_cols.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        g.gcge_hip_star_next_geometry_cols(int(box_local.size), int(dims[0]), int(dims[1]), int(dims[2]), box_local.ctypes.data_as(C.POINTER(C.c_int)))
    # (the halo rows' global ids go with the arrays: a slab of a grid matrix cut on plane boundaries keeps the plane sweep)
    g.gcge_hip_mat_create_local_ghosts.restype = C.c_void_p
    g.gcge_hip_mat_create_local_ghosts.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                   C.POINTER(C.c_double), C.POINTER(C.c_int)]
    gh = np.ascontiguousarray(ghosts, dtype=np.int32)
    m = g.gcge_hip_mat_create_local_ghosts(A.nrows, A.ncols, n_global, A.row_begin, A.rowptr, A.colidx, A.val,
                                           gh.ctypes.data_as(C.POINTER(C.c_int)))
    if geometry is not None:
        g.gcge_hip_star_next_geometry_cols(0, 0, 0, 0, None)          # (not consumed when the slab took another form)
    if not m:
        raise RuntimeError("gcge_hip_mat_create_local_ghosts failed")
    mat = C.c_void_p(m)
    cb, sp, rp = comm.make_exchange(send_cnt, recv_cnt, cap_cols)
    g.gcge_hip_mat_set_halo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.c_int, C.c_void_p, C.c_void_p]
    g.gcge_hip_mat_set_halo(mat, n_global, int(send_rows.size), send_rows.ctypes.data_as(C.POINTER(C.c_int)), sp, rp,
                            cap_cols, C.cast(cb, C.c_void_p), None)
    g.gcge_hip_mat_set_halo_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    g.gcge_hip_mat_set_halo_async(mat, C.cast(comm._begin_cb, C.c_void_p), C.cast(comm._end_cb, C.c_void_p))
    # the partition of all ranks travels with the handle: ops->MultiGridCreate coarsens a slab from it (csrc/hip/multigrid.hip)
    g.gcge_hip_mat_set_partition.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.c_int]
    g.gcge_hip_mat_set_partition(mat, (C.c_long * len(part))(*[int(v) for v in part]), len(part) - 1)
    return mat


def install_slab_factory(hip, comm):
    """Coarse slabs of a multigrid hierarchy (ops->MultiGridCreate on a row slab) through THIS transport: registers hip_slab_matrix as
    the back-end's slab constructor (gcge_hip_set_slab_factory; the default is gcge_hip_mat_create_slab over RCCL).  Returns the
    callback object (keep it alive)."""
    FACT = C.CFUNCTYPE(C.c_void_p, C.POINTER(C.c_long), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                       C.c_int, C.c_void_p)

    def factory(part_p, world, rank, rowptr, colidx, val, buf_cols, ctx):
        part = [int(part_p[i]) for i in range(world + 1)]
        nrows = part[rank + 1] - part[rank]
        A = CSR(nrows, part[world], part[rank], int(rowptr[nrows]), rowptr, colidx, val)
        m = hip_slab_matrix(hip, comm, A, part[world], part, cap_cols=buf_cols)      # (rewrites colidx to local numbering in place)
        return m.value
    cb = FACT(_loud(factory))
    hip.g.gcge_hip_set_slab_factory.argtypes = [C.c_void_p, C.c_void_p]
    hip.g.gcge_hip_set_slab_factory(C.cast(cb, C.c_void_p), None)
    comm._keep.append(cb)
    return cb


def weak_scaling_box(N, world):
    """Grid of bench.py's weak-scaling workload: N^3 rows per rank, as cube-like as possible so that the spectrum
    (and with it the GCG iteration count) stays comparable across rank counts: 1: N^3, 2: N x N x 2N,
    4: N x 2N x 2N, 8: (2N)^3 (= BASELINE config 4 at N = 256); other counts: N x N x (N world).
    Slabs are cut along the last (slowest) index, so the short edges come first: the halo plane (product of the
    first two) is the smallest face of the box."""
    dims = {1: (N, N, N), 2: (N, N, 2 * N), 4: (N, 2 * N, 2 * N), 8: (2 * N, 2 * N, 2 * N)}
    return dims.get(world, (N, N, N * world))


def lap3d_slab(hip, dims, rank, world, comm=None):
    """Slab of the 7-point Laplacian on an Nx x Ny x Nz grid: rank r owns Nz/world planes (rows in natural order).
    comm: NativeComm (RCCL in the back-end) or Comm (torch.distributed callbacks)."""
    h = host_lib()
    nx, ny, nz = dims
    n_global = nx * ny * nz
    part = row_partition(n_global, world)
    A = CSR()
    rc = h.gcge_problem_lap3d_box(C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int64(part[rank]),
                                  C.c_int64(part[rank + 1]), C.byref(A))
    if rc != 0:
        raise RuntimeError("gcge_problem_lap3d_box failed")
    if isinstance(comm, NativeComm):
        mat = comm.slab_matrix(A, part)
    else:
        mat = hip_slab_matrix(hip, comm, A, n_global, part)
    return A, mat
