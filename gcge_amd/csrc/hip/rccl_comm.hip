// Multi-GPU inside the back-end: RCCL over xGMI, one process per GPU, called from C.
//
// The reference's collectives are C calls in the solver layers — MPI_Allreduce of every Gram / dot result
// (src/ops_multi_vec.c:206-228, src/ops_lin_sol.c:313-321,361-369) — and its distributed back-ends exchange the
// off-process part of X inside MatDotMultiVec (overlap precedent: app/app_phg.c:292-359).  Here both live in
// libgcge_hip.so so that a plain C host (the reference harness, INTEGRATION.md's test_app_hip.c) can use all GPUs
// of a node:
//   * gcge_hip_comm_init      ncclCommInitRank on the current device + GCGE_COMM (include/gcge_ops.h) whose
//                             allreduce_sum is one ncclAllReduce on the back-end's stream;
//   * gcge_hip_mat_create_slab / gcge_hip_mat_set_halo_rccl   halo plan of a row slab: the rows of X other ranks
//                             need are packed by a kernel, moved by ONE grouped ncclSend/ncclRecv per product and
//                             unpacked into the halo rows — ordered by events, no host or device-wide synchronisation;
//                             in the split form the transfer runs on its own stream while the interior rows are
//                             multiplied (app_hip.hip: spmm_halo).
// xGMI is point-to-point, a slab has two neighbours: the grouped send/recv uses exactly the two links involved.
// Python (bench.py, gcge_amd/dist.py) keeps only the bootstrap: handing rank 0's 128-byte id to the other ranks.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "gcge_hip.h"
#include "gcge_hip_internal.h"

#define GCGE_NCCL_CHECK(expr)                                                                  \
  do {                                                                                         \
    ncclResult_t r_ = (expr);                                                                  \
    if (r_ != ncclSuccess) {                                                                   \
      fprintf(stderr, "gcge_hip: %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__,        \
              ncclGetErrorString(r_));                                                         \
      abort();                                                                                 \
    }                                                                                          \
  } while (0)

static ncclComm_t g_nccl = nullptr;
static int g_rank = 0, g_world = 1;
static hipStream_t g_cs = nullptr;              // transfer stream of the split halo exchange (non-blocking)
static hipEvent_t g_ev_packed = nullptr, g_ev_arrived = nullptr;
static double *g_ar_dev = nullptr, *g_ar_pin = nullptr; static size_t g_ar_cap = 0;
static GCGE_COMM g_comm_desc;
static long g_n_allreduce = 0, g_n_exchange = 0;

static_assert(sizeof(ncclUniqueId) == GCGE_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int gcge_hip_comm_unique_id(void* id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
  memcpy(id128, &id, sizeof(id));
  return 0;
}

// in-place sum over the ranks of n doubles in HOST memory (GCGE_COMM contract): pinned + device staging, one
// ncclAllReduce on the back-end's stream, one stream synchronisation
static void rccl_allreduce_host(double* buf, int n, void* ctx) {
  (void)ctx;
  if (n <= 0) return;
  hipStream_t st = (hipStream_t)gcge_hip_stream();
  if ((size_t)n > g_ar_cap) {
    GCGE_HIP_CHECK(hipStreamSynchronize(st));
    if (g_ar_dev) { GCGE_HIP_CHECK(hipFree(g_ar_dev)); GCGE_HIP_CHECK(hipHostFree(g_ar_pin)); }
    g_ar_cap = (size_t)n * 2 + 1024;
    GCGE_HIP_CHECK(hipMalloc(&g_ar_dev, g_ar_cap * sizeof(double)));
    GCGE_HIP_CHECK(hipHostMalloc(&g_ar_pin, g_ar_cap * sizeof(double)));
  }
  memcpy(g_ar_pin, buf, (size_t)n * sizeof(double));
  GCGE_HIP_CHECK(hipMemcpyAsync(g_ar_dev, g_ar_pin, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  GCGE_NCCL_CHECK(ncclAllReduce(g_ar_dev, g_ar_dev, (size_t)n, ncclDouble, ncclSum, g_nccl, st));
  GCGE_HIP_CHECK(hipMemcpyAsync(g_ar_pin, g_ar_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  GCGE_HIP_CHECK(hipStreamSynchronize(st));
  memcpy(buf, g_ar_pin, (size_t)n * sizeof(double));
  ++g_n_allreduce;
}

// in place on DEVICE memory, on the back-end's stream, nothing waited for (the fused CG's scalars)
extern "C" int gcge_hip_comm_allreduce_device(double* d_buf, int n) {
  if (g_nccl == nullptr || n <= 0) return 0;
  GCGE_NCCL_CHECK(ncclAllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, g_nccl, (hipStream_t)gcge_hip_stream()));
  ++g_n_allreduce;
  return 0;
}

extern "C" int gcge_hip_comm_is_native(const GCGE_COMM* comm) {
  return g_nccl != nullptr && comm != nullptr && comm->allreduce_sum == rccl_allreduce_host;
}

extern "C" int gcge_hip_comm_init(int rank, int world, const void* id128) {
  if (g_nccl != nullptr) { fprintf(stderr, "gcge_hip_comm_init: communicator exists already\n"); return -1; }
  if (world < 1 || rank < 0 || rank >= world || id128 == nullptr) return -2;
  if (gcge_hip_init(-1) != 0) return -3;      // the device was chosen by gcge_hip_init(device) / hipSetDevice before
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  GCGE_NCCL_CHECK(ncclCommInitRank(&g_nccl, world, id, rank));
  g_rank = rank; g_world = world;
  GCGE_HIP_CHECK(hipStreamCreateWithFlags(&g_cs, hipStreamNonBlocking));
  GCGE_HIP_CHECK(hipEventCreateWithFlags(&g_ev_packed, hipEventDisableTiming));
  GCGE_HIP_CHECK(hipEventCreateWithFlags(&g_ev_arrived, hipEventDisableTiming));
  g_comm_desc.rank = rank; g_comm_desc.size = world; g_comm_desc.allreduce_sum = rccl_allreduce_host; g_comm_desc.ctx = nullptr;
  GCGE_SetComm(&g_comm_desc);
  return 0;
}
extern "C" void gcge_hip_comm_finalize(void) {
  if (g_nccl == nullptr) return;
  GCGE_SetComm(nullptr);
  GCGE_HIP_CHECK(hipDeviceSynchronize());
  ncclCommDestroy(g_nccl); g_nccl = nullptr;
  hipStreamDestroy(g_cs); hipEventDestroy(g_ev_packed); hipEventDestroy(g_ev_arrived);
  if (g_ar_dev) { hipFree(g_ar_dev); hipHostFree(g_ar_pin); g_ar_dev = g_ar_pin = nullptr; g_ar_cap = 0; }
  g_rank = 0; g_world = 1;
}
extern "C" int gcge_hip_comm_rank(void) { return g_rank; }
extern "C" int gcge_hip_comm_size(void) { return g_world; }
extern "C" void gcge_hip_comm_stats(long* n_allreduce, long* n_exchange) {
  if (n_allreduce) *n_allreduce = g_n_allreduce;
  if (n_exchange) *n_exchange = g_n_exchange;
}

// ------------------------------------------------------------------ halo plan
struct HaloPlanRccl {
  int npeer;                                  // entries of the per-peer tables (== number of row slabs)
  std::vector<int> peer, send_cnt, recv_cnt;  // communicator rank of every slab, rows to ship to / to receive from it
  std::vector<long> send_off, recv_off;       // row offsets into sendbuf / recvbuf
};

static void post_transfers(const HaloPlanRccl* pl, double* sendbuf, double* recvbuf, int ncols, hipStream_t st) {
  GCGE_NCCL_CHECK(ncclGroupStart());
  for (int q = 0; q < pl->npeer; ++q) {
    if (pl->send_cnt[q] > 0)
      GCGE_NCCL_CHECK(ncclSend(sendbuf + pl->send_off[q] * ncols, (size_t)pl->send_cnt[q] * ncols, ncclDouble, pl->peer[q], g_nccl, st));
    if (pl->recv_cnt[q] > 0)
      GCGE_NCCL_CHECK(ncclRecv(recvbuf + pl->recv_off[q] * ncols, (size_t)pl->recv_cnt[q] * ncols, ncclDouble, pl->peer[q], g_nccl, st));
  }
  GCGE_NCCL_CHECK(ncclGroupEnd());
  ++g_n_exchange;
}
// gcge_halo_exchange_fn: the pack kernel was launched on the back-end's stream, the unpack kernel follows on it:
// the transfers go on the same stream, ordered by it
static void rccl_exchange(double* sendbuf, double* recvbuf, int ncols, void* ctx) {
  post_transfers((const HaloPlanRccl*)ctx, sendbuf, recvbuf, ncols, (hipStream_t)gcge_hip_stream());
}
// split form: the transfers run on their own stream behind an event recorded after the pack kernel ...
static void rccl_exchange_begin(double* sendbuf, double* recvbuf, int ncols, void* ctx) {
  hipStream_t st = (hipStream_t)gcge_hip_stream();
  GCGE_HIP_CHECK(hipEventRecord(g_ev_packed, st));
  GCGE_HIP_CHECK(hipStreamWaitEvent(g_cs, g_ev_packed, 0));
  post_transfers((const HaloPlanRccl*)ctx, sendbuf, recvbuf, ncols, g_cs);
  GCGE_HIP_CHECK(hipEventRecord(g_ev_arrived, g_cs));
}
// ... and whatever the back-end launches after `end` (unpack, boundary strips) waits for their arrival on the device
static void rccl_exchange_end(void* ctx) {
  (void)ctx;
  GCGE_HIP_CHECK(hipStreamWaitEvent((hipStream_t)gcge_hip_stream(), g_ev_arrived, 0));
}

extern "C" void gcge_hip_halo_native_free(GCGE_HIP_MAT_* A) {
  if (A->native_halo == nullptr) return;
  GCGE_HIP_CHECK(hipDeviceSynchronize());
  if (A->sendbuf) hipFree(A->sendbuf);
  if (A->recvbuf) hipFree(A->recvbuf);
  delete (HaloPlanRccl*)A->native_halo;
  A->native_halo = nullptr; A->sendbuf = A->recvbuf = nullptr;
}

// Install a halo plan that runs over RCCL.  The slab matrix A was created with LOCAL column indices
// (gcge_hip_mat_create_local; columns >= nrows are halo rows, ascending by global index).  npeer slabs; for slab q:
// peer[q] = communicator rank that owns it, send_cnt[q] / recv_cnt[q] = rows of X shipped to / received from it;
// send_rows = the local rows to ship, grouped by destination slab in ascending order (sum send_cnt of them).  The
// received rows arrive grouped by source slab in ascending order == the order of the halo rows.
extern "C" int gcge_hip_mat_set_halo_rccl(GCGE_HIP_MAT* A, int nglobal, int npeer, const int* peer, const int* send_cnt,
                                          const int* recv_cnt, const int* send_rows, int buf_cols) {
  if (g_nccl == nullptr) { fprintf(stderr, "gcge_hip_mat_set_halo_rccl: call gcge_hip_comm_init first\n"); return -1; }
  if (A == nullptr || npeer < 1 || buf_cols < 1) return -2;
  HaloPlanRccl* pl = new HaloPlanRccl;
  pl->npeer = npeer;
  long ns = 0, nr = 0;
  for (int q = 0; q < npeer; ++q) {
    if (peer[q] < 0 || peer[q] >= g_world || send_cnt[q] < 0 || recv_cnt[q] < 0) { delete pl; return -3; }
    pl->peer.push_back(peer[q]); pl->send_cnt.push_back(send_cnt[q]); pl->recv_cnt.push_back(recv_cnt[q]);
    pl->send_off.push_back(ns); pl->recv_off.push_back(nr);
    ns += send_cnt[q]; nr += recv_cnt[q];
  }
  if (nr != A->nghost) {
    fprintf(stderr, "gcge_hip_mat_set_halo_rccl: plan receives %ld rows, the matrix has %d halo rows\n", nr, A->nghost);
    delete pl; return -4;
  }
  gcge_hip_halo_native_free(A);
  double *sb = nullptr, *rb = nullptr;
  GCGE_HIP_CHECK(hipMalloc(&sb, std::max<size_t>(1, (size_t)ns * buf_cols) * sizeof(double)));
  GCGE_HIP_CHECK(hipMalloc(&rb, std::max<size_t>(1, (size_t)nr * buf_cols) * sizeof(double)));
  gcge_hip_mat_set_halo(A, nglobal, (int)ns, send_rows, sb, rb, buf_cols, rccl_exchange, pl);   // (checks the row range)
  gcge_hip_mat_set_halo_async(A, rccl_exchange_begin, rccl_exchange_end);
  A->native_halo = pl;
  return 0;
}

// the two int transports of the planner (GCGE_PLAN_TRANSPORT, include/gcge_problems.h) over RCCL: device staging, the back-end's stream
static void rccl_allgather_int(const int* send, int n, int* recv_all, void* ctx) {
  (void)ctx;
  hipStream_t st = (hipStream_t)gcge_hip_stream();
  int *d_s = nullptr, *d_r = nullptr;
  GCGE_HIP_CHECK(hipMalloc(&d_s, std::max(1, n) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&d_r, (size_t)std::max(1, n) * g_world * sizeof(int)));
  GCGE_HIP_CHECK(hipMemcpyAsync(d_s, send, n * sizeof(int), hipMemcpyHostToDevice, st));
  GCGE_NCCL_CHECK(ncclAllGather(d_s, d_r, n, ncclInt32, g_nccl, st));
  GCGE_HIP_CHECK(hipMemcpyAsync(recv_all, d_r, (size_t)n * g_world * sizeof(int), hipMemcpyDeviceToHost, st));
  GCGE_HIP_CHECK(hipStreamSynchronize(st));
  hipFree(d_s); hipFree(d_r);
}
static void rccl_exchange_int(const int* sendbuf, const int* send_cnt, int* recvbuf, const int* recv_cnt, void* ctx) {
  (void)ctx;
  hipStream_t st = (hipStream_t)gcge_hip_stream();
  long ns = 0, nr = 0;
  for (int q = 0; q < g_world; ++q) { ns += send_cnt[q]; nr += recv_cnt[q]; }
  int *d_s = nullptr, *d_r = nullptr;
  GCGE_HIP_CHECK(hipMalloc(&d_s, std::max<long>(1, ns) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&d_r, std::max<long>(1, nr) * sizeof(int)));
  GCGE_HIP_CHECK(hipMemcpyAsync(d_s, sendbuf, ns * sizeof(int), hipMemcpyHostToDevice, st));
  GCGE_NCCL_CHECK(ncclGroupStart());
  long so = 0, ro = 0;
  for (int q = 0; q < g_world; ++q) {
    if (send_cnt[q] > 0) GCGE_NCCL_CHECK(ncclSend(d_s + so, send_cnt[q], ncclInt32, q, g_nccl, st));
    if (recv_cnt[q] > 0) GCGE_NCCL_CHECK(ncclRecv(d_r + ro, recv_cnt[q], ncclInt32, q, g_nccl, st));
    so += send_cnt[q]; ro += recv_cnt[q];
  }
  GCGE_NCCL_CHECK(ncclGroupEnd());
  GCGE_HIP_CHECK(hipMemcpyAsync(recvbuf, d_r, nr * sizeof(int), hipMemcpyDeviceToHost, st));
  GCGE_HIP_CHECK(hipStreamSynchronize(st));
  hipFree(d_s); hipFree(d_r);
}

// Rows [part[rank], part[rank+1]) of a symmetric matrix with GLOBAL column indices (host CSR).  Everything a row-
// partitioned product needs is set up here, collectively (every rank of the communicator calls it): ghost list and
// local renumbering, who needs which of my rows (one all-gather of the per-slab counts, then the index lists by
// grouped send/recv), exchange buffers of buf_cols columns.  part: world + 1 row offsets, part[world] = n_global.
// ... of a matrix on a MASKED grid (gcge_hip_mat_create_grid on one rank): box_of_global_row[r] = x + nx (y + ny z) of global row r,
// rows in scan order, the partition cut between grid LINES (gcge_dist_partition_lines) — the slab keeps the plane sweep, its halo rows
// are found through the line table (spmm_star.hip).  NULL geometry: gcge_hip_mat_create_slab.
extern "C" void gcge_hip_star_next_geometry_cols(int ncols_local, int nx, int ny, int nz, const int* box_of_local_col);
static const int* g_slab_box = nullptr; static int g_slab_dims[3] = {0, 0, 0};
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_slab(const long* part, const int* rowptr, const int* colidx_global, const double* val, int buf_cols);
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_slab_grid(const long* part, const int* rowptr, const int* colidx_global, const double* val, int buf_cols,
                                                       int nx, int ny, int nz, const int* box_of_global_row) {
  g_slab_box = box_of_global_row; g_slab_dims[0] = nx; g_slab_dims[1] = ny; g_slab_dims[2] = nz;
  GCGE_HIP_MAT* A = gcge_hip_mat_create_slab(part, rowptr, colidx_global, val, buf_cols);
  g_slab_box = nullptr;
  return A;
}
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_slab(const long* part, const int* rowptr, const int* colidx_global,
                                                  const double* val, int buf_cols) {
  if (gcge_hip_init(-1) != 0) return nullptr;
  const int world = g_world, rank = g_rank;
  const long n_global = part[world];
  const int nrows = (int)(part[rank + 1] - part[rank]);
  const long nnz = rowptr[nrows];
  if (n_global > INT32_MAX) { fprintf(stderr, "gcge_hip_mat_create_slab: 32-bit indices, n_global = %ld\n", n_global); return nullptr; }
  GCGE_CSR S;
  memset(&S, 0, sizeof(S));
  S.nrows = nrows; S.ncols = (int)n_global; S.row_begin = (int)part[rank]; S.nnz = nnz;
  S.rowptr = (int*)rowptr;
  std::vector<int> cols(colidx_global, colidx_global + nnz);     // localised copy; the caller's array stays global
  S.colidx = cols.data(); S.val = (double*)val;
  int* ghosts = nullptr; int ng = 0;
  if (gcge_dist_ghosts(&S, &ghosts, &ng) != 0) return nullptr;
  std::vector<int> recv_cnt(world, 0), send_cnt(world, 0), peer(world);
  for (int q = 0; q < world; ++q) peer[q] = q;
  if (world > 1 && g_nccl == nullptr) { fprintf(stderr, "gcge_hip_mat_create_slab: call gcge_hip_comm_init first\n"); return nullptr; }
  // the plan comes from the ONE planner of the host library (gcge_dist_plan_halo, csrc/host/problems.c) — the function the gloo
  // tests run with 2 / 3 / 8 ranks; only the two int transports below are RCCL's own
  GCGE_PLAN_TRANSPORT tr;
  tr.rank = rank; tr.size = world; tr.allgather_int = rccl_allgather_int; tr.exchange_int = rccl_exchange_int; tr.ctx = nullptr;
  int* srows = nullptr; int nsend = 0;
  const int prc = gcge_dist_plan_halo(part, ghosts, ng, &tr, recv_cnt.data(), send_cnt.data(), &srows, &nsend);
  if (prc != 0) { fprintf(stderr, "gcge_hip_mat_create_slab: halo plan failed (%d)\n", prc); gcge_free_ints(ghosts); return nullptr; }
  std::vector<int> send_rows(srows, srows + nsend);
  gcge_free_ints(srows);
  if (gcge_dist_localize(&S, ghosts, ng) != 0) { gcge_free_ints(ghosts); return nullptr; }
  // (the halo rows' global ids travel with the arrays: a slab of a grid matrix cut on plane boundaries keeps the plane sweep)
  std::vector<int> box_local;
  if (g_slab_box != nullptr) {                                       // masked grid: the box index of every local column (own rows, then halo rows)
    box_local.resize((size_t)nrows + ng);
    for (int r = 0; r < nrows; ++r) box_local[r] = g_slab_box[part[rank] + r];
    for (int k = 0; k < ng; ++k) box_local[(size_t)nrows + k] = g_slab_box[ghosts[k]];
    gcge_hip_star_next_geometry_cols(nrows + ng, g_slab_dims[0], g_slab_dims[1], g_slab_dims[2], box_local.data());
  }
  GCGE_HIP_MAT* A = gcge_hip_mat_create_local_ghosts(nrows, nrows + ng, (int)n_global, (int)part[rank], rowptr, cols.data(), val, ghosts);
  if (g_slab_box != nullptr) gcge_hip_star_next_geometry_cols(0, 0, 0, 0, nullptr);   // (not consumed when the slab took another form)
  gcge_free_ints(ghosts);
  if (A == nullptr) return nullptr;
  if (world > 1) {
    const int rc = gcge_hip_mat_set_halo_rccl(A, (int)n_global, world, peer.data(), send_cnt.data(), recv_cnt.data(),
                                              send_rows.data(), buf_cols);
    if (rc != 0) { gcge_hip_mat_destroy(A); return nullptr; }
  }
  gcge_hip_mat_set_partition(A, part, world);      // (MultiGridCreate coarsens a slab from it: multigrid.hip)
  return A;
}
