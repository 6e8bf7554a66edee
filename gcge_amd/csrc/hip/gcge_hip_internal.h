// Internal helpers shared by the HIP translation units of libgcge_hip.so.
#ifndef GCGE_HIP_INTERNAL_H
#define GCGE_HIP_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define GCGE_HIP_CHECK(expr)                                                          \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "gcge_hip: %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__, \
              hipGetErrorString(e_));                                                 \
      abort();                                                                        \
    }                                                                                 \
  } while (0)

// Shape contract of every slot is checked on the HOST before a kernel is launched: a
// mismatch must abort here, never turn into an out-of-bounds access on the device.
#define GCGE_REQUIRE(cond, what)                                                        \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      fprintf(stderr, "gcge_hip: %s violated (%s) at %s:%d\n", what, #cond, __FILE__, __LINE__); \
      abort();                                                                          \
    }                                                                                   \
  } while (0)

#ifdef __cplusplus
#include <thread>
#include <vector>
#include "gcge_hip.h"
// host-side analysis of a matrix at upload: fn(chunk, first, last) over [0, n) cut into contiguous chunks, one thread each
// (GCGE_UPLOAD_THREADS, default: the cores the process may use, at most 16)
static inline int gcge_upload_threads() {
  const char* e = getenv("GCGE_UPLOAD_THREADS");
  int t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
  if (t < 1) t = 1;
  if (t > 16) t = 16;
  return t;
}
template <class F>
static inline void gcge_parallel_chunks(long n, int nchunks, F fn) {
  if (nchunks <= 1 || n < 65536) { for (int c = 0; c < nchunks; ++c) fn(c, n * c / nchunks, n * (c + 1) / nchunks); return; }
  std::vector<std::thread> th;
  for (int c = 1; c < nchunks; ++c) th.emplace_back([=]() { fn(c, n * c / nchunks, n * (c + 1) / nchunks); });
  fn(0, 0L, n / nchunks);
  for (auto& t : th) t.join();
}
// sparse matrix handle (CCSMAT counterpart): shared by app_hip.hip (the slots) and rccl_comm.hip (the halo plan)
struct GCGE_HIP_MAT_ {
  int nrows;      // local rows
  int nglobal;    // global dimension
  int row_begin;  // first global row
  int nghost;     // halo rows appended to every block of vectors
  long nnz;
  int *d_rowptr, *d_colidx; double* d_val;    // CSR, LOCAL column indices (ghosts >= nrows)
  int *d_orp, *d_pcol; double* d_pval;        // pad-8 copy for the 16-byte-lane kernel
  long noct;
  unsigned short* d_pid; void* d_tab; int npat, pat_lt; long pat_span, pat_span2;   // pattern format (spmm_pattern.hip); d_pid == NULL: not applicable
  double* d_rowval;   // patterns by OFFSETS only: the rows' values, 8 doubles per row in table-slot order (NULL: values in the table)
  long pat_near;  // > 0: chain + line table with slots [-S, 0, +S, -L, +L, -1, +1], the largest |offset| in it (spmm_ring.hip)
  // halo plan of a row-partitioned matrix (one process per GPU); nghost == 0 on a single rank
  int nsend; int* d_send_rows;                 // local rows other ranks need, grouped by destination rank
  double *sendbuf, *recvbuf; int buf_cols;     // exchange buffers (owned by the caller: torch tensors)
  gcge_halo_exchange_fn exchange; void* exchange_ctx;
  // optional split exchange (begin posts the transfers and returns, end completes them) and the rows that do not
  // touch a halo column, [ov_lo, ov_hi): lets the interior product run while the halo is in flight
  gcge_halo_exchange_fn exchange_begin; void (*exchange_end)(void*); int ov_lo, ov_hi;
  void* dense;         // supernode form (spmm_dense.hip): dense row blocks on MFMA + remainder CSR; NULL: no blocks found
  void* tile;          // LDS-staged X-tile form (spmm_tile.hip) of a matrix without a pattern form; NULL: generic kernels
  void* star;          // grid form (spmm_star.hip): rows that are exactly a star stencil, swept plane by plane; NULL: none
  void* star_rem;      // block form (spmm_dense.hip) of the rows the grid form leaves (multiplied first, writes every row)
  void* native_halo;   // RCCL plan of gcge_hip_mat_set_halo_rccl (rccl_comm.hip); it then owns sendbuf / recvbuf
  // rectangular matrices (the prolongations P_l of a multigrid hierarchy, multigrid.hip): nrows x rect_ncols in d_rowptr / d_colidx /
  // d_val, the transpose as a second CSR triple; rect_ncols == 0: an ordinary (symmetric) matrix
  int rect_ncols; int *d_t_rowptr, *d_t_colidx; double* d_t_val;
  int rect_one_per_row;   // 1: every row of a rectangular matrix holds exactly one entry (aggregation prolongations: the fused x += P e)
  // row slabs: the global rows behind the halo columns (host copy, nghost ints; NULL: not named) and the row partition of all ranks
  // (host, part_world + 1 entries; NULL: unknown) — what MultiGridCreate needs to coarsen a slab (multigrid.hip)
  int* h_ghost_global; long* h_part; int part_world;
  // round 5: the row order the back-end chose for itself (mat_upload.hip "row orders"): the device arrays hold P A P^T, every block of
  // vectors created for this matrix lives in the same order; NULL: the caller's order
  struct GcgePerm* perm;
};
// One row order per problem size and process: every matrix of n rows (A, then B of a generalised problem) and every block of vectors
// created for them share it.  perm[new] = old, iperm[old] = new; identity: a matrix of this size was uploaded in the caller's order and
// later ones of the same size must follow it.  Reference-counted by matrices and blocks.
struct GcgePerm { int n; int* perm; int* iperm; int identity; long refs; unsigned id; };
extern "C" struct GcgePerm* gcge_hip_perm_acquire(struct GcgePerm* p);      // ++refs (NULL passes through)
extern "C" void gcge_hip_perm_release(struct GcgePerm* p);
extern "C" void gcge_hip_halo_native_free(struct GCGE_HIP_MAT_* A);
// multigrid.hip: the MultiGridCreate / MultiGridDestroy slots of OPS_HIP_Set and the block-CG smoother it registers for BlockAMG
extern "C" void gcge_hip_multigrid_create(void*** A_array, void*** B_array, void*** P_array, int* num_levels, void* A, void* B, struct OPS_* ops);
extern "C" void gcge_hip_multigrid_destroy(void*** A_array, void*** B_array, void*** P_array, int* num_levels, struct OPS_* ops);
extern "C" void gcge_hip_amg_smoother_setup(int max_iter, double rate, double tol, const char* tol_type, struct OPS_* ops);
extern "C" double gcge_hip_amg_smoother_residual(struct OPS_* ops);
// A column scaling the slots hold back (column-wise Gram-Schmidt, app_hip.hip) is applied now.  First statement of every EXPORTED
// raw kernel that takes device pointers: the caller may have fetched its pointer before the scaling was held back.
extern "C" void gcge_hip_apply_pending(void);

#endif

#endif
