/* gcge_hip.h — C ABI of the MI355X (gfx950) back-end, libgcge_hip.so.
 *
 * Drop-in boundary: `OPS_HIP_Set` fills the operator table of gcge_ops.h exactly as
 * the reference's built-in back-ends do (pattern: app/app_ccs.c:213-249 OPS_CCS_Set,
 * app/app_slepc.c:610-634 OPS_SLEPC_Set), so GCGE's solver layers — the
 * reference's or ours — run with every O(n) operand resident in HBM.
 * Everything below is plain C: pointers and sizes only, no C++/torch types.
 *
 * Device layout of a block of vectors: ROW-major, element (r,c) at d[r*ld + c],
 * ld a multiple of 8 doubles; rows [nrows, nrows+nghost) are halo rows used only
 * while a row-partitioned SpMM runs.  Handles are opaque to the solver
 * (the reference never dereferences them: SURVEY.md §8b "Layout opacity").
 */
#ifndef GCGE_HIP_H
#define GCGE_HIP_H

#include <stddef.h>
#include "gcge_ops.h"
#include "gcge_problems.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ---------------------------------------------------------------- */
int  gcge_hip_init (int device);              /* hipSetDevice + workspace; 0 on success   */
void gcge_hip_finalize (void);
int  gcge_hip_device_count (void);
void gcge_hip_sync (void);
void *gcge_hip_stream (void);                 /* hipStream_t all kernels are launched on  */

/* ---- back-end registration (replaces OPS_CCS_Set, app/app_ccs.c:213-249) ------ */
void OPS_HIP_Set (struct OPS_ *ops);

/* ---- the driver a maintainer calls from test/main.c:40-49 (counterpart of TestAppCCS, test/test_app_ccs.c:86-140):
 * table + OPS_HIP_Set + OPS_Setup, matrices from a generator or a file, TestEigenSolverGCG(A, B, flag, argc, argv, ops).
 * Options: csrc/host/test_app_hip.c (-hip_problem, -hip_size, -hip_petsc_A/B, -hip_mtx_A/B, -hip_flag, -hip_device).  */
int TestAppHIP (int argc, char *argv[]);

/* ---- sparse matrix handle (replaces CCSMAT, app/app_ccs.h:20-24) -------------- */
typedef struct GCGE_HIP_MAT_ GCGE_HIP_MAT;
/* rows [row_begin,row_begin+nrows) of a symmetric matrix, GLOBAL column indices, host CSR
 * (== the CCS triple j_col/i_row/data).  Single rank: row_begin = 0, nrows = nglobal. */
GCGE_HIP_MAT *gcge_hip_mat_create (int nrows, int nglobal, int row_begin,
		const int *rowptr, const int *colidx, const double *val);
GCGE_HIP_MAT *gcge_hip_mat_create_csr (const GCGE_CSR *A);
/* the same for a matrix on a MASKED grid (one rank): row r is grid point box_of_row[r] = x + nx (y + ny z) of an nx x ny x nz box,
 * rows in scan order (ascending box index) — the grid points inside a sphere of the PARSEC matrices the reference's
 * test/submit.sh:9-15 lists.  Naming the geometry lets the star-stencil rows take the plane sweep (spmm_star.hip, through a row
 * map); results are those of gcge_hip_mat_create on the same arrays.                                                          */
GCGE_HIP_MAT *gcge_hip_mat_create_grid (int nrows, const int *rowptr, const int *colidx, const double *val,
		int nx, int ny, int nz, const int *box_of_row);
/* A matrix of that kind that names NO geometry (read from a file: gcge_load_matrix_market, gcge_load_petsc_binary) gets it
 * recovered at upload by gcge_hip_mat_create itself: x lines from the (r, r + 1) couplings, planes and the shifts between
 * lines / planes from the votes of the star rows' + y / + z neighbours.  A wrong guess costs speed, never the result (the
 * remainder takes every difference); on = 0 switches the recovery off (measurements).                                         */
void gcge_hip_spmm_star_infer (int on);
/* Row-partitioned use (one process per GPU): localize the slab with gcge_dist_localize
 * (include/gcge_problems.h), create it with ncols_local = nrows + nghost, then install the halo
 * plan.  exchange(sendbuf, recvbuf, ncols, ctx) must deliver, for every peer, rows
 * [send_off[p], +send_cnt[p]) x ncols of sendbuf into the peer's recvbuf at its ghost offset for
 * this rank (both buffers row-major with ncols columns; device memory for this back-end).    */
typedef void (*gcge_halo_exchange_fn) (double *sendbuf, double *recvbuf, int ncols, void *ctx);
GCGE_HIP_MAT *gcge_hip_mat_create_local (int nrows, int ncols_local, int nglobal, int row_begin,
		const int *rowptr, const int *colidx, const double *val);
/*     the same with the global rows behind the halo columns (ascending, as gcge_dist_ghosts lists them) and the global size:
 *     a slab of a grid matrix cut on plane boundaries then keeps the plane sweep of spmm_star.hip — the planes below and
 *     above the slab are found among its halo rows (gcge_hip_mat_create_slab passes them by itself)                      */
GCGE_HIP_MAT *gcge_hip_mat_create_local_ghosts (int nrows, int ncols_local, int nglobal, int row_begin,
		const int *rowptr, const int *colidx, const double *val, const int *ghost_global);
void gcge_hip_mat_set_halo (GCGE_HIP_MAT *A, int nglobal, int nsend, const int *send_rows,
		double *sendbuf, double *recvbuf, int buf_cols, gcge_halo_exchange_fn fn, void *ctx);
/*     optional split form of the exchange: begin posts the transfers of the packed rows and returns, end returns
 *     when recvbuf is complete; the back-end then multiplies the rows that touch no halo column in between      */
void gcge_hip_mat_set_halo_async (GCGE_HIP_MAT *A, gcge_halo_exchange_fn begin, void (*end) (void *ctx));
void gcge_hip_set_halo_overlap (int on);
void gcge_hip_mat_destroy (GCGE_HIP_MAT *A);
/* ---- multigrid (csrc/hip/multigrid.hip): OPS_HIP_Set fills ops->MultiGridCreate / MultiGridDestroy (reference slots
 * src/ops.h:134-139; what app/app_slepc.c:648-728 gets from PETSc GAMG): the CSR arrays come back from the device, the
 * aggregation hierarchy of include/gcge_multigrid.h is built on the host (2 x 2 x 2 cells of a detected grid, greedy aggregates
 * otherwise, A_{l+1} = scale P^T A_l P; gcge_mg_set_defaults), every level is uploaded like any other matrix, and the fused
 * block CG is registered as the smoother of BlockAMG for this table (GCGE_SetBlockAMGSmoother, include/gcge_solver.h).
 * One rank only.  A prolongation is a RECTANGULAR matrix handle: MatDotMultiVec applies P (rows of level l x rows of level
 * l + 1), MatTransDotMultiVec its transpose (the restriction of src/ops_multi_grid.c:95-113); MultiVecCreateByMat of it gives
 * blocks with its COLUMN count of rows (app_ccs.c:43).                                                                      */
GCGE_HIP_MAT *gcge_hip_mat_create_rect (int nrows, int ncols, const int *rowptr, const int *colidx, const double *val,
		const int *t_rowptr, const int *t_colidx, const double *t_val);      /* CSR of P and CSR of P^T */
GCGE_HIP_MAT *gcge_hip_mat_create_rect_csr (const GCGE_CSR *P);              /* the transpose is formed here */
double gcge_hip_multigrid_seconds (void);    /* host + upload time of the last MultiGridCreate */
/*     Row slabs (one rank per GPU): a slab that is whole planes of a detected grid (cut on any plane boundary) coarsens by itself
 *     (gcge_mg_build_slab, include/gcge_multigrid.h: local prolongations, coarse slabs with global columns); every coarse slab goes
 *     through the slab constructor — gcge_hip_mat_create_slab over RCCL by default, or the function registered here (a transport of
 *     the caller's: the torch.distributed callbacks of the tests).  MultiGridCreate needs the partition of all ranks:
 *     gcge_hip_mat_create_slab records it, other constructors call gcge_hip_mat_set_partition.  Collective.                     */
typedef GCGE_HIP_MAT *(*gcge_hip_slab_factory_fn) (const long *part, int world, int rank, const int *rowptr, const int *colidx_global,
		const double *val, int buf_cols, void *ctx);
void gcge_hip_set_slab_factory (gcge_hip_slab_factory_fn fn, void *ctx);     /* NULL: gcge_hip_mat_create_slab */
void gcge_hip_mat_set_partition (GCGE_HIP_MAT *A, const long *part, int world);
int  gcge_hip_mat_nrows (const GCGE_HIP_MAT *A);
long gcge_hip_mat_nnz (const GCGE_HIP_MAT *A);

/* ---- multi-GPU from C: RCCL inside the back-end (csrc/hip/rccl_comm.hip) ------------------------------------
 * One process per GPU.  Replaces what the reference does with MPI in its solver layers — MPI_Allreduce of every
 * Gram / dot result (src/ops_multi_vec.c:206-228, src/ops_lin_sol.c:313-321,361-369) — and in its distributed
 * back-ends (off-process part of X inside MatDotMultiVec, app/app_phg.c:292-359).
 *   rank 0:  gcge_hip_comm_unique_id(id);  hand the 128 bytes to every rank (MPI_Bcast, a file, torch.distributed ...)
 *   all:     gcge_hip_init(local_device); gcge_hip_comm_init(rank, world, id);   // also installs GCGE_COMM (gcge_ops.h)
 *            A = gcge_hip_mat_create_slab(part, rowptr, colidx_global, val, block_columns);   // collective
 *            OPS_HIP_Set(ops); ... the solver as on one GPU ...; gcge_hip_comm_finalize();                        */
#define GCGE_HIP_COMM_ID_BYTES 128
int  gcge_hip_comm_unique_id (void *id128);
int  gcge_hip_comm_init (int rank, int world, const void *id128);
void gcge_hip_comm_finalize (void);
int  gcge_hip_comm_rank (void);
int  gcge_hip_comm_size (void);
void gcge_hip_comm_stats (long *n_allreduce, long *n_exchange);
/*     in-place sum over the ranks of n doubles in DEVICE memory on the back-end's stream, nothing waited for      */
int  gcge_hip_comm_allreduce_device (double *d_buf, int n);
/*     1 if `comm` (GCGE_GetComm()) is the one gcge_hip_comm_init installed, i.e. device-side all-reduces are possible  */
int  gcge_hip_comm_is_native (const GCGE_COMM *comm);
/*     rows [part[rank], part[rank+1]) of a symmetric matrix, GLOBAL column indices (host CSR); part has world + 1
 *     entries.  Collective: builds the ghost list, the local numbering and the halo plan (who needs which rows) over
 *     RCCL; products then move buf_cols columns of halo rows per grouped ncclSend/ncclRecv, event-ordered            */
GCGE_HIP_MAT *gcge_hip_mat_create_slab (const long *part, const int *rowptr, const int *colidx_global,
		const double *val, int buf_cols);
/*     ... of a matrix on a MASKED grid (the real-space DFT matrices behind BASELINE config 5 on more than one device; reference
 *     test/submit.sh:9-15): box_of_global_row[r] = x + nx (y + ny z) of GLOBAL row r (rows in scan order), the partition cut between
 *     grid lines (gcge_amd.dist.partition_lines).  The slab keeps the plane sweep: own and halo rows are found through one line table.
 *     gcge_hip_star_next_geometry_cols: the same for callers that build the slab themselves (gcge_hip_mat_create_local_ghosts): the
 *     box index of every LOCAL column (own rows, then halo rows), consumed by the next upload                                      */
GCGE_HIP_MAT *gcge_hip_mat_create_slab_grid (const long *part, const int *rowptr, const int *colidx_global, const double *val,
		int buf_cols, int nx, int ny, int nz, const int *box_of_global_row);
void gcge_hip_star_next_geometry_cols (int ncols_local, int nx, int ny, int nz, const int *box_of_local_col);
/*     the same for a slab matrix that already has LOCAL column indices (gcge_hip_mat_create_local) and a plan computed
 *     elsewhere: npeer slabs, peer[q] = communicator rank owning slab q, rows shipped to / received from it, the
 *     local rows to ship grouped by destination slab                                                               */
int  gcge_hip_mat_set_halo_rccl (GCGE_HIP_MAT *A, int nglobal, int npeer, const int *peer, const int *send_cnt,
		const int *recv_cnt, const int *send_rows, int buf_cols);

/* ---- block-of-vectors transfers (tests, final eigenvectors) ------------------- */
/* columns [c0,c1) <-> host column-major array with leading dimension ldh (>= nrows) */
void gcge_hip_mv_to_host   (void **mv, int c0, int c1, double *host, long ldh);
void gcge_hip_mv_from_host (void **mv, int c0, int c1, const double *host, long ldh);
int  gcge_hip_mv_nrows (void **mv);
int  gcge_hip_mv_ncols (void **mv);
double *gcge_hip_mv_device_ptr (void **mv, long *ld);

/* 0: MultiVecSetRandomValue draws glibc rand() on the host in the reference's order
 *    (app/app_lapack.c:322-333) and uploads — bit-identical start vectors;
 * 1: counter-based generator on the device (for n ~ 1e7, where 2e9 rand() calls
 *    would dominate the run).                                                      */
void gcge_hip_set_random_mode (int mode, unsigned long long seed);

/* ---- fused block CG behind ops->MultiLinearSolver ------------------------------
 * Same recurrence/stopping rules as BlockPCG (src/ops_lin_sol.c:140-437) with the vector
 * work of an iteration fused into four launches.  GCG uses it through the reference's
 * user_defined_multi_linear_solver = 1 hook (ops_eig_sol_gcg.c:584-618, test_app_ccs.c:109-120):
 * call this once, then run the harness with flag = 1.                                  */
void gcge_hip_bpcg_setup (struct OPS_ *ops, int max_iter, double rate, double tol, const char *tol_type);
void gcge_hip_bpcg_stats (long *spmm_calls, long *spmm_cols, int *last_niter);
/* optional, after gcge_hip_bpcg_setup: create the solver's own blocks (r, p, w and the ring of direction slots) for systems with the
 * rows of mv_like and up to ncols right-hand sides now rather than inside the first solve — the reference hands BlockPCG its blocks
 * from EigenSolverCreateWorkspace_GCG, before the harness starts its timer (test/test_eig_sol_gcg.c:89-143).  Returns the ring
 * length (>= 1), -1 when ops->MultiLinearSolver is not this solver.  Collective when a communicator exists.                    */
int  gcge_hip_bpcg_prepare (struct OPS_ *ops, void *mat, void **mv_like, int ncols);
/*     tol_type: "abs", "rel" or "user" (src/ops_lin_sol.c:175-197; "user": scales from GCGE_GetLinearSolverUserScale).
 *     Residual of the recompute form: 0 automatic (not stored where rate >= 1e-4 and max_iter <= 100), 1 never stored
 *     (r_k = p_k - beta_{k-1} p_{k-1} rebuilt from the ring), 2 always stored                                         */
void gcge_hip_bpcg_residual_form (int form);
long gcge_hip_bpcg_surplus_iters (void);   /* iterations enqueued after the last column retired (device-scalar loop) */
/* iterations the fused solver ran in its recompute form (pattern matrices: the product A p is formed twice per
 * iteration and never stored, gcge_hip_cg_pass1_mv / gcge_hip_cg_pass2_mv); GCGE_CG_NO_RECOMPUTE=1 switches it off */
long gcge_hip_bpcg_recompute_iters (void);
/* of those, iterations whose scalars (alpha, beta, stopping test) were computed on the device, without a host round trip
 * inside the iteration (single rank or RCCL inside the back-end; GCGE_CG_HOST_SCALARS=1 switches it off)            */
long gcge_hip_bpcg_device_scalar_iters (void);
/* iterations of that device-scalar loop with the product STORED (matrices without a pattern form, no shift: product + column sums
 * left on the device by gcge_hip_spmm_dot2_dev, then one sweep over w and the directions); GCGE_CG_STORED_HOST=1: host scalars */
long gcge_hip_bpcg_stored_dev_iters (void);
/* solves with right-hand sides b = x diag(scale) (GCGE_SetLinearSolverRhsScale) on such a matrix that started as product + ONE sweep
 * (r = x diag(scale) - A x, p0 = r, r.r) instead of forming b; GCGE_CG_NO_FUSED_START=1 switches it off */
long gcge_hip_bpcg_fused_starts (void);
/*     y[:, cy : cy + m) = A x[:, cx : cx + m) with d_out[0, m) = x.y and d_out[m, 2m) = y.y (local rows) left on the device, nothing
 *     waited for; -1 (nothing touched): odd widths / offsets or unaligned blocks                                                */
int gcge_hip_spmm_dot2_dev (void *mat, void **x, void **y, int cx, int cy, int m, double *d_out);
int gcge_hip_spmm_dot2_dev_ok (void *mat, void **x, void **y, int cx, int cy, int m);   /* 1: it takes these operands */
/* columns the fused solver streamed, summed over its iterations, and how many of them were still active */
void gcge_hip_bpcg_column_stats (long *col_iters, long *active_col_iters);
/* CG iterations and host wall time spent inside the fused solver since the last reset (ms per CG iteration) */
void gcge_hip_bpcg_time_stats (long *iters, double *seconds, int reset);
void gcge_hip_bpcg_release (struct OPS_ *ops);

/* ---- live measurement of the K1 launches (HIP events on the launch stream) ---------- */
void gcge_hip_profile_enable (int on);        /* also clears what was recorded            */
long gcge_hip_profile_spmm (int ncols, double *total_ms, double *total_alg_bytes);
/*     by kind: 0 = MatDotMultiVec products (what gcge_hip_profile_spmm returns), 2 / 3 = first / second pass of a
 *     block-CG iteration in its recompute form (below); ncols == 0: all widths                                    */
long gcge_hip_profile_kind (int kind, int ncols, double *total_ms, double *total_alg_bytes);
/*     ... restricted to launches on matrices of nrows local rows (0: all): with BlockAMG the same kernels run on every level of the
 *     hierarchy, and a roofline figure belongs to one problem size                                                              */
long gcge_hip_profile_kind_rows (int kind, int ncols, long nrows, double *total_ms, double *total_alg_bytes);

/* ---- K7: small dense symmetric eigensolver on the device (replaces dsyevx, src/ops_eig_sol_gcg.c:1201-1203) ------
 * all eigenpairs of the symmetric n x n matrix a (HOST, column-major, ld lda; triangle `uplo` is read): w ascending,
 * z (HOST, ld ldz) orthonormal eigenvectors.  Householder tridiagonalisation and the accumulation of Q on the device,
 * implicit QL on the host with recorded rotations, replayed on the device (csrc/hip/eig_device.hip).  0 on success.  */
int gcge_hip_symeig (char uplo, int n, const double *a, int lda, double *w, double *z, int ldz);
long gcge_hip_symeig_calls (void);   /* calls so far: the hook is keyed to the HIP table (GCGE_SetSymEigHook owner), the CPU oracle never gets here */

/* ---- raw kernels (what the slots launch; exposed for micro-benchmarks) --------- */
/* K1  Y[:,0:m) = A X[:,0:m);  x/y point at (row 0, first column); see csrc/hip/spmm*.hip */
int gcge_hip_csr_spmm  (int nrows, const int *d_rowptr, const int *d_colidx, const double *d_val,
		const double *d_x, long ldx, double *d_y, long ldy, int ncols, void *stream);
int gcge_hip_pad8_spmm (int nrows, const int *d_orp, const int *d_pcol, const double *d_pval,
		const double *d_x, long ldx, double *d_y, long ldy, int ncols, void *stream);
/* K2  G(k x m, row-major on device, ld m) = Q[:,0:k)^T P[:,0:m)  over nrows rows (MFMA f64) */
int gcge_hip_gram (int nrows, const double *d_q, long ldq, int k, const double *d_p, long ldp, int m,
		double *d_g, void *stream);
/*     blocks of vectors freed through MultiVecDestroy are kept by size and reused (hipMalloc/hipFree of multi-GB
 *     blocks cost ~0.3 s each); release returns them to the driver, enable(0) switches the cache off            */
void gcge_hip_pool_release (void);
void gcge_hip_pool_enable (int on);
size_t gcge_hip_pool_cached_bytes (void);
/*     row orders (csrc/hip/mat_upload.hip, reorder.hip): a whole matrix (gcge_hip_mat_create, one rank) that shows neither a pattern
 *     form nor a grid in the order it arrives in is re-ordered inside the handle — grid coordinates recovered from the graph of a
 *     star stencil (scan order: the plane sweep applies again) or reverse Cuthill-McKee.  Blocks of vectors created for it live in
 *     the same order; gcge_hip_mv_to_host / from_host translate; a later matrix of the same size (B) adopts the order.
 *     mode: 0 automatic (>= 65 536 rows), 1 every matrix without a fast form (tests), -1 never                                   */
void gcge_hip_spmm_reorder_mode (int mode);
const char *gcge_hip_mat_row_order (const GCGE_HIP_MAT *A);      /* "as given" or what the upload did */
long gcge_hip_reorder_star_grid (int n, const int *rowptr, const int *colidx, const double *val, int *dims, int *box_of_row);   /* host only */
int  gcge_hip_reorder_rcm (int n, const int *rowptr, const int *colidx, int *perm);                                            /* host only: perm[new] = old */
void gcge_hip_set_spmm_path (int path);   /* 0 automatic (pattern > star rows + block form of the rest > dense blocks + remainder > pad-8 > CSR), 2 no pattern kernels (nor the star sweep), 3 pad-8 / CSR only, 4 no dense blocks */
/*     tile path (csrc/hip/spmm_tile.hip): matrices without a pattern form whose rows are long enough (>= 12 entries on
 *     average; automatic rule: see mode) are additionally kept as row tiles (bricks of a detected grid, or runs of rows) with 16-bit positions into
 *     the tile's list of X rows, which the kernel stages in LDS once per 8-column pass.  mode: 0 automatic (the remainder
 *     of a matrix whose long rows went into dense blocks, when it has short rows on a detected grid), 1 every matrix
 *     without a pattern form, 2 every matrix, -1 never (takes effect at the next gcge_hip_mat_create*)                  */
void gcge_hip_spmm_tile_mode (int mode);
/*     supernode path (csrc/hip/spmm_dense.hip): row sets that share a column set (the dense blocks real-space DFT
 *     Hamiltonians keep per atom) are found at upload and multiplied as dense blocks on FP64 MFMA, the rest of the
 *     matrix through the generic kernels.  mode: 0 automatic (seeds: rows of >= 96 entries, blocks must hold >= 10 % of
 *     the non-zeros), 1 rows of >= 24 entries may seed (tests), -1 never                                                */
void gcge_hip_spmm_dense_mode (int mode);
long gcge_hip_dense_selfcheck (int nrows, int ncols_local, const int *rowptr, const int *colidx, const double *val,
		int min_len, long *nblocks, double *share, double *fill);   /* host-only: blocks + remainder == CSR, bit for bit */
/*     grid path (csrc/hip/spmm_star.hip): matrices without a pattern form on a lexicographic 3-D grid whose rows are mostly
 *     ONE star stencil of arm length <= 6 with a diagonal of their own (the finite-difference Laplacian + local potential of
 *     real-space DFT Hamiltonians): those rows leave the CSR arrays and are multiplied by a plane sweep (z-neighbours in
 *     registers, x / y arms from LDS, 2.5 X rows fetched per row, no matrix stream), the other rows keep all their entries
 *     and take the block form.  Needs blocks among the other rows.  Row slabs (one process per GPU) keep the form when
 *     they are whole grid planes: the planes below / above come from the halo rows, and with a split exchange the planes that
 *     need none are swept while the halo is in flight (reference: app/app_phg.c:307-357).  mode: 0 automatic, -1 never   */
void gcge_hip_spmm_star_mode (int mode);
long gcge_hip_star_selfcheck (int nrows, int ncols_local, const int *rowptr, const int *colidx, const double *val,
		long *out /* nx, ny, nz, arm length, star rows */);   /* host-only: star rows + remainder == CSR, bit for bit; -1: no such form */
/*     the same for a row slab with LOCAL columns (halo column nrows + i = global row ghost[i]): additionally checks that the
 *     sweep's own addressing of the planes below / above lands on the halo rows the CSR arrays name.  out[5..10] = first /
 *     last + 1 plane of the slab, of the planes the sweep may load, first halo row of the planes below / above (-1: none) */
long gcge_hip_star_selfcheck_slab (int nrows, int ncols_local, long row_begin, long nglobal, const int *ghost,
		const int *rowptr, const int *colidx, const double *val, long *out);
/*     the same for a matrix on a masked grid (gcge_hip_mat_create_grid)                                                  */
long gcge_hip_star_selfcheck_grid (int nrows, const int *rowptr, const int *colidx, const double *val, int nx, int ny, int nz,
		const int *box_of_row, long *out);
/*     the geometry gcge_hip_mat_create recovers for a matrix on a masked grid that names none (host only): dims[0..2] = nx, ny,
 *     nz of the bounding box, box_of_row[r] = x + nx (y + ny z); 1 found, 0: the rows are no such domain in scan order    */
int  gcge_hip_star_infer_grid (int nrows, const int *rowptr, const int *colidx, int *dims, int *box_of_row);
/*     grid of a matrix whose rows are mostly one star stencil, from a slab of its rows with GLOBAL columns (host only;
 *     what a partitioner needs to cut on plane boundaries).  out[0..3] = nx, ny, nz, arm length; 1 found, 0 none         */
int  gcge_hip_star_grid (int nrows, long row_begin, long nglobal, const int *rowptr, const int *colidx_global,
		const double *val, long *out);
int  gcge_hip_mat_star_stats (const GCGE_HIP_MAT *A, long *out /* nx, ny, nz, arm length, star rows, rows, first / last + 1 plane of the slab */);   /* 0: the matrix has no grid form */
/* masked grids (gcge_hip_mat_create_grid, or a geometry recovered at upload): 3 = swept by the third form through a LINE table (every
 * grid line one run of rows), 2 = by the second form through a point-wise row map, 0 = every grid point is a row / no grid form.
 * gcge_hip_spmm_star_masked_third(0) keeps the second form (measurements).                                            */
int  gcge_hip_mat_star_masked_form (const GCGE_HIP_MAT *A);
void gcge_hip_spmm_star_masked_third (int on);
/* K1 block form: rows of at least `len` entries seed a dense block (default 96); at most `layers` launches of the block kernel (rows
 * inside overlapping blocks get the later ones), later layers seeded by rows of at least `seed_len` entries (defaults 4, 32)       */
void gcge_hip_spmm_dense_min_len (int len);
void gcge_hip_spmm_dense_layers (int layers, int seed_len);
void gcge_hip_spmm_pad8_acc_early (int on);      /* adding row lists request their Y rows when a wave starts (default) */
void gcge_hip_star_product_stats (long *products, long *split);   /* products through the grid form so far, and how many swept their interior planes while the halo travelled */
/*     stencils whose coefficients differ from row to row: the pattern table is then built from the rows' column OFFSETS
 *     only and the values are streamed per row (8 doubles per row), so such matrices keep the pattern kernels (tables of
 *     at most 8 slots); 0 switches that off (takes effect at the next gcge_hip_mat_create*)                             */
void gcge_hip_set_offset_patterns (int on);
/*     column-wise Gram-Schmidt over the slots (the reference's OrthSelf, src/ops_orth.c:45-118): the scaling of x_k is held
 *     back and folded, with the k x 1 Gram of the NEXT column, into the rank-1 update that follows it — one sweep per column
 *     instead of three, same operands and products (1 default, 0: every slot call launches its own kernel)               */
void gcge_hip_set_mgs_fusion (int on);
void gcge_hip_mgs_fusion_stats (long *fused_steps, long *served_grams);
/*     host-only structural self-check of that upload (no device needed): expands the tiles back into (row, column,
 *     value) triples and compares with the CSR arrays bit for bit; returns the number of differences (0 = identical),
 *     the X rows staged per matrix row, the ELL entries per non-zero and the grid strides it detected (0: none)       */
long gcge_hip_tile_selfcheck (int nrows, int ncols_local, const int *rowptr, const int *colidx, const double *val,
		double *xrows_per_row, double *ell_per_nnz, long *strides);
/*     pattern path (csrc/hip/spmm_pattern.hip): matrices whose rows repeat a few stencils {(col - row, value)} are
 *     additionally kept as 16-bit pattern ids + a table of npat * lt {double value; long offset} entries (span, span2 =
 *     longest and second longest |offset| of the interior stencil: launch geometry only); d_dots may be NULL.  gcge_hip_mat_patterns() tells whether a matrix qualified (0: served by the generic kernels).       */
int gcge_hip_pattern_spmm (int nrows, const unsigned short *d_pid, const void *d_tab, int npat, int lt, long span, long span2,
		const double *d_x, long ldx, double *d_y, long ldy, int ncols, double *d_dots, double *d_dots_yy, void *stream);
/*     the two passes of a block-CG iteration on a pattern matrix (reference recurrences: src/ops_lin_sol.c:296-380;
 *     here w = A p is formed twice and never stored).  mode 2: d_dots[j] = sum_r x[r,j] (A x)[r,j], d_dots_yy[j] =
 *     sum_r (A x)[r,j]^2, nothing written.  mode 3: r -= (A x) diag(alpha), pnew = r diag(cr) + x diag(cb),
 *     d_dots[j] = sum_r cr_j r[r,j]^2, with (alpha, cb, cr)_j = flag_j ? (alpha_j, beta_j, 1) : (0, 1, 0); pnew != x.
 *     -1: operands do not qualify (the caller keeps the stored-w form)                                              */
int gcge_hip_pattern_cg (int mode, int nrows, const unsigned short *d_pid, const void *d_tab, int npat, int lt,
		long span, long span2, const double *d_x, long ldx, double *d_r, long ldr, double *d_pnew, long ldp, int ncols,
		const double *d_alpha, const double *d_beta, const int *d_flag, double *d_dots, double *d_dots_yy, void *stream,
		const double *d_b, long ldb);
/*     mode 4: d_dots[j] = sum_r ((A x)[r,j] - alpha_j x[r,j])^2 (residual norms of Ritz pairs);  mode 5: r = b - A x,
 *     pnew = r, d_dots[j] = sum_r r[r,j]^2 (start of the CG; d_b / ldb only used here)                              */
/*     near > 0: the caller knows the table's slots are [-S, 0, +S, -L, +L, -1, +1] (7-point stencil) and passes the
 *     largest |offset| in the table; the passes that
 *     store nothing (modes 2, 4) then run the LDS-ring sweep of csrc/hip/spmm_ring.hip (X rows several grid planes
 *     ahead by LDS-DMA); gcge_hip_spmm_ring_tune(on, planes_ahead) switches it off / picks 2 or 3 planes            */
int gcge_hip_pattern_cg_near (int mode, int nrows, const unsigned short *d_pid, const void *d_tab, int npat, int lt,
		long span, long span2, const double *d_x, long ldx, double *d_r, long ldr, double *d_pnew, long ldp, int ncols,
		const double *d_alpha, const double *d_beta, const int *d_flag, double *d_dots, double *d_dots_yy, void *stream,
		const double *d_b, long ldb, long near);
int gcge_hip_pattern_spmm_near (int nrows, const unsigned short *d_pid, const void *d_tab, int npat, int lt, long span, long span2,
		const double *d_x, long ldx, double *d_y, long ldy, int ncols, double *d_dots, double *d_dots_yy, void *stream, long near);
void gcge_hip_spmm_ring_tune (int on, int planes_ahead);
void gcge_hip_spmm_ring_product (int on);    /* 1: Y = A X through the ring as well (default 0: measured no faster, the product is bound by its read + write traffic) */
void gcge_hip_spmm_ring_wide (int on);      /* 1: 64-bit lane addresses even where 32-bit lane offsets would do (tests) */
void gcge_hip_spmm_ring_xcd (int on);       /* 1: contiguous tile runs per XCD (fabric reads 12.0 -> 9.9 GB per 64 columns at 256^3, same time) */
long gcge_hip_spmm_ring_launches (void);   /* 16-column launches the ring sweep has taken so far */
/*     the same on operator-table objects (halo rows of p fetched by pass 1 and reused by pass 2; sums are the LOCAL
 *     parts, on the host); gcge_hip_cg_fusable: 1 if (mat, p, ncols) qualify                                        */
int gcge_hip_cg_fusable (void *mat, void **p, int ncols);
int gcge_hip_cg_recompute_pays (void *mat);   /* 1: chain + line-exchange layout (HBM-bound product) */
/*     the GCGE_RESIDUAL_FN (include/gcge_ops.h) OPS_HIP_Set registers: squared residual norms of Ritz pairs of a
 *     standard problem in one read of x (kernel MODE 4); returned as void* for test harnesses                    */
void *gcge_hip_residual_hook (void);
int gcge_hip_cg_pass1_mv (void *mat, void **p, int c0, int m, double *host_pw, double *host_ww);
int gcge_hip_cg_start_mv (void *mat, void **x, int xc0, void **b, int bc0, void **r, void **p0, int rc0, int m,
		double *host_rho);   /* r = b - A x, p0 = r, rho = column sums of r^2 (local rows) in one sweep */
int gcge_hip_cg_pass2_mv (void *mat, void **p, void **r, void **pnew, int c0, int m, const double *d_alpha,
		const double *d_beta, const int *d_flag, double *host_rho);
/*     the same two passes with the column sums left on the DEVICE and nothing waited for (the fused CG computes its
 *     scalars there): pass 1: d_out[0,m) = p.(A p), d_out[m,2m) = |A p|^2 (d_out: >= 6 m doubles); pass 2: d_rho[0,m) */
int gcge_hip_cg_pass1_dev (void *mat, void **p, int c0, int m, double *d_out);
/*     second pass without a stored residual (kernel MODE 7): r_k = p_k - beta_{k-1} p_{k-1} rebuilt from the previous
 *     direction pprev (d_betaprev: zeros in the first iteration, where pprev may be p); reads p, pprev, writes pnew */
int gcge_hip_cg_pass2i_dev (void *mat, void **p, void **pprev, void **pnew, int c0, int m, const double *d_alpha,
		const double *d_beta, const int *d_flag, const double *d_betaprev, double *d_rho);
int gcge_hip_cg_pass2i_mv (void *mat, void **p, void **pprev, void **pnew, int c0, int m, const double *d_alpha,
		const double *d_beta, const int *d_flag, const double *d_betaprev, double *host_rho);
long gcge_hip_bpcg_implicit_r_iters (void);   /* CG iterations that ran without a stored residual */
int gcge_hip_cg_pass2_dev (void *mat, void **p, void **r, void **pnew, int c0, int m, const double *d_alpha,
		const double *d_beta, const int *d_flag, double *d_rho);
int gcge_hip_pattern_width (int max_row_len);
int gcge_hip_mat_patterns (const GCGE_HIP_MAT *A);
const char *gcge_hip_mat_spmm_form (const GCGE_HIP_MAT *A);   /* name of the K1 kernel family MatDotMultiVec takes for this matrix */
int gcge_hip_mat_pattern_chain (const GCGE_HIP_MAT *A);   /* 0 none, 1 chain layout (span2 == -1), 2 chain + line exchange (span2 == -L) */
/*     d_out[j] = sum_r x[r,j] y[r,j] */
int gcge_hip_coldots (int nrows, const double *d_x, long ldx, const double *d_y, long ldy, int m,
		double *d_out, void *stream);
/*     d_out[j] = x_j . y_j, d_out[m + j] = y_j . y_j in one sweep (bitwise what two gcge_hip_coldots calls return) */
int gcge_hip_coldots2 (int nrows, const double *d_x, long ldx, const double *d_y, long ldy, int m,
		double *d_out, void *stream);
/*     in-solve rate of K2 / K3 per shape: bracket every Gram and panel update the slots launch with HIP events (no
 *     synchronisation added); the report lists, per (kernel, k, m), calls, average time and 2 n k m flop / time            */
void gcge_hip_dense_profile (int on);
int  gcge_hip_dense_profile_report (char *buf, int len);
/*     the same as numbers: rows of 7 doubles (0 Gram / 1 panel update, k, m, calls, ms in all, flop in all, bytes in all — what each
 *     launch had to move: a panel with beta == NULL or updated in place is not read), largest total time first, at most max_rows;
 *     returns the number of shapes seen (bench.py: roofline_gram / roofline_panel_update)                                       */
int  gcge_hip_dense_profile_shapes (double *out, int max_rows);
/* K3  Y[:,0:m) = X[:,0:k) C + Y diag(beta);  d_c row-major k x m; d_beta NULL => overwrite */
int gcge_hip_lincomb (int nrows, const double *d_x, long ldx, int k, const double *d_c, int m,
		const double *d_beta, double *d_y, long ldy, void *stream);
void gcge_hip_lincomb_tune (int row_fragments);   /* 0 automatic (2 for m > 64 on large blocks), 1, 2: 16-row fragments per wave */
/* K4  Y[:,0:m) = alpha X[:,0:m) + beta Y   (d_x NULL: scale only; beta == 0: no read of Y) */
int gcge_hip_axpby (int nrows, double alpha, const double *d_x, long ldx, double beta,
		double *d_y, long ldy, int m, void *stream);

#ifdef __cplusplus
}
#endif
#endif
