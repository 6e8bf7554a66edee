/* Synthetic symmetric test matrices for the GCG hot path (host CSR arrays).
 *
 * These are the deterministic generators SURVEY.md §8(d) names; for a symmetric
 * matrix the three arrays are at the same time the reference's CCS triple
 * (app/app_ccs.h:20-24: data / i_row / j_col), so one generator feeds the
 * reference build under oracle/_ref, the CPU oracle and the HIP back-end.
 *
 * Every generator can emit a contiguous slab of rows [row_begin,row_end) with
 * GLOBAL column indices — the row partition used by the multi-GPU path.
 */
#ifndef GCGE_PROBLEMS_H
#define GCGE_PROBLEMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct GCGE_CSR_ {
	int     nrows;      /* rows held (= row_end-row_begin)            */
	int     ncols;      /* global dimension                           */
	int     row_begin;  /* first global row of this slab              */
	int64_t nnz;
	int    *rowptr;     /* nrows+1, 0-based, local to this slab       */
	int    *colidx;     /* nnz, global, ascending inside a row        */
	double *val;        /* nnz                                        */
} GCGE_CSR;

void gcge_csr_free(GCGE_CSR *A);

/* 3-D 7-point Laplacian on an N^3 grid, index c = i + N (j + N k):
 * diagonal 6, six -1 neighbours, Dirichlet truncation.  B = NULL problem.
 * eigenvalues 6 - 2cos(i pi/(N+1)) - 2cos(j pi/(N+1)) - 2cos(k pi/(N+1)).   */
int gcge_problem_lap3d(int N, int64_t row_begin, int64_t row_end, GCGE_CSR *A);
/* same stencil on an Nx x Ny x Nz box (weak scaling: Nz grows with the number of slabs) */
int gcge_problem_lap3d_box(int Nx, int Ny, int Nz, int64_t row_begin, int64_t row_end, GCGE_CSR *A);

/* The reference's stock pair (test/test_app_ccs.c:142-184): 1-D linear FE,
 * A = tridiag(-1,2,-1)/h, B = h I, h = 1/(n+1).                             */
int gcge_problem_fe1d(int n, GCGE_CSR *A, GCGE_CSR *B);

/* P1 stiffness/mass pair on the Kuhn (6-tet) triangulation of a uniform cube,
 * M^3 interior nodes, all-Dirichlet, h = 1/(M+1)  (SURVEY.md §8d):
 *   A = h   [6; -1 at +-e_x,+-e_y,+-e_z]
 *   B = h^3 [0.4; 1/20 at +-e_x,+-e_y,+-e_z,+-(1,1,1); 1/30 at +-(1,1,0),+-(1,0,1),+-(0,1,1)] */
int gcge_problem_fe3d(int M, int64_t row_begin, int64_t row_end, GCGE_CSR *A, GCGE_CSR *B);

/* SiO2-like irregular SPD matrix on a G^3 grid: 12th-order central-difference
 * -Laplacian (37-point) plus K Gaussian "atoms" u u^T with heavy-tailed radius
 * R = R0 + R1 u1 u2 (grid cells) and weight in [0.5,1.5]; rows differ in
 * length by more than an order of magnitude (load-imbalance stress).        */
int gcge_problem_sio2_like(int G, int K, double R0, double R1, uint64_t seed,
		int64_t row_begin, int64_t row_end, GCGE_CSR *A);

/* The same operator on the BALL inscribed in the G^3 box, rows = the grid points inside in scan order (x fastest) — the domain
 * and numbering of the PARSEC real-space matrices of the reference's test/submit.sh:9-15 (SiO2, Ga41As41H72, ...): the
 * principal submatrix of the box matrix (stencil arms and atom blocks cut at the sphere).  box_of_row (may be NULL; free with
 * gcge_free_ints): the box index x + G (y + G z) of every row of the slab — the geometry gcge_hip_mat_create_grid takes. */
int64_t gcge_problem_sio2_ball_rows(int G);
int gcge_problem_sio2_ball(int G, int K, double R0, double R1, uint64_t seed,
		int64_t row_begin, int64_t row_end, GCGE_CSR *A, int **box_of_row);

/* ---- matrix ingestion (the formats the reference's users hold) --------------------------
 * PETSc binary Mat file (what test/test_app_slepc.c:416-445 loads with MatLoad: SiO2, Ga41As41H72, ... of
 * submit.sh:9-15): big-endian int32 header {1211216, rows, cols, nnz}, int32 row lengths, int32 column
 * indices, float64 values (AIJ, 32-bit indices).  Rows [row_begin,row_end) only (row_end < 0: all), global
 * columns.  Returns 0, -1 cannot open / short file, -2 not a (32-bit index, real) Mat file, -3 out of memory. */
int gcge_load_petsc_binary(const char *path, int64_t row_begin, int64_t row_end, GCGE_CSR *A);
/* writer of the same format (tests, data exchange with a PETSc build) */
int gcge_save_petsc_binary(const char *path, const GCGE_CSR *A);
/* Matrix Market coordinate file (what the SuiteSparse collection ships SiO2, Ga41As41H72, ... of test/submit.sh:9-15 in; the
 * reference's users convert them to PETSc binary for test_app_slepc.c:416-445): general / symmetric / skew-symmetric,
 * real / integer / pattern, any entry order, duplicates summed -> the full matrix in CSR, ascending columns.
 * Returns 0, -1 cannot open / short file, -2 not a real coordinate file, -3 out of memory. */
int gcge_load_matrix_market(const char *path, GCGE_CSR *A);
/* writer (tests, data exchange): symmetric != 0 writes the lower triangle as "real symmetric" */
int gcge_save_matrix_market(const char *path, const GCGE_CSR *A, int symmetric);
/* Compressed-column triple (app/app_ccs.h:20-24 CCSMAT; MATLAB's jc/ir/pr of app/app_matlab.c:80-98) of a
 * general square or rectangular matrix -> CSR (transposition by counting; a symmetric matrix comes out
 * identical).  one_based != 0: MATLAB-style 1-based indices. */
int gcge_csr_from_ccs(int nrows, int ncols, const int *j_col, const int *i_row, const double *data,
		int one_based, GCGE_CSR *A);

/* ---- row partition helpers (one process per GPU) ---------------------
 * A slab holds rows [row_begin,row_begin+nrows) with GLOBAL column indices.
 * gcge_dist_ghosts   lists (ascending, unique) the global columns the slab references outside
 *                    its own row range — the halo rows of X an SpMM needs from other ranks;
 * gcge_dist_localize rewrites colidx in place to slab-local numbering: owned column g ->
 *                    g-row_begin, ghost ghosts[i] -> nrows+i (ncols becomes nrows+nghost).  */
int gcge_dist_ghosts (const GCGE_CSR *A, int **ghosts_out, int *nghost_out);
int gcge_dist_localize (GCGE_CSR *A, const int *ghosts, int nghost);
void gcge_free_ints (int *p);
/* gcge_dist_plan_halo: the halo plan of this rank's slab — collective over `t`.  part: size + 1 row offsets; ghosts: my halo rows
 * (ascending global ids, gcge_dist_ghosts).  Out: recv_cnt[q] rows I receive from rank q (in ghost order), send_cnt[q] rows I ship
 * to rank q, send_rows = MY local rows to ship, grouped by destination rank, ascending inside a group (free with gcge_free_ints).
 * The transport moves ints once, at set-up: allgather_int(send, n, recv_all): every rank contributes n ints, recv_all[r * n + i] =
 * rank r's i-th; exchange_int(sendbuf, send_cnt, recvbuf, recv_cnt): the ints for rank q are the q-th group of sendbuf (send_cnt[q]
 * of them), those from rank q land in the q-th group of recvbuf (recv_cnt[q]).  Returns 0; < 0: inconsistent inputs / a peer
 * asked for a row this rank does not own.  ONE planner for every transport: RCCL inside the HIP back-end
 * (gcge_hip_mat_create_slab, csrc/hip/rccl_comm.hip) and torch.distributed in the tests (gcge_amd/dist.py) run this function. */
typedef struct GCGE_PLAN_TRANSPORT_ {
	int rank, size;
	void (*allgather_int) (const int *send, int n, int *recv_all, void *ctx);
	void (*exchange_int) (const int *sendbuf, const int *send_cnt, int *recvbuf, const int *recv_cnt, void *ctx);
	void *ctx;
} GCGE_PLAN_TRANSPORT;
int gcge_dist_plan_halo (const long *part, const int *ghosts, int nghost, const GCGE_PLAN_TRANSPORT *t,
		int *recv_cnt, int *send_cnt, int **send_rows_out, int *nsend_out);

/* Reproducible U[0,1) stream shared by C and the python tests:
 * splitmix64(seed + index) >> 11 scaled by 2^-53.                           */
double gcge_uniform(uint64_t seed, uint64_t index);

#ifdef __cplusplus
}
#endif
#endif
