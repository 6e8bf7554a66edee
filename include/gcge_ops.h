/* gcge_ops.h — the operator table ("plugin ABI") of the GCG hot path.
 *
 * This is OUR declaration of the function-pointer table that GCGE's solver
 * layers are written against.  It must stay bit-identical in layout (member
 * order, types, the five opaque workspace pointers and the two nested table
 * pointers) to the reference's `struct OPS_`  (/root/reference/src/ops.h:43-152)
 * so that
 *   - a back-end written against this header (OPS_HIP_Set) can be handed to the
 *     reference's unmodified solver code (GCG, ModifiedGramSchmidt, BlockPCG), and
 *   - our solver stack can be driven by a back-end compiled against the
 *     reference header (e.g. app_ccs in oracle/_ref).
 * tests/test_abi.py checks sizeof/offsetof of every member against the reference
 * header when /root/reference is present.
 *
 * Argument conventions (reference: src/ops.h:78-103):
 *   - a "multivector" is an opaque handle `void **`; only the back-end that
 *     created it may look inside;
 *   - column ranges are half open: start[0]:end[0] selects columns of the FIRST
 *     multivector argument, start[1]:end[1] of the SECOND;
 *   - small dense results (inner products, Q^T A P, coefficients) live in HOST
 *     memory, column-major, and are complete when the call returns.
 */
#ifndef GCGE_OPS_H
#define GCGE_OPS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OPS_ {
	/* ---- services -------------------------------------------------------- */
	void   (*Printf) (const char *fmt, ...);
	double (*GetWtime) (void);
	int    (*GetOptionFromCommandLine) (const char *name, char type, void *data,
			int argc, char *argv[], struct OPS_ *ops);
	/* ---- sparse matrix --------------------------------------------------- */
	void (*MatView)  (void *mat, struct OPS_ *ops);
	void (*MatAxpby) (double alpha, void *matX, double beta, void *matY, struct OPS_ *ops);
	/* ---- single vector (unused when the multivector slots are set) -------- */
	void (*VecCreateByMat)    (void **des_vec, void *src_mat, struct OPS_ *ops);
	void (*VecCreateByVec)    (void **des_vec, void *src_vec, struct OPS_ *ops);
	void (*VecDestroy)        (void **des_vec, struct OPS_ *ops);
	void (*VecView)           (void *x, struct OPS_ *ops);
	void (*VecInnerProd)      (void *x, void *y, double *inner_prod, struct OPS_ *ops);
	void (*VecLocalInnerProd) (void *x, void *y, double *inner_prod, struct OPS_ *ops);
	void (*VecSetRandomValue) (void *x, struct OPS_ *ops);
	void (*VecAxpby)          (double alpha, void *x, double beta, void *y, struct OPS_ *ops);
	void (*MatDotVec)         (void *mat, void *x, void *y, struct OPS_ *ops);
	void (*MatTransDotVec)    (void *mat, void *x, void *y, struct OPS_ *ops);
	/* ---- block of vectors ------------------------------------------------- */
	void (*MultiVecCreateByMat)      (void ***multi_vec, int num_vec, void *src_mat, struct OPS_ *ops);
	void (*MultiVecCreateByVec)      (void ***multi_vec, int num_vec, void *src_vec, struct OPS_ *ops);
	void (*MultiVecCreateByMultiVec) (void ***multi_vec, int num_vec, void **src_mv, struct OPS_ *ops);
	void (*MultiVecDestroy)          (void ***multi_vec, int num_vec, struct OPS_ *ops);
	void (*GetVecFromMultiVec)       (void **multi_vec, int col, void **vec, struct OPS_ *ops);
	void (*RestoreVecForMultiVec)    (void **multi_vec, int col, void **vec, struct OPS_ *ops);
	void (*MultiVecView)             (void **x, int start, int end, struct OPS_ *ops);
	/* inner_prod(k x m) = x[:,s0:e0)^T y[:,s1:e1); nsdIP: 'N' full, 'S' symmetric, 'D' diagonal */
	void (*MultiVecLocalInnerProd)   (char nsdIP, void **x, void **y, int is_vec,
			int *start, int *end, double *inner_prod, int ldIP, struct OPS_ *ops);
	void (*MultiVecInnerProd)        (char nsdIP, void **x, void **y, int is_vec,
			int *start, int *end, double *inner_prod, int ldIP, struct OPS_ *ops);
	void (*MultiVecSetRandomValue)   (void **multi_vec, int start, int end, struct OPS_ *ops);
	/* y = alpha x + beta y on column ranges (x == NULL: scale only) */
	void (*MultiVecAxpby)            (double alpha, void **x, double beta, void **y,
			int *start, int *end, struct OPS_ *ops);
	/* y = x coef + y diag(beta) */
	void (*MultiVecLinearComb)       (void **x, void **y, int is_vec, int *start, int *end,
			double *coef, int ldc, double *beta, int incb, struct OPS_ *ops);
	void (*MatDotMultiVec)           (void *mat, void **x, void **y, int *start, int *end, struct OPS_ *ops);
	void (*MatTransDotMultiVec)      (void *mat, void **x, void **y, int *start, int *end, struct OPS_ *ops);
	/* qAp = Q^T A P ; ntsdQAP: 'N','S','D' or 'T' (store the transpose) */
	void (*MultiVecQtAP)             (char ntsA, char ntsdQAP, void **mvQ, void *matA, void **mvP,
			int is_vec, int *start, int *end, double *qAp, int ldQAP, void **mv_ws, struct OPS_ *ops);
	/* ---- small dense (host) ----------------------------------------------- */
	struct OPS_ *lapack_ops;
	void (*DenseMatQtAP) (char ntluA, char nsdC, int nrowsA, int ncolsA, int nrowsC, int ncolsC,
			double alpha, double *matQ, int ldQ, double *matA, int ldA, double *matP, int ldP,
			double beta, double *matC, int ldC, double *dbl_ws);
	void (*DenseMatOrth) (double *mat, int nrows, int ldm, int start, int *end,
			double orth_zero_tol, double *dbl_ws, int length, int *int_ws);
	/* ---- linear solvers ---------------------------------------------------- */
	void (*LinearSolver)      (void *mat, void *b, void *x, struct OPS_ *ops);
	void *linear_solver_workspace;
	void (*MultiLinearSolver) (void *mat, void **b, void **x, int *start, int *end, struct OPS_ *ops);
	void *multi_linear_solver_workspace;
	/* ---- block orthonormalisation ------------------------------------------ */
	void (*MultiVecOrth) (void **x, int start_x, int *end_x, void *B, struct OPS_ *ops);
	void *orth_workspace;
	/* ---- multigrid: hierarchy from the back-end, transfers through P_array (used by BlockAMG) ------ */
	void (*MultiGridCreate)  (void ***A_array, void ***B_array, void ***P_array,
			int *num_levels, void *A, void *B, struct OPS_ *ops);
	void (*MultiGridDestroy) (void ***A_array, void ***B_array, void ***P_array,
			int *num_levels, struct OPS_ *ops);
	void (*VecFromItoJ)      (void **P_array, int level_i, int level_j,
			void *vec_i, void *vec_j, void **vec_ws, struct OPS_ *ops);
	void (*MultiVecFromItoJ) (void **P_array, int level_i, int level_j,
			void **multi_vec_i, void **multi_vec_j, int *startIJ, int *endIJ,
			void ***multi_vec_ws, struct OPS_ *ops);
	/* ---- eigensolver ------------------------------------------------------- */
	void (*EigenSolver) (void *A, void *B, double *eval, void **evec,
			int nevGiven, int *nevConv, struct OPS_ *ops);
	void *eigen_solver_workspace;
	/* ---- composite back-ends (PAS) ----------------------------------------- */
	struct OPS_ *app_ops;
} OPS;

/* life cycle (reference: src/ops.c:26-149) */
void OPS_Create  (OPS **ops);   /* all slots NULL                               */
void OPS_Setup   (OPS  *ops);   /* back-fill NULL slots with the defaults below */
void OPS_Destroy (OPS **ops);

/* defaults installed by OPS_Setup (reference: src/ops_multi_vec.c) */
void   DefaultPrintf (const char *fmt, ...);
double DefaultGetWtime (void);
int    DefaultGetOptionFromCommandLine (const char *name, char type, void *value,
		int argc, char *argv[], struct OPS_ *ops);
void   DefaultMultiVecInnerProd (char nsdIP, void **x, void **y, int is_vec,
		int *start, int *end, double *inner_prod, int ldIP, struct OPS_ *ops);
void   DefaultMultiVecQtAP (char ntsA, char ntsdQAP, void **mvQ, void *matA, void **mvP,
		int is_vec, int *startQP, int *endQP, double *qAp, int ldQAP,
		void **mv_ws, struct OPS_ *ops);

/* multigrid transfers through P_array (reference: src/ops_multi_grid.c:20-117; installed by OPS_Setup, src/ops.c:107-112) */
void   DefaultVecFromItoJ (void **P_array, int level_i, int level_j, void *vec_i, void *vec_j, void **vec_ws, struct OPS_ *ops);
void   DefaultMultiVecFromItoJ (void **P_array, int level_i, int level_j, void **multi_vec_i, void **multi_vec_j,
		int *startIJ, int *endIJ, void ***multi_vec_ws, struct OPS_ *ops);

/* Host dense back-end: column-major blocks in host memory.  Layout-compatible
 * with the reference's LAPACKVEC/LAPACKMAT (app/app_lapack.h:17-20).          */
typedef struct GCGE_DENSE_ {
	double *data; int nrows; int ncols; int ldd;
} GCGE_DENSE;
void OPS_DENSE_Set (struct OPS_ *ops);   /* counterpart of OPS_LAPACK_Set (app_lapack.c) */

/* Communicator hook for row-partitioned back-ends (one process per GPU).
 * The reference reduces partial Gram matrices with MPI_Allreduce
 * (src/ops_multi_vec.c:206-228, src/ops_lin_sol.c:317,365); here the reduction
 * is a callback so the same host code runs over RCCL, gloo or nothing.       */
typedef struct GCGE_COMM_ {
	int rank, size;
	/* in-place sum over ranks of n contiguous doubles in HOST memory */
	void (*allreduce_sum) (double *buf, int n, void *ctx);
	void *ctx;
} GCGE_COMM;
void       GCGE_SetComm (const GCGE_COMM *comm);   /* NULL: single rank */
GCGE_COMM *GCGE_GetComm (void);
/* Opt-in: the back-end's MultiVecLocalInnerProd slot returns the sum over the ranks too.  For solver stacks that reduce their
 * "local" products through MPI only — the reference's BlockPCG (src/ops_lin_sol.c:306-321,355-369: MultiVecLocalInnerProd, then
 * MPI_Allreduce under OPS_USE_MPI) — so that a NON-MPI build of the reference spans the ranks with flag 0 as well.  Honoured by
 * OPS_HIP_Set's slot; libgcge_host.so's own BlockPCG skips its reduction while it is on.  Get: 1 only while a communicator exists. */
void       GCGE_SetLocalInnerProdReduces (int on);
int        GCGE_GetLocalInnerProdReduces (void);
/* Shift of the W systems for a user-defined MultiLinearSolver (flag 1): the reference calls such a solver with A
 * only (ops_eig_sol_gcg.c:584-618) and leaves sigma to it; our GCG publishes (sigma, B) here before every call so
 * that a shift-aware solver (the fused block CG of the HIP back-end) can apply A + sigma B.  An application that
 * drives the REFERENCE's GCG sets a fixed shift itself.  sigma == 0: no shift. */
void       GCGE_SetLinearSolverShift (double sigma, void *matB);
void       GCGE_GetLinearSolverShift (double *sigma, void **matB);
/*     the column scales of the "user" tolerance type of BlockPCG (src/ops_lin_sol.c:186-192: a column has converged when
 *     its residual is below tol * |scale_j|; the reference's GCG leaves lambda_j + sigma at the start of BlockPCG's scalar
 *     scratch).  A solver installed behind flag 1 has no such scratch: the GCG driver publishes the n scales here for the
 *     duration of the call, NULL otherwise.                                                                              */
void       GCGE_SetLinearSolverUserScale (const double *scale, int n);
const double *GCGE_GetLinearSolverUserScale (int *n);
/* Optional fast path of CheckConvergence (src/ops_eig_sol_gcg.c:195-315 forms A x, B x, lambda B x, the difference and
 * its column norms through five slots = 11 block streams).  A back-end may offer the squared residual norms
 *   res_sq[j] = sum over its LOCAL rows of ((A x_j) - lambda_j (B x_j))^2 ,  j = start .. end-1  (columns of x)
 * in one go; it returns 1 if it did, 0 to decline (the driver then takes the slots).  `owner` ties the hook to one
 * operator table: it is only used when ops->MatDotMultiVec == owner.  The driver sums over ranks (GCGE_COMM). */
typedef int (*GCGE_RESIDUAL_FN) (void *A, void *B, void **x, int start, int end, const double *lambda, double *res_sq);
void       GCGE_SetResidualHook (GCGE_RESIDUAL_FN fn, void *owner);
GCGE_RESIDUAL_FN GCGE_GetResidualHook (void *owner);
/* Optional capability of a user-defined MultiLinearSolver: the driver's systems A w = (lambda + sigma) B x are started
 * from w = x, so for B == NULL the right-hand side is the initial guess scaled column by column.  A solver that
 * declared the capability (owner = its function pointer, as installed in ops->MultiLinearSolver) is called with the x
 * block holding the initial guess, GCGE_GetLinearSolverRhsScale() returning the factors and the b block NOT filled in
 * (it may use that block as scratch).  The reference forms b through MatDotMultiVec + MultiVecLinearComb
 * (src/ops_eig_sol_gcg.c:560-577): two block sweeps and a third read at the start of the solve. */
void       GCGE_SetRhsScaleCapability (void *owner);
void       GCGE_SetRhsScaleCapabilityOfBlockAMG (void *owner);   /* second owner: BlockAMG over a back-end that forms b in one sweep (NULL: none) */
int        GCGE_HasRhsScaleCapability (void *owner);
void       GCGE_SetLinearSolverRhsScale (const double *scale);   /* NULL: b is an ordinary right-hand side */
const double *GCGE_GetLinearSolverRhsScale (void);
/* Blocks of vectors the driver does not need while MultiLinearSolver runs (its orthonormalisation / residual work
 * blocks: contents dead across the call).  A solver may use them as scratch — the fused HIP solver takes those that
 * match its own work blocks as additional slots of its direction ring, which is what limits it when HBM is nearly
 * full (BASELINE config 4's shape: 244 of 288 GB are the solver stack's own blocks).  Valid only during the call. */
void       GCGE_SetLinearSolverIdleBlocks (void ***blocks, int count);
void    ***GCGE_GetLinearSolverIdleBlocks (int *count);
/* Optional capability: a back-end whose MultiVecLinearComb works ROW BY ROW (row-major blocks: every output row is
 * formed from the same row of x and written after that row has been read) may declare panel updates IN PLACE safe:
 * y == x with the output columns inside the input column range, at most `max_cols` output columns per call.  The
 * block orthonormalisation and ComputeP of this solver stack then skip the work block + copy back the reference's
 * LAPACKVEC layout needs (src/ops_orth.c, src/ops_eig_sol_gcg.c:624-640): same arithmetic, same results, two block
 * streams less per update.  `owner` = the table's MultiVecLinearComb; 0 columns: not declared. */
void       GCGE_SetInplaceLinearComb (void *owner, int max_cols);
int        GCGE_InplaceLinearCombCols (void *owner);
void       GCGE_SetQuiet (OPS *ops, int quiet);   /* silence ops->Printf (and the dense table's) */

#ifdef __cplusplus
}
#endif
#endif /* GCGE_OPS_H */
