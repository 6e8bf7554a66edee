/* gcge_solver.h — host-side solver stack of the GCG hot path, written against
 * the operator table of gcge_ops.h only (never looks inside a multivector).
 *
 * Public names, argument order and defaults mirror the reference so that its
 * harness and drivers read the same:
 *   block orthonormalisation   src/ops_orth.h:18-41
 *   block PCG                  src/ops_lin_sol.h:29-45
 *   GCG eigensolver            src/ops_eig_sol_gcg.h:21-83
 *   harness                    test/test_eig_sol_gcg.c:28
 */
#ifndef GCGE_SOLVER_H
#define GCGE_SOLVER_H

#include "gcge_ops.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- small dense symmetric eigensolver (replaces dsyevx / dsyev) ---------- */
int GCGE_SymEig (char uplo, int n, const double *a, int lda, double *w,
		double *z, int ldz, double *work /* >= 2n */);
/*     the host implementation itself (GCGE_SymEig is this; kept under both names) */
int GCGE_SymEigHost (char uplo, int n, const double *a, int lda, double *w,
		double *z, int ldz, double *work /* >= 2n */);
/*     a back-end's own solver with the same contract (host pointers in and out, 0 on success), registered for ONE
 *     operator table: owner = that table's MultiVecLinearComb slot.  GCGE_SymEigFor(owner, ...) — what the GCG driver and
 *     the orthonormalisation call with their table's slot — takes it for n >= min_n when the owners match, the host solver
 *     otherwise.  The HIP back-end registers gcge_hip_symeig (csrc/hip/eig_device.hip) in OPS_HIP_Set.                  */
typedef int (*GCGE_SYMEIG_FN) (char uplo, int n, const double *a, int lda, double *w, double *z, int ldz);
void GCGE_SetSymEigHook (GCGE_SYMEIG_FN fn, int min_n, void *owner);
int GCGE_SymEigFor (void *owner, char uplo, int n, const double *a, int lda, double *w,
		double *z, int ldz, double *work /* >= 2n */);

/* ---- block orthonormalisation (sets ops->MultiVecOrth + orth_workspace) ---- */
typedef struct ModifiedGramSchmidtOrth_ {
	int    block_size;     /* columns orthonormalised among themselves per sweep (<=0: half) */
	int    max_reorth;
	double orth_zero_tol;  /* a column with B-norm below this is dropped                      */
	double reorth_tol;     /* stop re-orthogonalising when max |coef| falls below this        */
	void   **mv_ws;        /* >= block columns of scratch (holds B x)                         */
	double *dbl_ws;
} ModifiedGramSchmidtOrth;
typedef ModifiedGramSchmidtOrth BinaryGramSchmidtOrth;

void MultiVecOrthSetup_ModifiedGramSchmidt (int block_size, int max_reorth,
		double orth_zero_tol, void **mv_ws, double *dbl_ws, struct OPS_ *ops);
void MultiVecOrthSetup_BinaryGramSchmidt (int block_size, int max_reorth,
		double orth_zero_tol, void **mv_ws, double *dbl_ws, struct OPS_ *ops);
/* block Cholesky-QR variant of the MGS scheme (method name "chol"): the same contract with
 * BLOCK operations only — what a GPU back-end wants (block_size <= columns of mv_ws).      */
void MultiVecOrthSetup_CholeskyQR (int block_size, int max_reorth,
		double orth_zero_tol, void **mv_ws, double *dbl_ws, struct OPS_ *ops);

/* ---- block conjugate gradients (sets ops->MultiLinearSolver) --------------- */
typedef struct BlockPCGSolver_ {
	int max_iter; double rate; double tol; char tol_type[8];   /* "abs" | "rel" | "user" */
	void   **mv_ws[3];    /* r, p, w : one column per right-hand side                      */
	double *dbl_ws;       /* 6 * (number of right-hand sides)                              */
	int    *int_ws;       /* 2 * (number of right-hand sides)                              */
	void   *pc;           /* stored, never applied (as in the reference)                   */
	/* optional replacement for y = A x (the A + sigma B product); z[s:...) is scratch     */
	void  (*MatDotMultiVec) (void **x, void **y, int *start, int *end, void **z, int s, struct OPS_ *ops);
	int niter; double residual;
} BlockPCGSolver;

void MultiLinearSolverSetup_BlockPCG (int max_iter, double rate, double tol,
		const char *tol_type, void **mv_ws[3], double *dbl_ws, int *int_ws, void *pc,
		void (*MatDotMultiVec) (void **x, void **y, int *start, int *end, void **z, int s, struct OPS_ *ops),
		struct OPS_ *ops);

/* ---- V-cycle multigrid with block CG smoothing (sets ops->MultiLinearSolver) ------
 * The reference's BlockAMG (src/ops_lin_sol.h:47-60, src/ops_lin_sol.c:466-715): same struct, same argument order.
 * max_iter[0] = V-cycles at most, max_iter[2l + 1] / [2l + 2] = pre- / post-smoothing CG iterations on level l (the
 * coarsest level only pre-smooths: that is its solve); rate[l], tol[l] the CG's stopping parameters on level l, tol[0]
 * also the stopping residual of the cycles.  A_array / P_array: num_levels matrices and num_levels - 1 prolongations of
 * the back-end (ops->MultiGridCreate).  mv_array_ws[i][l]: 0 right-hand side and 1 solution of level l >= 1, 2 / 3 / 4 the
 * CG's r / p / w on level l (2 also holds the residual and the prolongated correction).                                  */
typedef struct BlockAMGSolver_ {
	int    *max_iter; double *rate; double *tol; char tol_type[8];
	void   **A_array; void **P_array; int num_levels;
	void   ***mv_array_ws[5]; double *dbl_ws; int *int_ws;
	void   *pc;
	int    niter; double residual;
} BlockAMGSolver;
void MultiLinearSolverSetup_BlockAMG (int *max_iter, double *rate, double *tol, const char *tol_type,
		void **A_array, void **P_array, int num_levels, void ***mv_array_ws[5], double *dbl_ws, int *int_ws,
		void *pc, struct OPS_ *ops);
/*     The smoother of BlockAMG.  Default: MultiLinearSolverSetup_BlockPCG on the level's r / p / w blocks, as the reference
 *     (src/ops_lin_sol.c:482-486,626-629).  A back-end may register its own block CG for ITS table (owner = the table's
 *     MatDotMultiVec slot): setup(max_iter, rate, tol, tol_type, ops) installs it in ops->MultiLinearSolver (it brings its own
 *     work blocks), residual(ops) returns what BlockPCGSolver.residual would hold after the call.  The HIP back-end
 *     registers its fused device CG in OPS_HIP_Set; GCGE_AMG_HOST_SMOOTHER=1 keeps the default.                          */
typedef void   (*GCGE_SMOOTHER_SETUP_FN) (int max_iter, double rate, double tol, const char *tol_type, struct OPS_ *ops);
typedef double (*GCGE_SMOOTHER_RESIDUAL_FN) (struct OPS_ *ops);
void GCGE_SetBlockAMGSmoother (GCGE_SMOOTHER_SETUP_FN setup, GCGE_SMOOTHER_RESIDUAL_FN residual, void *owner);
int  GCGE_HasBlockAMGSmoother (struct OPS_ *ops);      /* 1: a smoother is registered for THIS table and not switched off */
/*     Two steps of a V-cycle that a back-end may do in one sweep each for ITS table (owner as above), with the arithmetic
 *     of the slot calls they replace (src/ops_lin_sol.c:596-606, :626-640) — results identical bit for bit:
 *       residual(A, b, b0, x, x0, r, r0, ncols, ops):  r[:, r0..) = b[:, b0..) - A x[:, x0..)   (MatDotMultiVec + MultiVecAxpby)
 *       prolong_add(P, xc, c0, xf, f0, ncols, ops):    xf[:, f0..) += P xc[:, c0..)             (MultiVecFromItoJ + MultiVecAxpby)
 *     each returns 1 when it did the work and 0 to decline (the V-cycle then issues the slot calls).
 *     GCGE_AMG_NO_FUSIONS=1 in the environment keeps the slot calls.
 *       form_rhs(b, b0, x, x0, scale, ncols, ops):  b[:, b0..) = x[:, x0..) diag(scale) — with it BlockAMG takes the GCG driver's
 *     "b = x diag(scale)" systems (GCGE_SetRhsScaleCapability, include/gcge_ops.h) and forms b itself in one sweep (the driver's
 *     MatDotMultiVec(B = NULL) + MultiVecLinearComb are two: reference src/ops_eig_sol_gcg.c:560-577); NULL: b comes formed.     */
typedef int (*GCGE_AMG_RESIDUAL_FN) (void *A, void **b, int b0, void **x, int x0, void **r, int r0, int ncols, struct OPS_ *ops);
typedef int (*GCGE_AMG_PROLONG_ADD_FN) (void *P, void **xc, int c0, void **xf, int f0, int ncols, struct OPS_ *ops);
typedef int (*GCGE_AMG_FORM_RHS_FN) (void **b, int b0, void **x, int x0, const double *scale, int ncols, struct OPS_ *ops);
void GCGE_SetBlockAMGFusions (GCGE_AMG_RESIDUAL_FN residual, GCGE_AMG_PROLONG_ADD_FN prolong_add, void *owner);
void GCGE_SetBlockAMGFormRhs (GCGE_AMG_FORM_RHS_FN form_rhs, void *owner);
/*     BlockAMG as the solver of GCG's W systems, the way the reference's SiO2 driver sets it up under OPS_USE_AMG
 *     (test/test_eig_sol_SiO2_MAT.c:96-128,160-170): hierarchy from ops->MultiGridCreate (at most max_levels), work blocks of
 *     block_size columns per level, max_iter = {cycles, smooth0, smooth0, smooth, smooth, ...} (reference: {1, 5, 5, 4, 4, ...}),
 *     rate = {rate0, 1e-16, ...}, tol = {1e-14, 1e-16, ...}, "abs".  Create once (set-up, like the matrix upload), Install
 *     puts BlockAMG into ops->MultiLinearSolver; run the harness with flag 1.  `-gcge_amg_levels L` makes the harness do all
 *     of that itself.  NULL: the back-end has no MultiGridCreate.                                                           */
typedef struct GCGE_AMG_ {
	void **A_array, **B_array, **P_array; int num_levels, block_size, own_smoother;
	void ***mv_ws[5]; int *max_iter; double *rate, *tol; double *dbl_ws; int *int_ws;
} GCGE_AMG;
GCGE_AMG *GCGE_AMGCreate (void *A, void *B, int max_levels, int block_size, int cycles, int smooth0, int smooth, double rate0,
		struct OPS_ *ops);
void GCGE_AMGInstall (GCGE_AMG *amg, struct OPS_ *ops);
void GCGE_AMGDestroy (GCGE_AMG **amg, struct OPS_ *ops);

/* ---- GCG eigensolver (sets ops->EigenSolver) -------------------------------- */
typedef struct GCGSolver_ {
	void   *A; void *B; double sigma;
	double *eval; void **evec;
	int    nevMax; int multiMax; double gapMin;
	int    nevInit; int nevGiven; int nevConv;
	int    block_size; double tol[2]; int numIterMax;   /* tol = {absolute, relative} */
	int    numIter; int sizeV;
	void   **mv_ws[4]; double *dbl_ws; int *int_ws;
	int    length_dbl_ws;
	int    user_defined_multi_linear_solver;   /* 0: BlockPCG, 1: ops->MultiLinearSolver, 2: both */
	int    check_conv_max_num;
	char   initX_orth_method[8]; int initX_orth_block_size; int initX_orth_max_reorth; double initX_orth_zero_tol;
	char   compP_orth_method[8]; int compP_orth_block_size; int compP_orth_max_reorth; double compP_orth_zero_tol;
	char   compW_orth_method[8]; int compW_orth_block_size; int compW_orth_max_reorth; double compW_orth_zero_tol;
	int    compW_cg_max_iter; double compW_cg_rate; double compW_cg_tol; char compW_cg_tol_type[8];
	int    compW_cg_auto_shift; double compW_cg_shift; int compW_cg_order;
	int    compRR_min_num; double compRR_min_gap; double compRR_tol;
} GCGSolver;

void EigenSolverSetup_GCG (int multiMax, double gapMin, int nevInit, int nevMax,
		int block_size, double tol[2], int numIterMax,
		int user_defined_multi_linear_solver,
		void **mv_ws[4], double *dbl_ws, int *int_ws, struct OPS_ *ops);
void EigenSolverCreateWorkspace_GCG (int nevInit, int nevMax, int block_size, void *mat,
		void ***mv_ws, double **dbl_ws, int **int_ws, struct OPS_ *ops);
void EigenSolverDestroyWorkspace_GCG (int nevInit, int nevMax, int block_size, void *mat,
		void ***mv_ws, double **dbl_ws, int **int_ws, struct OPS_ *ops);
void EigenSolverSetParameters_GCG (int check_conv_max_num,
		const char *initX_orth_method, int initX_orth_block_size, int initX_orth_max_reorth, double initX_orth_zero_tol,
		const char *compP_orth_method, int compP_orth_block_size, int compP_orth_max_reorth, double compP_orth_zero_tol,
		const char *compW_orth_method, int compW_orth_block_size, int compW_orth_max_reorth, double compW_orth_zero_tol,
		int compW_cg_max_iter, double compW_cg_rate, double compW_cg_tol, const char *compW_cg_tol_type,
		int compW_cg_auto_shift, int compRR_min_num, double compRR_min_gap, double compRR_tol,
		struct OPS_ *ops);
void EigenSolverSetParametersFromCommandLine_GCG (int argc, char *argv[], struct OPS_ *ops);

/* phase timers of the last solve (same phases as the reference's TIME_GCG table) */
typedef struct GCGE_Timing_ {
	double initX, checkconv, compP, compRR, rr_matW, dsyevx, compRV, compW, linsol, compX, total;
} GCGE_Timing;
const GCGE_Timing *GCGE_LastTiming (void);

/* ---- harness ---------------------------------------------------------------- */
/* flag: 0 BlockPCG inside GCG, 1 the back-end's own ops->MultiLinearSolver, 2 both */
int TestEigenSolverGCG (void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops);

/* Same run, but returning the results instead of printing them (used by the python
 * tests and bench.py).  eval must hold nevMax doubles.  Returns 0 on success. */
typedef struct GCGE_RunResult_ {
	int nevConv, numIter, nevMax, block_size, nevInit;
	double seconds;
	GCGE_Timing timing;
} GCGE_RunResult;
int GCGE_RunGCG (void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops,
		double *eval, void ***evec_out /* NULL: destroy */, GCGE_RunResult *res);
/* Warm start: evec is a block of nevMax columns (MultiVecCreateByMat) owned by the caller whose first nevGiven
 * columns are start vectors — the nevGiven argument of ops->EigenSolver (reference src/ops_eig_sol_gcg.c:101-158,
 * 1253); the eigenvectors are returned in the same block. */
int GCGE_RunGCGGiven (void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops,
		double *eval, void **evec, int nevGiven, GCGE_RunResult *res);

#ifdef __cplusplus
}
#endif
#endif
