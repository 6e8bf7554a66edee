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
