/* TEST INFRASTRUCTURE — never linked into the product.
 *
 * Thin ctypes-friendly entry points over the REAL reference (compiled in place
 * from /root/reference by oracle/Makefile into oracle/_ref/libgcge_ref.so).
 * This file contains no reference code: it only calls the reference's public
 * API (src/ops.h, src/ops_orth.h, src/ops_lin_sol.h, src/ops_eig_sol_gcg.h,
 * app/app_ccs.h) on plain arrays, so that tests/golden/make_golden.py can
 * produce golden vectors and tests can pin oracle/cpu_backend.c against it.
 *
 * Conventions: dense blocks are column-major (the reference's LAPACKVEC,
 * app/app_lapack.h:17-20) with ld == nrows; sparse matrices are the CCS triple
 * of a symmetric matrix (== CSR arrays).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>

#include "ops.h"
#include "ops_orth.h"
#include "ops_lin_sol.h"
#include "ops_eig_sol_gcg.h"
#include "app_ccs.h"
#include "app_lapack.h"

static int g_verbose = 0;
static void quiet_printf(const char *fmt, ...) { (void)fmt; }
static double wall_now(void)
{
	struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void ref_set_verbose(int v) { g_verbose = v; }

static OPS *make_ops(void)
{
	OPS *ops = NULL;
	OPS_Create(&ops);
	OPS_CCS_Set(ops);
	OPS_Setup(ops);
	if (!g_verbose) { ops->Printf = quiet_printf; ops->lapack_ops->Printf = quiet_printf; }
	ops->GetWtime = wall_now;
	return ops;
}
static void set_ccs(CCSMAT *m, int n, int *rowptr, int *colidx, double *val)
{
	m->nrows = n; m->ncols = n; m->j_col = rowptr; m->i_row = colidx; m->data = val;
}
static void set_vec(LAPACKVEC *v, double *data, int nrows, int ncols)
{
	v->data = data; v->nrows = nrows; v->ncols = ncols; v->ldd = nrows;
}

/* Drop-in check: the REFERENCE solver stack (GCG, ModifiedGramSchmidt, BlockPCG, OPS_Setup defaults)
 * driving a back-end that was written against OUR gcge_ops.h.  `foreign_ops` is a table created and
 * filled by that back-end (its MultiVec and MatDotMultiVec slots); everything else is the reference one. */
static OPS *g_foreign_ops = NULL;
void ref_use_foreign_backend(void *foreign_ops) { g_foreign_ops = (OPS*)foreign_ops; }
void *ref_make_ccs_ops(void) { return (void*)make_ops(); }   /* reference app_ccs table, for OUR solver to drive */

/* ---- eigensolver: same parameter flow as test/test_eig_sol_gcg.c:28-169 ---- */
/* core: any operator table + the matrix handles of ITS back-end.  given != NULL (app_ccs tables only): the first
 * nevGiven columns of the eigenvector block (column-major, n x nevGiven) are start vectors */
static int gcg_solve_core(OPS *ops, int own_ops, void *matA, void *matB, int n,
		int nevConv, int nevMax, int block_size, int nevInit,
		double abs_tol, double rel_tol, int max_iter, int flag,
		int argc, char **argv,
		double *eval_out, double *evec_out, int *nevConv_out, int *numIter_out,
		double *seconds_out, int nevGiven, const double *given)
{
	int multiMax = 1; double gapMin = 1e-5;
	if (nevMax <= 0) nevMax = 2 * nevConv;
	if (block_size <= 0) block_size = nevConv < 30 ? (nevMax - nevConv) : nevConv / 5;
	if (nevInit <= 0 || nevInit > nevMax) nevInit = nevMax;
	double tol_gcg[2] = {abs_tol, rel_tol};
	double *eval = calloc(nevMax, sizeof(double)); void **evec;
	ops->MultiVecCreateByMat(&evec, nevMax, matA, ops);
	ops->MultiVecSetRandomValue(evec, 0, nevMax, ops);
	if (given != NULL && nevGiven > 0 && own_ops)
		memcpy(((LAPACKVEC*)evec)->data, given, (size_t)n * nevGiven * sizeof(double));
	else nevGiven = 0;
	void **ws[4]; double *dbl_ws; int *int_ws;
	ops->MultiVecCreateByMat(&ws[0], nevMax + 2 * block_size, matA, ops);
	ops->MultiVecSetRandomValue(ws[0], 0, nevMax + 2 * block_size, ops);
	int i;
	for (i = 1; i < 4; ++i) {
		ops->MultiVecCreateByMat(&ws[i], block_size, matA, ops);
		ops->MultiVecSetRandomValue(ws[i], 0, block_size, ops);
	}
	int sizeV = nevInit + 2 * block_size;
	int length_dbl_ws = 2 * sizeV * sizeV + 10 * sizeV + (nevMax + 2 * block_size) + nevMax * block_size;
	int length_int_ws = 6 * sizeV + 2 * (block_size + 3);
	dbl_ws = calloc(length_dbl_ws, sizeof(double));
	int_ws = calloc(length_int_ws, sizeof(int));

	srand(0);
	double t0 = wall_now();
	EigenSolverSetup_GCG(multiMax, gapMin, nevInit, nevMax, block_size,
			tol_gcg, max_iter, flag, ws, dbl_ws, int_ws, ops);
	EigenSolverSetParameters_GCG(50,
			"mgs", 80, 2, 2 * DBL_EPSILON,
			"mgs", -1, 2, 2 * DBL_EPSILON,
			"mgs", 80, 2, 2 * DBL_EPSILON,
			30, 1e-2, 1e-14, "abs", 0,
			-1, gapMin, 2 * DBL_EPSILON, ops);
	if (argc > 0) {
		int k, has = 0;
		for (k = 0; k < argc; ++k) if (0 == strcmp(argv[k], "-gcge_print_usage")) has = 1;
		if (has) EigenSolverSetParametersFromCommandLine_GCG(argc, argv, ops);
		else {
			char **av = malloc((argc + 2) * sizeof(char*));
			memcpy(av, argv, argc * sizeof(char*));
			av[argc] = "-gcge_print_usage"; av[argc + 1] = "0";
			EigenSolverSetParametersFromCommandLine_GCG(argc + 2, av, ops);
			free(av);
		}
	}
	int conv = nevConv;
	ops->EigenSolver(matA, matB, eval, evec, nevGiven, &conv, ops);
	double t1 = wall_now();
	if (seconds_out) *seconds_out = t1 - t0;
	if (numIter_out) *numIter_out = ((GCGSolver*)ops->eigen_solver_workspace)->numIter;
	if (nevConv_out) *nevConv_out = conv;
	memcpy(eval_out, eval, nevMax * sizeof(double));
	if (evec_out && own_ops) memcpy(evec_out, ((LAPACKVEC*)evec)->data, (size_t)n * nevMax * sizeof(double));

	ops->MultiVecDestroy(&ws[0], nevMax + 2 * block_size, ops);
	for (i = 1; i < 4; ++i) ops->MultiVecDestroy(&ws[i], block_size, ops);
	ops->MultiVecDestroy(&evec, nevMax, ops);
	free(dbl_ws); free(int_ws); free(eval);
	return 0;
}
int ref_gcg_solve_given(int n, int *a_rowptr, int *a_colidx, double *a_val,
		int *b_rowptr, int *b_colidx, double *b_val,
		int nevConv, int nevMax, int block_size, int nevInit,
		double abs_tol, double rel_tol, int max_iter, int flag,
		int argc, char **argv,
		double *eval_out, double *evec_out, int *nevConv_out, int *numIter_out,
		double *seconds_out, int nevGiven, const double *given)
{
	OPS *ops;
	CCSMAT A, B; void *matA, *matB = NULL;
	if (g_foreign_ops != NULL) {
		ops = g_foreign_ops;
		OPS_Setup(ops);                       /* the reference back-fills lapack_ops, QtAP, InnerProd */
		if (!g_verbose) { ops->Printf = quiet_printf; ops->lapack_ops->Printf = quiet_printf; }
	} else ops = make_ops();
	set_ccs(&A, n, a_rowptr, a_colidx, a_val); matA = &A;
	if (b_rowptr != NULL) { set_ccs(&B, n, b_rowptr, b_colidx, b_val); matB = &B; }
	int rc = gcg_solve_core(ops, g_foreign_ops == NULL, matA, matB, n, nevConv, nevMax, block_size, nevInit, abs_tol,
			rel_tol, max_iter, flag, argc, argv, eval_out, evec_out, nevConv_out, numIter_out, seconds_out, nevGiven, given);
	if (g_foreign_ops == NULL) OPS_Destroy(&ops);
	return rc;
}
/* The literal drop-in: `foreign_ops` was created and filled by ANOTHER back-end's OPS_xxx_Set (OPS_HIP_Set of
 * libgcge_hip.so), matA / matB are that back-end's own matrix handles.  The reference's OPS_Setup back-fills every
 * default (src/ops.c:60-149), then the reference's GCG / ModifiedGramSchmidt / BlockPCG run over the foreign slots.
 * flag = 1: the table's own ops->MultiLinearSolver is used for the W systems (ops_eig_sol_gcg.c:584-618). */
int ref_gcg_solve_foreign(void *foreign_ops, void *matA, void *matB,
		int nevConv, int nevMax, int block_size, int nevInit,
		double abs_tol, double rel_tol, int max_iter, int flag,
		double *eval_out, int *nevConv_out, int *numIter_out, double *seconds_out)
{
	OPS *ops = (OPS*)foreign_ops;
	void (*lin_sol)(void*, void**, void**, int*, int*, struct OPS_*) = ops->MultiLinearSolver;
	void *lin_ws = ops->multi_linear_solver_workspace;
	OPS_Setup(ops);
	ops->MultiLinearSolver = lin_sol; ops->multi_linear_solver_workspace = lin_ws;
	if (!g_verbose) { ops->Printf = quiet_printf; ops->lapack_ops->Printf = quiet_printf; }
	ops->GetWtime = wall_now;
	return gcg_solve_core(ops, 0, matA, matB, 0, nevConv, nevMax, block_size, nevInit, abs_tol, rel_tol, max_iter,
			flag, 0, NULL, eval_out, NULL, nevConv_out, numIter_out, seconds_out, 0, NULL);
}
/* The reference's OWN harness function — TestEigenSolverGCG of test/test_eig_sol_gcg.c:28-169, compiled into this library by
 * oracle/Makefile — over a table another back-end's OPS_xxx_Set filled (OPS_HIP_Set), with that back-end's matrix handles:
 * north_star's "TestEigenSolverGCG() is a drop-in", literally.  The harness reports through ops->Printf only ("numIter = %d,
 * nevConv = %d", then "%d: %6.14e" per converged pair, :139-165), so Printf is pointed at a buffer the caller parses. */
extern int TestEigenSolverGCG(void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops);
static char *g_log = NULL; static size_t g_log_len = 0, g_log_cap = 0;
#include <stdarg.h>
static void capture_printf(const char *fmt, ...)
{
	char line[1024];
	va_list ap; va_start(ap, fmt);
	int k = vsnprintf(line, sizeof line, fmt, ap);
	va_end(ap);
	if (k < 0) return;
	if (k >= (int)sizeof line) k = (int)sizeof line - 1;
	if (g_log_len + (size_t)k + 1 > g_log_cap) {
		g_log_cap = 2 * (g_log_len + (size_t)k + 1) + 4096;
		g_log = realloc(g_log, g_log_cap);
	}
	memcpy(g_log + g_log_len, line, (size_t)k); g_log_len += (size_t)k; g_log[g_log_len] = 0;
}
/* returns the harness's return value; *log_out points at its whole output (valid until the next call) */
int ref_test_eigen_solver_gcg(void *foreign_ops, void *matA, void *matB, int flag, int argc, char **argv, const char **log_out)
{
	OPS *ops = (OPS*)foreign_ops;
	void (*lin_sol)(void*, void**, void**, int*, int*, struct OPS_*) = ops->MultiLinearSolver;
	void *lin_ws = ops->multi_linear_solver_workspace;
	OPS_Setup(ops);                      /* the reference's defaulting, as test/main.c does after OPS_xxx_Set */
	ops->MultiLinearSolver = lin_sol; ops->multi_linear_solver_workspace = lin_ws;
	ops->Printf = capture_printf; ops->lapack_ops->Printf = capture_printf;
	ops->GetWtime = wall_now;
	g_log_len = 0; if (g_log) g_log[0] = 0;
	int rc = TestEigenSolverGCG(matA, matB, flag, argc, argv, ops);
	if (log_out) *log_out = g_log ? g_log : "";
	return rc;
}
/* address of a function of THIS library by name: lets a test see that the reference's tables hold the reference's own
 * functions (the library is linked -Bsymbolic: its calls to its exported functions must not be interposed) */
void *ref_address_of(const char *name)
{
	if (0 == strcmp(name, "OPS_Setup")) return (void*)OPS_Setup;
	if (0 == strcmp(name, "EigenSolverSetup_GCG")) return (void*)EigenSolverSetup_GCG;
	if (0 == strcmp(name, "TestEigenSolverGCG")) return (void*)TestEigenSolverGCG;
	if (0 == strcmp(name, "DefaultMultiVecQtAP")) return (void*)DefaultMultiVecQtAP;
	return NULL;
}
int ref_gcg_solve(int n, int *a_rowptr, int *a_colidx, double *a_val,
		int *b_rowptr, int *b_colidx, double *b_val,
		int nevConv, int nevMax, int block_size, int nevInit,
		double abs_tol, double rel_tol, int max_iter, int flag,
		int argc, char **argv,
		double *eval_out, double *evec_out, int *nevConv_out, int *numIter_out,
		double *seconds_out)
{
	return ref_gcg_solve_given(n, a_rowptr, a_colidx, a_val, b_rowptr, b_colidx, b_val, nevConv, nevMax, block_size,
			nevInit, abs_tol, rel_tol, max_iter, flag, argc, argv, eval_out, evec_out, nevConv_out, numIter_out,
			seconds_out, 0, NULL);
}

/* ---- slot-level calls on plain column-major arrays ---- */
void ref_spmm(int n, int *rowptr, int *colidx, double *val,
		double *x, int ncx, double *y, int ncy, int *start, int *end)
{
	OPS *ops = make_ops(); CCSMAT A; LAPACKVEC vx, vy;
	set_vec(&vx, x, n, ncx); set_vec(&vy, y, n, ncy);
	if (rowptr) { set_ccs(&A, n, rowptr, colidx, val); ops->MatDotMultiVec(&A, (void**)&vx, (void**)&vy, start, end, ops); }
	else ops->MatDotMultiVec(NULL, (void**)&vx, (void**)&vy, start, end, ops);
	OPS_Destroy(&ops);
}
void ref_inner_prod(char nsd, int n, double *x, int ncx, double *y, int ncy,
		int *start, int *end, double *ip, int ldip)
{
	OPS *ops = make_ops(); LAPACKVEC vx, vy;
	set_vec(&vx, x, n, ncx); set_vec(&vy, y, n, ncy);
	ops->MultiVecInnerProd(nsd, (void**)&vx, (void**)&vy, 0, start, end, ip, ldip, ops);
	OPS_Destroy(&ops);
}
void ref_qtap(char ntsA, char ntsd, int n, double *q, int ncq,
		int *rowptr, int *colidx, double *val, double *p, int ncp,
		int *start, int *end, double *qap, int ldqap, double *ws, int ncws)
{
	OPS *ops = make_ops(); CCSMAT A; LAPACKVEC vq, vp, vw; void *mat = NULL;
	set_vec(&vq, q, n, ncq); set_vec(&vp, p, n, ncp); set_vec(&vw, ws, n, ncws);
	if (rowptr) { set_ccs(&A, n, rowptr, colidx, val); mat = &A; }
	ops->MultiVecQtAP(ntsA, ntsd, (void**)&vq, mat, (void**)&vp, 0, start, end, qap, ldqap, (void**)&vw, ops);
	OPS_Destroy(&ops);
}
void ref_axpby(double alpha, int n, double *x, int ncx, double beta, double *y, int ncy,
		int *start, int *end)
{
	OPS *ops = make_ops(); LAPACKVEC vx, vy;
	if (x) set_vec(&vx, x, n, ncx);
	set_vec(&vy, y, n, ncy);
	ops->MultiVecAxpby(alpha, x ? (void**)&vx : NULL, beta, (void**)&vy, start, end, ops);
	OPS_Destroy(&ops);
}
void ref_lincomb(int n, double *x, int ncx, double *y, int ncy, int *start, int *end,
		double *coef, int ldc, double *beta, int incb)
{
	OPS *ops = make_ops(); LAPACKVEC vx, vy;
	if (x) set_vec(&vx, x, n, ncx);
	set_vec(&vy, y, n, ncy);
	ops->MultiVecLinearComb(x ? (void**)&vx : NULL, (void**)&vy, 0, start, end, coef, ldc, beta, incb, ops);
	OPS_Destroy(&ops);
}
void ref_set_random(unsigned seed, int n, double *x, int ncx, int start, int end)
{
	OPS *ops = make_ops(); LAPACKVEC vx;
	set_vec(&vx, x, n, ncx);
	srand(seed);
	ops->MultiVecSetRandomValue((void**)&vx, start, end, ops);
	OPS_Destroy(&ops);
}
/* method: 0 = ModifiedGramSchmidt (ops_orth.c:203), 1 = BinaryGramSchmidt (ops_orth.c:517) */
int ref_orth(int method, int n, double *x, int ncx, int start, int end,
		int *rowptr, int *colidx, double *val,
		int block_size, int max_reorth, double zero_tol)
{
	OPS *ops = make_ops(); CCSMAT B; LAPACKVEC vx, vw; void *mat = NULL;
	int ncws = end - start > 0 ? end : 1;
	double *ws = calloc((size_t)n * ncws, sizeof(double));
	double *dbl = calloc((size_t)ncx * ncx * 4 + 64, sizeof(double));
	set_vec(&vx, x, n, ncx); set_vec(&vw, ws, n, ncws);
	if (rowptr) { set_ccs(&B, n, rowptr, colidx, val); mat = &B; }
	if (method == 0) MultiVecOrthSetup_ModifiedGramSchmidt(block_size, max_reorth, zero_tol, (void**)&vw, dbl, ops);
	else MultiVecOrthSetup_BinaryGramSchmidt(block_size, max_reorth, zero_tol, (void**)&vw, dbl, ops);
	ops->MultiVecOrth((void**)&vx, start, &end, mat, ops);
	free(ws); free(dbl);
	OPS_Destroy(&ops);
	return end;
}
/* BlockPCG (ops_lin_sol.c:140): b columns [start[0],end[0]) , x columns [start[1],end[1]) */
void ref_block_pcg(int n, int *rowptr, int *colidx, double *val,
		double *b, int ncb, double *x, int ncx, int *start, int *end,
		int max_iter, double rate, double tol, const char *tol_type,
		int *niter_out, double *residual_out)
{
	OPS *ops = make_ops(); CCSMAT A; LAPACKVEC vb, vx, w0, w1, w2; void **mv_ws[3];
	int nc = end[0] - start[0];
	double *r = calloc((size_t)n * nc, sizeof(double)), *p = calloc((size_t)n * nc, sizeof(double));
	double *w = calloc((size_t)n * nc, sizeof(double));
	double *dbl = calloc(6 * nc + 8, sizeof(double)); int *iw = calloc(2 * nc + 8, sizeof(int));
	set_ccs(&A, n, rowptr, colidx, val);
	set_vec(&vb, b, n, ncb); set_vec(&vx, x, n, ncx);
	set_vec(&w0, r, n, nc); set_vec(&w1, p, n, nc); set_vec(&w2, w, n, nc);
	mv_ws[0] = (void**)&w0; mv_ws[1] = (void**)&w1; mv_ws[2] = (void**)&w2;
	MultiLinearSolverSetup_BlockPCG(max_iter, rate, tol, tol_type, mv_ws, dbl, iw, NULL, NULL, ops);
	ops->MultiLinearSolver(&A, (void**)&vb, (void**)&vx, start, end, ops);
	BlockPCGSolver *s = (BlockPCGSolver*)ops->multi_linear_solver_workspace;
	if (niter_out) *niter_out = s->niter;
	if (residual_out) *residual_out = s->residual;
	free(r); free(p); free(w); free(dbl); free(iw);
	OPS_Destroy(&ops);
}
/* host dense helpers (app_lapack.c:24-227, :653-699) */
void ref_dense_qtap(char ntluA, char nsdC, int nrowsA, int ncolsA, int nrowsC, int ncolsC,
		double alpha, double *Q, int ldQ, double *A, int ldA, double *P, int ldP,
		double beta, double *C, int ldC, double *ws)
{
	OPS *ops = make_ops();
	ops->DenseMatQtAP(ntluA, nsdC, nrowsA, ncolsA, nrowsC, ncolsC, alpha, Q, ldQ, A, ldA, P, ldP, beta, C, ldC, ws);
	OPS_Destroy(&ops);
}
int ref_dense_orth(double *mat, int nrows, int ldm, int start, int end, double zero_tol)
{
	OPS *ops = make_ops();
	int n = end - start, length = 2 * n + (n + 1) * 64 + nrows + nrows * n + 64;
	double *dbl = calloc(length, sizeof(double)); int *iw = calloc(n + 8, sizeof(int));
	ops->DenseMatOrth(mat, nrows, ldm, start, &end, zero_tol, dbl, length, iw);
	free(dbl); free(iw);
	OPS_Destroy(&ops);
	return end;
}

/* ---- multigrid: the reference's dense back-end (app/app_lapack.c) is the only built-in one that can run BlockAMG — app_ccs.c:140-150
 * refuses a rectangular P^T.  Blocks are column-major, ld == rows. */
static OPS *make_dense_ops(void)
{
	OPS *ops = NULL;
	OPS_Create(&ops);
	OPS_LAPACK_Set(ops);
	OPS_Setup(ops);
	if (!g_verbose) { ops->Printf = quiet_printf; if (ops->lapack_ops) ops->lapack_ops->Printf = quiet_printf; }
	ops->GetWtime = wall_now;
	return ops;
}
/* the reference's own (toy) hierarchy of a dense matrix: MultiGridCreate, app_lapack.c:863-929.  n_out[l] = rows of level l;
 * A_out / P_out: the level matrices one after the other (column-major); returns the number of levels */
int ref_dense_multigrid(int n0, double *A0, int num_levels, int *n_out, double *A_out, double *P_out)
{
	OPS *ops = make_dense_ops(); LAPACKMAT A; void **A_array, **B_array = NULL, **P_array; int l; size_t oa = 0, op = 0;
	A.data = A0; A.nrows = n0; A.ncols = n0; A.ldd = n0;
	ops->MultiGridCreate(&A_array, &B_array, &P_array, &num_levels, &A, NULL, ops);
	for (l = 0; l < num_levels; ++l) {
		LAPACKMAT *a = (LAPACKMAT*)A_array[l];
		n_out[l] = a->nrows;
		memcpy(A_out + oa, a->data, (size_t)a->nrows * a->ncols * sizeof(double)); oa += (size_t)a->nrows * a->ncols;
		if (l + 1 < num_levels) {
			LAPACKMAT *p = (LAPACKMAT*)P_array[l];
			memcpy(P_out + op, p->data, (size_t)p->nrows * p->ncols * sizeof(double)); op += (size_t)p->nrows * p->ncols;
		}
	}
	ops->MultiGridDestroy(&A_array, NULL, &P_array, &num_levels, ops);
	OPS_Destroy(&ops);
	return num_levels;
}
/* BlockAMG (src/ops_lin_sol.c:466-715) over a given hierarchy: n[l] rows, A_flat / P_flat as above (P_l: n[l] x n[l+1]);
 * b, x: n[0] x m; max_iter = {cycles, pre_0, post_0, pre_1, post_1, ...}, rate / tol per level */
void ref_block_amg_dense(int num_levels, int *n, double *A_flat, double *P_flat, int m, double *b, double *x,
		int *max_iter, double *rate, double *tol, const char *tol_type, int *niter_out, double *residual_out)
{
	OPS *ops = make_dense_ops();
	LAPACKMAT *A = calloc(num_levels, sizeof(LAPACKMAT)), *P = calloc(num_levels, sizeof(LAPACKMAT));
	void **A_array = calloc(num_levels, sizeof(void*)), **P_array = calloc(num_levels, sizeof(void*));
	void ***ws[5]; LAPACKVEC *blk = calloc(5 * (size_t)num_levels, sizeof(LAPACKVEC)), vb, vx;
	double *dbl = calloc(6 * m + 8, sizeof(double)); int *iw = calloc(2 * m + 8, sizeof(int));
	int l, i, start[2] = {0, 0}, end[2] = {m, m}; size_t oa = 0, op = 0;
	for (l = 0; l < num_levels; ++l) {
		A[l].data = A_flat + oa; A[l].nrows = A[l].ncols = A[l].ldd = n[l]; oa += (size_t)n[l] * n[l]; A_array[l] = &A[l];
		if (l + 1 < num_levels) { P[l].data = P_flat + op; P[l].nrows = P[l].ldd = n[l]; P[l].ncols = n[l + 1]; op += (size_t)n[l] * n[l + 1]; P_array[l] = &P[l]; }
	}
	for (i = 0; i < 5; ++i) {
		ws[i] = calloc(num_levels, sizeof(void**));
		for (l = 0; l < num_levels; ++l) {
			LAPACKVEC *v = &blk[5 * l + i];
			v->data = calloc((size_t)n[l] * m, sizeof(double)); v->nrows = v->ldd = n[l]; v->ncols = m;
			ws[i][l] = (void**)v;
		}
	}
	set_vec(&vb, b, n[0], m); set_vec(&vx, x, n[0], m);
	MultiLinearSolverSetup_BlockAMG(max_iter, rate, tol, tol_type, A_array, P_array, num_levels, ws, dbl, iw, NULL, ops);
	ops->MultiLinearSolver(A_array[0], (void**)&vb, (void**)&vx, start, end, ops);
	{
		BlockAMGSolver *s = (BlockAMGSolver*)ops->multi_linear_solver_workspace;
		if (niter_out) *niter_out = s->niter;
		if (residual_out) *residual_out = s->residual;
	}
	for (i = 0; i < 5; ++i) { for (l = 0; l < num_levels; ++l) free(blk[5 * l + i].data); free(ws[i]); }
	free(blk); free(A); free(P); free(A_array); free(P_array); free(dbl); free(iw);
	OPS_Destroy(&ops);
}
/* MultiVecFromItoJ (src/ops_multi_grid.c:69-117) through the dense back-end: from level i (n[i] x m in `from`) to level j */
void ref_from_i_to_j_dense(int num_levels, int *n, double *P_flat, int level_i, int level_j, int m, double *from, double *to)
{
	OPS *ops = make_dense_ops();
	LAPACKMAT *P = calloc(num_levels, sizeof(LAPACKMAT)); void **P_array = calloc(num_levels, sizeof(void*));
	void ***ws = calloc(num_levels, sizeof(void**)); LAPACKVEC *blk = calloc(num_levels, sizeof(LAPACKVEC)), vf, vt;
	int l, start[2] = {0, 0}, end[2] = {m, m}; size_t op = 0;
	for (l = 0; l + 1 < num_levels; ++l) { P[l].data = P_flat + op; P[l].nrows = P[l].ldd = n[l]; P[l].ncols = n[l + 1]; op += (size_t)n[l] * n[l + 1]; P_array[l] = &P[l]; }
	for (l = 0; l < num_levels; ++l) { blk[l].data = calloc((size_t)n[l] * m, sizeof(double)); blk[l].nrows = blk[l].ldd = n[l]; blk[l].ncols = m; ws[l] = (void**)&blk[l]; }
	set_vec(&vf, from, n[level_i], m); set_vec(&vt, to, n[level_j], m);
	ops->MultiVecFromItoJ(P_array, level_i, level_j, (void**)&vf, (void**)&vt, start, end, ws, ops);
	for (l = 0; l < num_levels; ++l) free(blk[l].data);
	free(blk); free(ws); free(P); free(P_array);
	OPS_Destroy(&ops);
}
/* The literal drop-in of the multigrid leg: the REFERENCE's BlockAMG (src/ops_lin_sol.c:466-715), its BlockPCG as the smoother and
 * its DefaultMultiVecFromItoJ, over a table another back-end filled (OPS_HIP_Set) — hierarchy from THAT back-end's
 * ops->MultiGridCreate, work blocks from its MultiVecCreateByMat; b / x are its blocks of m columns.  The set-up of
 * test/test_eig_sol_SiO2_MAT.c:96-128,160-170.  Returns the number of levels the back-end delivered. */
int ref_block_amg_foreign(void *foreign_ops, void *matA, int num_levels, int *max_iter, double *rate, double *tol,
		int m, void **b, void **x, double *residual_out)
{
	OPS *ops = (OPS*)foreign_ops;
	void **A_array = NULL, **B_array = NULL, **P_array = NULL; void ***ws[5];
	double *dbl = calloc(6 * m + 8, sizeof(double)); int *iw = calloc(2 * m + 8, sizeof(int));
	int l, i, start[2] = {0, 0}, end[2] = {m, m};
	OPS_Setup(ops);
	if (!g_verbose) { ops->Printf = quiet_printf; ops->lapack_ops->Printf = quiet_printf; }
	if (ops->MultiGridCreate == NULL) return -1;
	ops->MultiGridCreate(&A_array, &B_array, &P_array, &num_levels, matA, NULL, ops);
	for (i = 0; i < 5; ++i) {
		ws[i] = calloc(num_levels, sizeof(void**));
		for (l = 0; l < num_levels; ++l) ops->MultiVecCreateByMat(&ws[i][l], m, A_array[l], ops);
	}
	MultiLinearSolverSetup_BlockAMG(max_iter, rate, tol, "abs", A_array, P_array, num_levels, ws, dbl, iw, NULL, ops);
	ops->MultiLinearSolver(A_array[0], b, x, start, end, ops);
	if (residual_out) *residual_out = ((BlockAMGSolver*)ops->multi_linear_solver_workspace)->residual;
	for (i = 0; i < 5; ++i) { for (l = 0; l < num_levels; ++l) ops->MultiVecDestroy(&ws[i][l], m, ops); free(ws[i]); }
	l = num_levels;
	ops->MultiGridDestroy(&A_array, &B_array, &P_array, &l, ops);
	free(dbl); free(iw);
	return num_levels;
}
