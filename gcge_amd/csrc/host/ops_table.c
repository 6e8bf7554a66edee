/* Operator-table life cycle and the solver-side default slots.
 *
 * Restates the behaviour of the reference's src/ops.c:26-149 (OPS_Create,
 * OPS_Setup, OPS_Destroy) and of the defaults in src/ops_multi_vec.c
 * (DefaultPrintf :26-44, DefaultGetWtime :45-56, DefaultGetOptionFromCommandLine
 * :58-95, DefaultMultiVecInnerProd :202-230, DefaultMultiVecQtAP :351-411).
 *
 * Differences by design:
 *  - GetWtime is a monotonic WALL clock (the reference falls back to clock(),
 *    i.e. CPU time, in serial builds — SURVEY.md §5.1);
 *  - the cross-rank reduction of Gram results goes through GCGE_COMM (one
 *    process per GPU, RCCL/gloo behind a callback) instead of MPI_Allreduce; a
 *    strided result (ldIP > rows) is packed contiguously before the reduction,
 *    which replaces the reference's MPI_Type_vector + user MPI_Op (src/ops.c:259-318).
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <assert.h>

#include "gcge_ops.h"

/* ---------------------------------------------------------------- comm hook */
static GCGE_COMM  g_comm_storage;
static GCGE_COMM *g_comm = NULL;

void GCGE_SetComm(const GCGE_COMM *comm)
{
	/* a communicator of one rank is dropped (nothing to sum) unless GCGE_COMM_KEEP_SINGLE is set: the single-GPU
	 * loop-back tests route every reduction through the real transport that way */
	if (comm == NULL || (comm->size <= 1 && getenv("GCGE_COMM_KEEP_SINGLE") == NULL)) { g_comm = NULL; return; }
	g_comm_storage = *comm; g_comm = &g_comm_storage;
}
GCGE_COMM *GCGE_GetComm(void) { return g_comm; }
/* Opt-in for solver stacks that sum their "local" inner products through MPI only (the reference's BlockPCG:
 * MultiVecLocalInnerProd + MPI_Allreduce under OPS_USE_MPI, src/ops_lin_sol.c:306-321,355-369): a back-end that honours the
 * switch (OPS_HIP_Set's MultiVecLocalInnerProd) then returns the sum over the ranks from its LOCAL slot as well, so a non-MPI
 * build of such a stack spans the ranks with flag 0.  Our own BlockPCG (lin_sol.c) asks before it reduces a second time. */
static int g_local_ip_reduces = 0;
void GCGE_SetLocalInnerProdReduces(int on) { g_local_ip_reduces = on != 0; }
int  GCGE_GetLocalInnerProdReduces(void) { return g_local_ip_reduces && g_comm != NULL; }

static GCGE_RESIDUAL_FN g_res_hook = NULL; static void *g_res_owner = NULL;
void GCGE_SetResidualHook(GCGE_RESIDUAL_FN fn, void *owner) { g_res_hook = fn; g_res_owner = owner; }
GCGE_RESIDUAL_FN GCGE_GetResidualHook(void *owner) { return (g_res_hook != NULL && owner == g_res_owner) ? g_res_hook : NULL; }
/* (two owners: a back-end's block CG and, over it, BlockAMG where the back-end forms b in one sweep — lin_sol.c) */
static void *g_rhs_scale_owner[2] = {NULL, NULL}; static const double *g_rhs_scale = NULL;
void GCGE_SetRhsScaleCapability(void *owner) { g_rhs_scale_owner[0] = owner; }
void GCGE_SetRhsScaleCapabilityOfBlockAMG(void *owner) { g_rhs_scale_owner[1] = owner; }
int GCGE_HasRhsScaleCapability(void *owner)
{
	return owner != NULL && (owner == g_rhs_scale_owner[0] || owner == g_rhs_scale_owner[1]) && getenv("GCGE_NO_RHS_SCALE") == NULL;
}
void GCGE_SetLinearSolverRhsScale(const double *scale) { g_rhs_scale = scale; }
const double *GCGE_GetLinearSolverRhsScale(void) { return g_rhs_scale; }
static void ***g_idle_blocks = NULL; static int g_idle_count = 0;
void GCGE_SetLinearSolverIdleBlocks(void ***blocks, int count) { g_idle_blocks = blocks; g_idle_count = blocks != NULL ? count : 0; }
void ***GCGE_GetLinearSolverIdleBlocks(int *count) { if (count != NULL) *count = g_idle_count; return g_idle_blocks; }
static void *g_inplace_owner = NULL; static int g_inplace_cols = 0;
void GCGE_SetInplaceLinearComb(void *owner, int max_cols) { g_inplace_owner = owner; g_inplace_cols = max_cols; }
int GCGE_InplaceLinearCombCols(void *owner)
{
	return (g_inplace_owner != NULL && owner == g_inplace_owner && getenv("GCGE_NO_INPLACE_LINCOMB") == NULL) ? g_inplace_cols : 0;
}

static double g_ls_sigma = 0.0; static void *g_ls_matB = NULL;
void GCGE_SetLinearSolverShift(double sigma, void *matB) { g_ls_sigma = sigma; g_ls_matB = matB; }
static const double *g_user_scale = NULL; static int g_user_scale_n = 0;
void GCGE_SetLinearSolverUserScale(const double *scale, int n) { g_user_scale = scale; g_user_scale_n = scale != NULL ? n : 0; }
const double *GCGE_GetLinearSolverUserScale(int *n) { if (n != NULL) *n = g_user_scale_n; return g_user_scale; }
void GCGE_GetLinearSolverShift(double *sigma, void **matB)
{
	if (sigma) *sigma = g_ls_sigma;
	if (matB) *matB = g_ls_matB;
}

/* ---------------------------------------------------------------- services */
void DefaultPrintf(const char *fmt, ...)
{
	va_list ap;
	if (g_comm != NULL && g_comm->rank != 0) return;   /* rank 0 prints */
	va_start(ap, fmt);
	vprintf(fmt, ap);
	va_end(ap);
}

double DefaultGetWtime(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* scan argv for `name`; the token after it is parsed as int ('i'), double ('f')
 * or copied as a string ('s').  Returns 1 when a value was read. */
int DefaultGetOptionFromCommandLine(const char *name, char type, void *value,
		int argc, char *argv[], struct OPS_ *ops)
{
	int k;
	for (k = 0; k < argc; ++k) {
		if (argv[k] == NULL || strcmp(argv[k], name) != 0) continue;
		if (ops != NULL && ops->Printf != NULL)
			ops->Printf("argv[%d] = \"%s\", name = \"%s\"\n", k, argv[k], name);
		if (k + 1 >= argc) return 0;
		switch (type) {
			case 'i': *(int*)value    = atoi(argv[k + 1]); break;
			case 'f': *(double*)value = atof(argv[k + 1]); break;
			case 's': strcpy((char*)value, argv[k + 1]);   break;
			default : break;
		}
		return 1;
	}
	return 0;
}

/* ---------------------------------------------------------------- defaults */
void DefaultMultiVecInnerProd(char nsdIP, void **x, void **y, int is_vec,
		int *start, int *end, double *inner_prod, int ldIP, struct OPS_ *ops)
{
	int nrows = end[0] - start[0], ncols = end[1] - start[1];
	ops->MultiVecLocalInnerProd(nsdIP, x, y, is_vec, start, end, inner_prod, ldIP, ops);
	if (g_comm == NULL || nrows <= 0 || ncols <= 0) return;
	if (nsdIP == 'D') nrows = 1;          /* one value per column, stride ldIP */
	if (nrows == ldIP) {
		g_comm->allreduce_sum(inner_prod, nrows * ncols, g_comm->ctx);
	} else {
		double *pack = (double*)malloc((size_t)nrows * ncols * sizeof(double));
		int c;
		for (c = 0; c < ncols; ++c)
			memcpy(pack + (size_t)c * nrows, inner_prod + (size_t)c * ldIP, nrows * sizeof(double));
		g_comm->allreduce_sum(pack, nrows * ncols, g_comm->ctx);
		for (c = 0; c < ncols; ++c)
			memcpy(inner_prod + (size_t)c * ldIP, pack + (size_t)c * nrows, nrows * sizeof(double));
		free(pack);
	}
}

/* qAp = Q[:,s0:e0)^T A P[:,s1:e1).  A != NULL: mv_ws[:,0:m) = A P is a visible side
 * effect the orthogonalisation re-uses; 'T' stores the transpose (m x k). */
void DefaultMultiVecQtAP(char ntsA, char ntsdQAP, void **mvQ, void *matA, void **mvP,
		int is_vec, int *startQP, int *endQP, double *qAp, int ldQAP,
		void **mv_ws, struct OPS_ *ops)
{
	int s[2], e[2];
	int k = endQP[0] - startQP[0], m = endQP[1] - startQP[1];
	if (k <= 0 || m <= 0) return;
	if (matA == NULL) {
		if (ntsdQAP == 'T') {
			s[0] = startQP[1]; e[0] = endQP[1]; s[1] = startQP[0]; e[1] = endQP[0];
			ops->MultiVecInnerProd('N', mvP, mvQ, is_vec, s, e, qAp, ldQAP, ops);
		} else {
			ops->MultiVecInnerProd(ntsdQAP, mvQ, mvP, is_vec, startQP, endQP, qAp, ldQAP, ops);
		}
		return;
	}
	s[0] = startQP[1]; e[0] = endQP[1]; s[1] = 0; e[1] = m;
	if (ntsA == 'T') ops->MatTransDotMultiVec(matA, mvP, mv_ws, s, e, ops);
	else             ops->MatDotMultiVec     (matA, mvP, mv_ws, s, e, ops);
	if (ntsdQAP == 'T') {
		s[0] = 0; e[0] = m; s[1] = startQP[0]; e[1] = endQP[0];
		ops->MultiVecInnerProd('N', mv_ws, mvQ, is_vec, s, e, qAp, ldQAP, ops);
	} else {
		s[0] = startQP[0]; e[0] = endQP[0]; s[1] = 0; e[1] = m;
		ops->MultiVecInnerProd(ntsdQAP, mvQ, mv_ws, is_vec, s, e, qAp, ldQAP, ops);
	}
}

/* ---------------------------------------------------------------- multigrid transfers
 * src/ops_multi_grid.c:20-117.  P_array[k] maps level k + 1 to level k.  From a coarse level i to a finer level j the vector is
 * multiplied by P_{i-1}, ..., P_j in turn (MatDot*), from a fine level i to a coarser level j by P_i^T, ..., P_{j-1}^T
 * (MatTransDot*); the levels in between are staged in the caller's per-level work vectors; i == j copies. */
void DefaultVecFromItoJ(void **P_array, int level_i, int level_j, void *vec_i, void *vec_j, void **vec_ws, struct OPS_ *ops)
{
	void *from, *to; int k;
	if (level_i > level_j) {
		for (k = level_i; k > level_j; --k) {
			from = (k == level_i) ? vec_i : vec_ws[k];
			to = (k == level_j + 1) ? vec_j : vec_ws[k - 1];
			ops->MatDotVec(P_array[k - 1], from, to, ops);
		}
	} else if (level_i < level_j) {
		for (k = level_i; k < level_j; ++k) {
			from = (k == level_i) ? vec_i : vec_ws[k];
			to = (k == level_j - 1) ? vec_j : vec_ws[k + 1];
			ops->MatTransDotVec(P_array[k], from, to, ops);
		}
	} else {
		ops->VecAxpby(1.0, vec_i, 0.0, vec_j, ops);
	}
}
void DefaultMultiVecFromItoJ(void **P_array, int level_i, int level_j, void **multi_vec_i, void **multi_vec_j,
		int *startIJ, int *endIJ, void ***multi_vec_ws, struct OPS_ *ops)
{
	void **from, **to; int k, start[2], end[2];
	const int m = endIJ[0] - startIJ[0];
	if (level_i > level_j) {
		for (k = level_i; k > level_j; --k) {
			if (k == level_i) { from = multi_vec_i; start[0] = startIJ[0]; end[0] = endIJ[0]; }
			else              { from = multi_vec_ws[k]; start[0] = 0; end[0] = m; }
			if (k == level_j + 1) { to = multi_vec_j; start[1] = startIJ[1]; end[1] = endIJ[1]; }
			else                  { to = multi_vec_ws[k - 1]; start[1] = 0; end[1] = m; }
			ops->MatDotMultiVec(P_array[k - 1], from, to, start, end, ops);
		}
	} else if (level_i < level_j) {
		for (k = level_i; k < level_j; ++k) {
			if (k == level_i) { from = multi_vec_i; start[0] = startIJ[0]; end[0] = endIJ[0]; }
			else              { from = multi_vec_ws[k]; start[0] = 0; end[0] = m; }
			if (k == level_j - 1) { to = multi_vec_j; start[1] = startIJ[1]; end[1] = endIJ[1]; }
			else                  { to = multi_vec_ws[k + 1]; start[1] = 0; end[1] = m; }
			ops->MatTransDotMultiVec(P_array[k], from, to, start, end, ops);
		}
	} else {
		ops->MultiVecAxpby(1.0, multi_vec_i, 0.0, multi_vec_j, startIJ, endIJ, ops);
	}
}

/* ---------------------------------------------------------------- life cycle */
void OPS_Create(OPS **ops)
{
	*ops = (OPS*)calloc(1, sizeof(OPS));   /* every slot and workspace NULL */
}

void OPS_Setup(OPS *ops)
{
	if (ops->Printf == NULL)                   ops->Printf = DefaultPrintf;
	if (ops->GetWtime == NULL)                 ops->GetWtime = DefaultGetWtime;
	if (ops->GetOptionFromCommandLine == NULL) ops->GetOptionFromCommandLine = DefaultGetOptionFromCommandLine;
	if (ops->lapack_ops == NULL) {
		OPS_Create(&ops->lapack_ops);
		OPS_DENSE_Set(ops->lapack_ops);
	}
	if (ops->DenseMatQtAP == NULL)      ops->DenseMatQtAP = ops->lapack_ops->DenseMatQtAP;
	if (ops->DenseMatOrth == NULL)      ops->DenseMatOrth = ops->lapack_ops->DenseMatOrth;
	if (ops->MultiVecInnerProd == NULL) ops->MultiVecInnerProd = DefaultMultiVecInnerProd;
	if (ops->MultiVecQtAP == NULL)      ops->MultiVecQtAP = DefaultMultiVecQtAP;
	if (ops->VecFromItoJ == NULL)       ops->VecFromItoJ = DefaultVecFromItoJ;               /* src/ops.c:107-112 */
	if (ops->MultiVecFromItoJ == NULL)  ops->MultiVecFromItoJ = DefaultMultiVecFromItoJ;
}

void OPS_Destroy(OPS **ops)
{
	if (ops == NULL || *ops == NULL) return;
	if ((*ops)->lapack_ops != NULL) OPS_Destroy(&(*ops)->lapack_ops);
	free(*ops); *ops = NULL;
}

/* convenience for drivers that want the numbers, not the log */
static void QuietPrintf(const char *fmt, ...) { (void)fmt; }
void GCGE_SetQuiet(OPS *ops, int quiet)
{
	ops->Printf = quiet ? QuietPrintf : DefaultPrintf;
	if (ops->lapack_ops != NULL) ops->lapack_ops->Printf = ops->Printf;
}
