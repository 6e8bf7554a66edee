/* Block conjugate gradients with per-column scalars — the "inverse power" step
 * of GCG (W ~ A^-1 (lambda B x)).  Semantics of the reference's BlockPCG
 * (src/ops_lin_sol.c:140-437): every right-hand side runs its own CG recurrence
 * (own alpha, beta, rho), columns retire individually once their residual has
 * dropped by `rate` or below tol*||b||, contiguous runs of still-active columns are
 * multiplied by A together, and no preconditioner is applied.
 *
 * Written against the operator table only.  The two per-iteration reductions
 * (p^T w and r^T r) use MultiVecLocalInnerProd + one GCGE_COMM all-reduce each,
 * as the reference does with MPI_Allreduce (:317, :365).
 */
#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gcge_solver.h"

static void reduce_over_ranks(double *v, int n)
{
	GCGE_COMM *c = GCGE_GetComm();
	if (GCGE_GetLocalInnerProdReduces()) return;      /* the back-end's MultiVecLocalInnerProd summed over the ranks already */
	if (c != NULL && n > 0) c->allreduce_sum(v, n, c->ctx);
}

static void BlockPCG(void *mat, void **mv_b, void **mv_x, int *start_bx, int *end_bx, struct OPS_ *ops)
{
	BlockPCGSolver *s = (BlockPCGSolver*)ops->multi_linear_solver_workspace;
	void **mv_r = s->mv_ws[0], **mv_p = s->mv_ws[1], **mv_w = s->mv_ws[2];
	const int nrhs = end_bx[0] - start_bx[0];
	double *norm_b = s->dbl_ws, *rho1 = norm_b + nrhs, *rho2 = rho1 + nrhs, *pTw = rho2 + nrhs;
	double *init_res = pTw + nrhs, *last_res = init_res + nrhs;
	int *active = s->int_ws, *run = active + nrhs;      /* run[]: starts of contiguous runs */
	int nact, nrun, niter, i, c, start[2], end[2];
	assert(nrhs == end_bx[1] - start_bx[1]);
	if (nrhs <= 0) { s->niter = 0; return; }

	if (0 == strcmp(s->tol_type, "rel")) {
		start[0] = start_bx[0]; end[0] = end_bx[0]; start[1] = start_bx[0]; end[1] = end_bx[0];
		ops->MultiVecInnerProd('D', mv_b, mv_b, 0, start, end, norm_b, 1, ops);
		for (i = 0; i < nrhs; ++i) norm_b[i] = sqrt(norm_b[i]);
	} else if (0 == strcmp(s->tol_type, "user")) {
		for (i = 0; i < nrhs; ++i) norm_b[i] = fabs(norm_b[i]);   /* caller stored the scales */
	} else {
		for (i = 0; i < nrhs; ++i) norm_b[i] = 1.0;
	}
	/* r = b - A x ;  rho2 = diag(r^T r) */
	start[0] = start_bx[1]; end[0] = end_bx[1]; start[1] = 0; end[1] = nrhs;
	if (s->MatDotMultiVec != NULL) s->MatDotMultiVec(mv_x, mv_r, start, end, mv_p, 0, ops);
	else ops->MatDotMultiVec(mat, mv_x, mv_r, start, end, ops);
	start[0] = start_bx[0]; end[0] = end_bx[0]; start[1] = 0; end[1] = nrhs;
	ops->MultiVecAxpby(1.0, mv_b, -1.0, mv_r, start, end, ops);
	start[0] = 0; end[0] = nrhs; start[1] = 0; end[1] = nrhs;
	ops->MultiVecInnerProd('D', mv_r, mv_r, 0, start, end, rho2, 1, ops);
	for (i = 0; i < nrhs; ++i) init_res[i] = sqrt(rho2[i]);
	nact = 0;
	for (i = 0; i < nrhs; ++i)
		if (init_res[i] > s->tol * norm_b[i]) { active[nact] = i; rho2[nact] = rho2[i]; ++nact; }

	niter = 0;
	while (niter < s->max_iter && nact > 0) {
		double *out;
		/* split the active list into runs of consecutive column indices */
		nrun = 0; run[nrun++] = 0;
		for (i = 1; i < nact; ++i) if (active[i] - active[i - 1] > 1) run[nrun++] = i;
		run[nrun] = nact;
		/* p = r + beta p ; w = A p ; pTw = diag(p^T w) */
		out = pTw;
		for (i = 0; i < nrun; ++i) {
			for (c = run[i]; c < run[i + 1]; ++c) {
				double beta = (niter == 0) ? 0.0 : rho2[c] / rho1[c];
				start[0] = start[1] = active[c]; end[0] = end[1] = active[c] + 1;
				ops->MultiVecAxpby(1.0, mv_r, beta, mv_p, start, end, ops);
			}
			start[0] = start[1] = active[run[i]]; end[0] = end[1] = active[run[i + 1] - 1] + 1;
			if (s->MatDotMultiVec != NULL) s->MatDotMultiVec(mv_p, mv_w, start, end, mv_b, start_bx[0], ops);
			else ops->MatDotMultiVec(mat, mv_p, mv_w, start, end, ops);
			ops->MultiVecLocalInnerProd('D', mv_p, mv_w, 0, start, end, out, 1, ops);
			out += run[i + 1] - run[i];
		}
		reduce_over_ranks(pTw, nact);
		memcpy(rho1, rho2, nact * sizeof(double));
		/* x += alpha p ; r -= alpha w ; rho2 = diag(r^T r) */
		out = rho2;
		for (i = 0; i < nrun; ++i) {
			for (c = run[i]; c < run[i + 1]; ++c) {
				double alpha = rho2[c] / pTw[c];
				start[0] = active[c]; end[0] = active[c] + 1;
				start[1] = start_bx[1] + active[c]; end[1] = start[1] + 1;
				ops->MultiVecAxpby(alpha, mv_p, 1.0, mv_x, start, end, ops);
				start[0] = start[1] = active[c]; end[0] = end[1] = active[c] + 1;
				ops->MultiVecAxpby(-alpha, mv_w, 1.0, mv_r, start, end, ops);
			}
			start[0] = start[1] = active[run[i]]; end[0] = end[1] = active[run[i + 1] - 1] + 1;
			ops->MultiVecLocalInnerProd('D', mv_r, mv_r, 0, start, end, out, 1, ops);
			out += run[i + 1] - run[i];
		}
		reduce_over_ranks(rho2, nact);
		for (i = 0; i < nact; ++i) last_res[active[i]] = sqrt(rho2[i]);
		/* retire converged columns, compacting rho1/rho2 */
		{
			int keep = 0;
			for (i = 0; i < nact; ++i) {
				c = active[i];
				if (last_res[c] > s->rate * init_res[c] && last_res[c] > s->tol * norm_b[c]) {
					active[keep] = c; rho1[keep] = rho1[i]; rho2[keep] = rho2[i]; ++keep;
				}
			}
			nact = keep;
		}
		++niter;
	}
	s->niter = niter;
	s->residual = (niter > 0) ? last_res[active[0]] : init_res[0];
}

void MultiLinearSolverSetup_BlockPCG(int max_iter, double rate, double tol, const char *tol_type,
		void **mv_ws[3], double *dbl_ws, int *int_ws, void *pc,
		void (*MatDotMultiVec)(void **x, void **y, int *start, int *end, void **z, int s, struct OPS_ *ops),
		struct OPS_ *ops)
{
	static BlockPCGSolver bpcg;
	bpcg.max_iter = max_iter; bpcg.rate = rate; bpcg.tol = tol;
	strncpy(bpcg.tol_type, tol_type, sizeof(bpcg.tol_type) - 1);
	bpcg.tol_type[sizeof(bpcg.tol_type) - 1] = '\0';
	bpcg.mv_ws[0] = mv_ws[0]; bpcg.mv_ws[1] = mv_ws[1]; bpcg.mv_ws[2] = mv_ws[2];
	bpcg.dbl_ws = dbl_ws; bpcg.int_ws = int_ws; bpcg.pc = pc;
	bpcg.MatDotMultiVec = MatDotMultiVec;
	bpcg.niter = 0; bpcg.residual = -1.0;
	ops->multi_linear_solver_workspace = (void*)&bpcg;
	ops->MultiLinearSolver = BlockPCG;
}

/* ---- V-cycle multigrid with block CG as the smoother: the reference's BlockAlgebraicMultiGrid / BlockAMG
 * (src/ops_lin_sol.c:466-715; set up as in test/test_eig_sol_SiO2_MAT.c:96-180, test/test_multi_grid.c:97-129).
 * The hierarchy A_array / P_array comes from the back-end's MultiGridCreate slot.  One cycle on level l:
 *   max_iter[2l + 1] CG iterations on A_l x = b from the current x (pre-smoothing; on the coarsest level that is all),
 *   r = b - A_l x, restricted by P_l^T to the right-hand side of level l + 1, zero start there, the cycle on level l + 1,
 *   x += P_l (coarse x), max_iter[2l + 2] CG iterations (post-smoothing).
 * max_iter[0] cycles at most, stopping once the residual the last smoothing call reports is below tol[0].
 * Workspace per level (mv_array_ws[i][level], as the reference): 0 coarse right-hand side, 1 coarse x, 2 / 3 / 4 the CG's
 * r / p / w — 2 doubles as the residual / correction block, 4 as the scratch of MultiVecFromItoJ.
 *
 * The smoother is installed through a hook: by default MultiLinearSolverSetup_BlockPCG (the reference's choice,
 * :482-486,:626-629); a back-end may register its own block CG for ITS table (owner = the table's MatDotMultiVec slot) —
 * the HIP back-end's fused device CG — with the same stopping rules. */
static GCGE_SMOOTHER_SETUP_FN g_smoother_fn = NULL; static GCGE_SMOOTHER_RESIDUAL_FN g_smoother_res = NULL;
static void *g_smoother_owner = NULL;
void GCGE_SetBlockAMGSmoother(GCGE_SMOOTHER_SETUP_FN setup, GCGE_SMOOTHER_RESIDUAL_FN residual, void *owner)
{
	g_smoother_fn = setup; g_smoother_res = residual; g_smoother_owner = owner;
}
static int own_smoother(struct OPS_ *ops)
{
	return g_smoother_fn != NULL && g_smoother_res != NULL && g_smoother_owner == (void*)ops->MatDotMultiVec &&
	       getenv("GCGE_AMG_HOST_SMOOTHER") == NULL;
}
int GCGE_HasBlockAMGSmoother(struct OPS_ *ops) { return own_smoother(ops); }
/* residual and prolongation + correction as one sweep each where the back-end of this table offers them (include/gcge_solver.h) */
static GCGE_AMG_RESIDUAL_FN g_fuse_residual = NULL; static GCGE_AMG_PROLONG_ADD_FN g_fuse_prolong = NULL;
static void *g_fuse_owner = NULL;
void GCGE_SetBlockAMGFusions(GCGE_AMG_RESIDUAL_FN residual, GCGE_AMG_PROLONG_ADD_FN prolong_add, void *owner)
{
	g_fuse_residual = residual; g_fuse_prolong = prolong_add; g_fuse_owner = owner;
}
static GCGE_AMG_FORM_RHS_FN g_fuse_rhs = NULL; static void *g_fuse_rhs_owner = NULL;
void GCGE_SetBlockAMGFormRhs(GCGE_AMG_FORM_RHS_FN form_rhs, void *owner) { g_fuse_rhs = form_rhs; g_fuse_rhs_owner = owner; }
static int own_fusions(struct OPS_ *ops)
{
	return g_fuse_owner != NULL && g_fuse_owner == (void*)ops->MatDotMultiVec && getenv("GCGE_AMG_NO_FUSIONS") == NULL;
}
static void smoother_setup(int max_iter, double rate, double tol, const char *tol_type, void **mv_ws[3], double *dbl_ws,
		int *int_ws, struct OPS_ *ops)
{
	if (own_smoother(ops)) g_smoother_fn(max_iter, rate, tol, tol_type, ops);
	else MultiLinearSolverSetup_BlockPCG(max_iter, rate, tol, tol_type, mv_ws, dbl_ws, int_ws, NULL, NULL, ops);
}
/* residual of the smoothing call that ran last (src/ops_lin_sol.c:643: read from the BlockPCG struct behind the table) */
static double smoother_residual(struct OPS_ *ops)
{
	if (own_smoother(ops)) return g_smoother_res(ops);
	return ((BlockPCGSolver*)ops->multi_linear_solver_workspace)->residual;
}

static void BlockAlgebraicMultiGrid(int current_level, void **mv_b, void **mv_x, int *start_bx, int *end_bx, struct OPS_ *ops)
{
	BlockAMGSolver *bamg = (BlockAMGSolver*)ops->multi_linear_solver_workspace;
	void (*multi_linear_sol)(void*, void**, void**, int*, int*, struct OPS_*) = ops->MultiLinearSolver;
	const int coarsest_level = bamg->num_levels - 1, block_size = end_bx[1] - start_bx[1];
	const int fused = own_fusions(ops);
	void *A = bamg->A_array[current_level];
	void **mv_ws[3], **mv_r, **coarse_b, **coarse_x;
	int start[2], end[2];
	assert(end_bx[0] - start_bx[0] == end_bx[1] - start_bx[1]);
	mv_ws[0] = bamg->mv_array_ws[2][current_level];
	mv_ws[1] = bamg->mv_array_ws[3][current_level];
	mv_ws[2] = bamg->mv_array_ws[4][current_level];
	/* pre-smoothing (the coarsest level's "solve") */
	smoother_setup(bamg->max_iter[current_level * 2 + 1], bamg->rate[current_level], bamg->tol[current_level],
			bamg->tol_type, mv_ws, bamg->dbl_ws, bamg->int_ws, ops);
	ops->MultiLinearSolver(A, mv_b, mv_x, start_bx, end_bx, ops);
	if (current_level < coarsest_level) {
		const int coarse_level = current_level + 1;
		/* r = b - A x */
		start[0] = start_bx[1]; end[0] = end_bx[1]; start[1] = 0; end[1] = block_size;
		mv_r = bamg->mv_array_ws[2][current_level];
		if (!(fused && g_fuse_residual != NULL &&
				g_fuse_residual(A, mv_b, start_bx[0], mv_x, start_bx[1], mv_r, 0, block_size, ops))) {
			ops->MatDotMultiVec(A, mv_x, mv_r, start, end, ops);
			start[0] = start_bx[0]; end[0] = end_bx[0]; start[1] = 0; end[1] = block_size;
			ops->MultiVecAxpby(1.0, mv_b, -1.0, mv_r, start, end, ops);
		}
		/* restrict, zero start, recurse */
		coarse_b = bamg->mv_array_ws[0][coarse_level];
		coarse_x = bamg->mv_array_ws[1][coarse_level];
		start[0] = 0; end[0] = block_size; start[1] = 0; end[1] = block_size;
		ops->MultiVecFromItoJ(bamg->P_array, current_level, coarse_level, mv_r, coarse_b, start, end, bamg->mv_array_ws[4], ops);
		ops->MultiVecAxpby(0.0, NULL, 0.0, coarse_x, start, end, ops);
		ops->multi_linear_solver_workspace = (void*)bamg;
		BlockAlgebraicMultiGrid(coarse_level, coarse_b, coarse_x, start, end, ops);
		/* prolongate and correct */
		if (!(fused && g_fuse_prolong != NULL &&
				g_fuse_prolong(bamg->P_array[current_level], coarse_x, 0, mv_x, start_bx[1], block_size, ops))) {
			ops->MultiVecFromItoJ(bamg->P_array, coarse_level, current_level, coarse_x, mv_r, start, end, bamg->mv_array_ws[4], ops);
			start[0] = 0; end[0] = block_size; start[1] = start_bx[1]; end[1] = end_bx[1];
			ops->MultiVecAxpby(1.0, mv_r, 1.0, mv_x, start, end, ops);
		}
		/* post-smoothing */
		smoother_setup(bamg->max_iter[current_level * 2 + 2], bamg->rate[current_level], bamg->tol[current_level],
				bamg->tol_type, mv_ws, bamg->dbl_ws, bamg->int_ws, ops);
		ops->MultiLinearSolver(A, mv_b, mv_x, start_bx, end_bx, ops);
	}
	bamg->residual = smoother_residual(ops);
	/* the table's solver is BlockAMG again */
	ops->multi_linear_solver_workspace = (void*)bamg;
	ops->MultiLinearSolver = multi_linear_sol;
}

static void BlockAMG(void *mat, void **mv_b, void **mv_x, int *start_bx, int *end_bx, struct OPS_ *ops)
{
	BlockAMGSolver *bamg = (BlockAMGSolver*)ops->multi_linear_solver_workspace;
	int idx;
	/* systems declared as b = x diag(scale) (only to a BlockAMG that registered for them: MultiLinearSolverSetup_BlockAMG):
	 * b is formed here, once, from the initial guess; the smoothing calls then see an ordinary right-hand side */
	const double *rhs_scale = GCGE_GetLinearSolverRhsScale();
	(void)mat;      /* level 0 of the hierarchy IS the matrix (src/ops_lin_sol.c:477) */
	if (rhs_scale != NULL) {
		const int ncols = end_bx[1] - start_bx[1];
		if (!(g_fuse_rhs != NULL && g_fuse_rhs_owner == (void*)ops->MatDotMultiVec &&
				g_fuse_rhs(mv_b, start_bx[0], mv_x, start_bx[1], rhs_scale, ncols, ops))) {
			int s[2], e[2];
			s[0] = start_bx[1]; e[0] = end_bx[1]; s[1] = start_bx[0]; e[1] = end_bx[0];
			ops->MultiVecAxpby(1.0, mv_x, 0.0, mv_b, s, e, ops);
			ops->MultiVecLinearComb(NULL, mv_b, 0, s, e, NULL, 0, (double*)rhs_scale, 1, ops);
		}
		GCGE_SetLinearSolverRhsScale(NULL);
	}
	for (idx = 0; idx < bamg->max_iter[0]; ++idx) {
		BlockAlgebraicMultiGrid(0, mv_b, mv_x, start_bx, end_bx, ops);
		bamg->niter = idx + 1;
		if (bamg->residual < bamg->tol[0]) break;
	}
	if (rhs_scale != NULL) GCGE_SetLinearSolverRhsScale(rhs_scale);   /* (the caller clears it) */
}

void MultiLinearSolverSetup_BlockAMG(int *max_iter, double *rate, double *tol, const char *tol_type,
		void **A_array, void **P_array, int num_levels, void ***mv_array_ws[5], double *dbl_ws, int *int_ws,
		void *pc, struct OPS_ *ops)
{
	static BlockAMGSolver bamg;
	int i;
	bamg.max_iter = max_iter; bamg.rate = rate; bamg.tol = tol;
	strncpy(bamg.tol_type, tol_type, sizeof(bamg.tol_type) - 1);
	bamg.tol_type[sizeof(bamg.tol_type) - 1] = '\0';
	bamg.A_array = A_array; bamg.P_array = P_array; bamg.num_levels = num_levels;
	for (i = 0; i < 5; ++i) bamg.mv_array_ws[i] = mv_array_ws[i];
	bamg.dbl_ws = dbl_ws; bamg.int_ws = int_ws; bamg.pc = pc;
	bamg.niter = 0; bamg.residual = -1.0;
	ops->multi_linear_solver_workspace = (void*)&bamg;
	ops->MultiLinearSolver = BlockAMG;
	/* over a back-end that forms b = x diag(scale) in one sweep, BlockAMG takes the GCG driver's scaled right-hand sides */
	GCGE_SetRhsScaleCapabilityOfBlockAMG((g_fuse_rhs != NULL && g_fuse_rhs_owner == (void*)ops->MatDotMultiVec &&
			getenv("GCGE_AMG_NO_FUSIONS") == NULL) ? (void*)BlockAMG : NULL);
}
