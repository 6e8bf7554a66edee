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
