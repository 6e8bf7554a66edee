/* GCG (generalized conjugate gradient / block damping inverse power) eigensolver
 * for A x = lambda B x, smallest eigenpairs — outer loop, locking, P/X/W
 * construction and the Rayleigh–Ritz projection.
 *
 * Algorithm and parameter semantics follow the reference's
 * src/ops_eig_sol_gcg.c (GCG :1253-1558, InitializeX :101-158, ComputeRitzVec
 * :159-194, CheckConvergence :195-315, ComputeP :316-457, ComputeX :458-471,
 * ComputeW :472-695, ComputeRayleighRitz :925-1252, Setup/parameters :1561-1864),
 * restated from the math in SURVEY.md Appendix B.  Written from scratch:
 *   - state lives in one context struct instead of file-scope statics;
 *   - the projected eigenproblem is solved by GCGE_SymEig (no LAPACK);
 *   - all scratch that the reference carves out of aliased regions of dbl_ws is
 *     a private arena here (the caller's dbl_ws only holds ss_eval/diag/matA/evec);
 *   - only operator-table slots touch O(n) data, so the same driver runs on the
 *     HIP back-end, on the CPU oracle and on the host dense table.
 */
#include <assert.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gcge_solver.h"

static GCGE_Timing g_timing;
const GCGE_Timing *GCGE_LastTiming(void) { return &g_timing; }

typedef struct {
	struct OPS_ *ops; GCGSolver *p;
	void *A, *B;
	void **V, **ritz, **ws0, **ws1, **ws2;
	/* column bookkeeping of V = [ C | N.. X | P | W ] */
	int sizeC, sizeN, startN, endN, sizeX, endX;
	int sizeP, startP, endP, sizeW, startW, endW, sizeV;
	double *ss_eval, *ss_diag, *ss_matA, *ss_evec;   /* inside p->dbl_ws            */
	double *scratch; size_t scratch_len;             /* private arena               */
	int    *iscratch;
	int    *offsetP, *offsetW;                       /* {count; [lo,hi) ...}        */
} Ctx;

static int imin(int a, int b) { return a < b ? a : b; }

static void rebase(Ctx *c)
{
	int N = c->sizeV - c->sizeC;
	c->ss_matA = c->ss_diag + N;
	c->ss_evec = c->ss_matA + (size_t)N * N;
}

static void setup_orth(Ctx *c, const char *method, int block, int reorth, double zero_tol,
		void **mv_ws, struct OPS_ *target)
{
	if (0 == strcmp(method, "bgs"))
		MultiVecOrthSetup_BinaryGramSchmidt(block, reorth, zero_tol, mv_ws, c->scratch, target);
	else if (0 == strcmp(method, "chol"))
		MultiVecOrthSetup_CholeskyQR(block, reorth, zero_tol, mv_ws, c->scratch, target);
	else
		MultiVecOrthSetup_ModifiedGramSchmidt(block, reorth, zero_tol, mv_ws, c->scratch, target);
}

/* y = (A + sigma B) x, z[s:...) is scratch — only used by BlockPCG when sigma != 0 */
static Ctx *g_shift_ctx = NULL;
static void MatDotMultiVecShift(void **x, void **y, int *start, int *end, void **z, int s, struct OPS_ *ops)
{
	Ctx *c = g_shift_ctx; double sigma = c->p->sigma;
	ops->MatDotMultiVec(c->A, x, y, start, end, ops);
	if (sigma == 0.0) return;
	if (c->B == NULL) {
		ops->MultiVecAxpby(sigma, x, 1.0, y, start, end, ops);
	} else {
		int st[2], en[2], m = end[0] - start[0];
		st[0] = start[0]; en[0] = end[0]; st[1] = s; en[1] = s + m;
		ops->MatDotMultiVec(c->B, x, z, st, en, ops);
		st[0] = s; en[0] = s + m; st[1] = start[1]; en[1] = end[1];
		ops->MultiVecAxpby(sigma, z, 1.0, y, st, en, ops);
	}
}

static void InitializeX(Ctx *c, int nevGiven)
{
	struct OPS_ *ops = c->ops; GCGSolver *p = c->p; int s[2], e[2], given = nevGiven;
	double t0 = ops->GetWtime();
	s[0] = 0; e[0] = nevGiven; s[1] = 0; e[1] = nevGiven;
	if (nevGiven > 0) ops->MultiVecAxpby(1.0, c->ritz, 0.0, c->V, s, e, ops);
	ops->Printf("sizeX = %d, nevGiven = %d, %s\n", c->sizeX, nevGiven, p->initX_orth_method);
	setup_orth(c, p->initX_orth_method, p->initX_orth_block_size, p->initX_orth_max_reorth,
			p->initX_orth_zero_tol, c->ritz, ops);
	ops->MultiVecOrth(c->V, 0, &given, c->B, ops);
	ops->MultiVecSetRandomValue(c->V, given, c->sizeX, ops);
	ops->MultiVecOrth(c->V, given, &c->endX, c->B, ops);
	assert(c->endX == c->sizeX);      /* the random block must have full rank */
	g_timing.initX += ops->GetWtime() - t0;
}

static void ComputeRitzVec(Ctx *c)
{
	struct OPS_ *ops = c->ops; int s[2], e[2]; double t0 = ops->GetWtime();
	s[0] = c->startN; e[0] = c->endW; s[1] = c->startN; e[1] = c->endX;
	ops->MultiVecLinearComb(c->V, c->ritz, 0, s, e, c->ss_evec, c->sizeV - c->sizeC, NULL, 0, ops);
	g_timing.compRV += ops->GetWtime() - t0;
}

/* residual test of the first numCheck active Ritz pairs; returns the new locked count and
 * writes the runs of unconverged column indices to offset[] */
static int CheckConvergence(Ctx *c, int numCheck, int *offset)
{
	struct OPS_ *ops = c->ops; GCGSolver *p = c->p;
	double *res = c->scratch, *tol = p->tol, *ev = c->ss_eval + c->startN;
	int s[2], e[2], idx, state, nun, nevConv; double t0 = ops->GetWtime();
	GCGE_RESIDUAL_FN hook = GCGE_GetResidualHook((void*)ops->MatDotMultiVec);
	if (numCheck > 0 && hook != NULL && getenv("GCGE_NO_RESIDUAL_HOOK") == NULL &&
			hook(c->A, c->B, c->ritz, c->startN, c->startN + numCheck, ev, res)) {
		GCGE_COMM *comm = GCGE_GetComm();          /* the back-end summed over its own rows */
		if (comm != NULL) comm->allreduce_sum(res, numCheck, comm->ctx);
		for (idx = 0; idx < numCheck; ++idx) res[idx] = sqrt(res[idx]);
	} else if (numCheck > 0) {
		s[0] = c->startN; e[0] = c->startN + numCheck; s[1] = 0; e[1] = numCheck;
		ops->MatDotMultiVec(c->A, c->ritz, c->ws0, s, e, ops);
		ops->MatDotMultiVec(c->B, c->ritz, c->ws1, s, e, ops);
		ops->MultiVecLinearComb(NULL, c->ws1, 0, s, e, NULL, 0, ev, 1, ops);   /* lambda B x */
		s[0] = 0; e[0] = numCheck; s[1] = 0; e[1] = numCheck;
		ops->MultiVecAxpby(-1.0, c->ws1, 1.0, c->ws0, s, e, ops);             /* A x - lambda B x */
		ops->MultiVecInnerProd('D', c->ws0, c->ws0, 0, s, e, res, 1, ops);
		for (idx = 0; idx < numCheck; ++idx) res[idx] = sqrt(res[idx]);
	}
	for (idx = 0; idx < numCheck; ++idx) {
		int bad;
		if (fabs(ev[idx]) > tol[1]) bad = (res[idx] > tol[0] || res[idx] > fabs(ev[idx]) * tol[1]);
		else bad = (res[idx] > tol[0]);
		if (bad) {
			ops->Printf("GCG: [%d] %6.14e (%6.4e, %6.4e)\n", c->startN + idx, ev[idx],
					res[idx], res[idx] / fabs(ev[idx]));
			break;
		}
	}
	/* never lock only part of a cluster */
	for (; idx > 0; --idx)
		if (fabs((ev[idx - 1] - ev[idx]) / ev[idx - 1]) > p->gapMin) break;
	nevConv = c->sizeC + idx;

	offset[0] = 0; state = 1; nun = 0;
	for (idx = 0; idx < numCheck; ++idx) {
		if (res[idx] > tol[0] || res[idx] > fabs(ev[idx]) * tol[1]) {
			if (state) { offset[offset[0] * 2 + 1] = c->startN + idx; state = 0; }
			if (++nun == c->sizeN) { offset[offset[0] * 2 + 2] = c->startN + idx + 1; ++offset[0]; break; }
		} else if (!state) {
			offset[offset[0] * 2 + 2] = c->startN + idx; ++offset[0]; state = 1;
		}
	}
	if (nun < c->sizeN) {                 /* pad with not-yet-checked columns */
		if (state == 1) offset[offset[0] * 2 + 1] = c->startN + numCheck;
		offset[offset[0] * 2 + 2] = imin(c->startN + numCheck + c->sizeN - nun, c->endX);
		assert(offset[offset[0] * 2 + 1] < offset[offset[0] * 2 + 2]);
		++offset[0];
	}
	assert(offset[0] > 0);
	g_timing.checkconv += ops->GetWtime() - t0;
	return nevConv;
}

static void ComputeP(Ctx *c, int *offset)
{
	struct OPS_ *ops = c->ops; GCGSolver *p = c->p;
	int N = c->sizeV - c->sizeC, idx, col, blk = 0, s[2], e[2], sP, eP;
	double *evec = c->ss_evec, *coef; double t0 = ops->GetWtime();
	/* gather the RR eigenvector columns of the (previously) unconverged pairs behind X */
	for (idx = 0; idx < offset[0]; ++idx) {
		int lo = offset[idx * 2 + 1], hi = offset[idx * 2 + 2];
		memmove(evec + (size_t)N * (c->sizeX - c->sizeC + blk), evec + (size_t)N * (lo - c->sizeC),
				(size_t)N * (hi - lo) * sizeof(double));
		blk += hi - lo;
	}
	c->sizeP = blk;
	/* zero the rows that belong to those same X columns */
	for (idx = 0; idx < offset[0]; ++idx) {
		int lo = offset[idx * 2 + 1], hi = offset[idx * 2 + 2];
		for (col = 0; col < c->sizeP; ++col)
			memset(evec + (size_t)N * (c->sizeX - c->sizeC + col) + (lo - c->sizeC), 0, (hi - lo) * sizeof(double));
	}
	/* orthonormalise them (Euclidean) against the first X-C eigenvector columns and each other */
	sP = c->sizeX - c->sizeC; eP = sP + c->sizeP;
	if (0 == strcmp("bqr", p->compP_orth_method)) {
		ops->DenseMatOrth(evec, N, N, sP, &eP, p->compP_orth_zero_tol,
				c->scratch, (int)imin((int)c->scratch_len, 1 << 30), c->iscratch);
	} else {
		GCGE_DENSE blockP, blockWs; struct OPS_ *dense = ops->lapack_ops;
		double *save = c->scratch;
		blockP.data = evec; blockP.nrows = N; blockP.ncols = eP; blockP.ldd = N;
		blockWs.data = c->scratch; blockWs.nrows = N; blockWs.ncols = eP - sP; blockWs.ldd = N;
		c->scratch += (size_t)N * (eP - sP);          /* orth scratch sits behind its multivector */
		setup_orth(c, p->compP_orth_method, p->compP_orth_block_size, p->compP_orth_max_reorth,
				p->compP_orth_zero_tol, (void**)&blockWs, dense);
		c->scratch = save;
		dense->MultiVecOrth((void**)&blockP, sP, &eP, NULL, dense);
	}
	c->startP = sP + c->sizeC; c->endP = eP + c->sizeC; c->sizeP = c->endP - c->startP;
	/* P = V[:, N..W) * coef, staged through a work block */
	coef = evec + (size_t)N * (c->sizeX - c->sizeC);
	if (c->sizeP <= GCGE_InplaceLinearCombCols((void*)ops->MultiVecLinearComb) &&
			c->startP >= c->startN && c->endP <= c->endW) {
		/* the P columns lie inside [N, W): a back-end that works row by row writes them in place */
		s[0] = c->startN; e[0] = c->endW; s[1] = c->startP; e[1] = c->endP;
		ops->MultiVecLinearComb(c->V, c->V, 0, s, e, coef, N, NULL, 0, ops);
	} else {
		s[0] = c->startN; e[0] = c->endW; s[1] = 0; e[1] = c->sizeP;
		ops->MultiVecLinearComb(c->V, c->ws0, 0, s, e, coef, N, NULL, 0, ops);
		s[0] = 0; e[0] = c->sizeP; s[1] = c->startP; e[1] = c->endP;
		ops->MultiVecAxpby(1.0, c->ws0, 0.0, c->V, s, e, ops);
	}
	g_timing.compP += ops->GetWtime() - t0;
}

static void ComputeX(Ctx *c)
{
	struct OPS_ *ops = c->ops; int s[2], e[2]; double t0 = ops->GetWtime();
	s[0] = c->startN; e[0] = c->endX; s[1] = c->startN; e[1] = c->endX;
	ops->MultiVecAxpby(1.0, c->ritz, 0.0, c->V, s, e, ops);
	g_timing.compX += ops->GetWtime() - t0;
}

static void ComputeW(Ctx *c, int *offset)
{
	struct OPS_ *ops = c->ops; GCGSolver *p = c->p; void **b = c->ritz;
	int s[2], e[2], idx, blk = 0, i;
	double sigma = 0.0, *scales = c->scratch, t0 = ops->GetWtime(), t1;
	void (*saved_solver)(void*, void**, void**, int*, int*, struct OPS_*) = ops->MultiLinearSolver;
	void *saved_ws = ops->multi_linear_solver_workspace;
	void **cg_ws[3];
	if (p->compW_cg_auto_shift == 1)
		sigma = -c->ss_eval[c->sizeC] + (c->ss_eval[c->sizeC + 1] - c->ss_eval[c->sizeC]) * 0.01;
	p->sigma = sigma = p->compW_cg_shift + sigma;
	/* (the reference asserts here that auto-shift and a user-defined solver are not combined, ops_eig_sol_gcg.c:497: its
	 *  hook hands the solver A only.  Ours publishes sigma and B through GCGE_SetLinearSolverShift below, so a solver
	 *  installed behind flag 1 that reads them — the fused device CG does — takes the automatic shift like a fixed one;
	 *  flag 2 solvers still get A alone and are refused) */
	assert(p->compW_cg_auto_shift == 0 || p->user_defined_multi_linear_solver != 2);

	/* B == NULL and a solver that takes "b = x diag(scale)" (GCGE_SetRhsScaleCapability): b is not formed */
	const int scaled_rhs = p->user_defined_multi_linear_solver == 1 && c->B == NULL &&
			GCGE_HasRhsScaleCapability((void*)ops->MultiLinearSolver);
	c->startW = c->endP;
	for (idx = 0; idx < offset[0]; ++idx) {
		int lo = offset[idx * 2 + 1], hi = offset[idx * 2 + 2], len = hi - lo;
		/* initial guess: the current Ritz vectors */
		s[0] = lo; e[0] = hi; s[1] = c->startW + blk; e[1] = s[1] + len;
		ops->MultiVecAxpby(1.0, c->ritz, 0.0, c->V, s, e, ops);
		/* right-hand side (lambda + sigma) B x, packed from column offset[1] of ritz_vec */
		for (i = 0; i < len; ++i) scales[blk + i] = c->ss_eval[lo + i] + sigma;
		if (!scaled_rhs) {
			s[0] = lo; e[0] = hi; s[1] = offset[1] + blk; e[1] = s[1] + len;
			ops->MatDotMultiVec(c->B, c->V, b, s, e, ops);
			ops->MultiVecLinearComb(NULL, b, 0, s, e, NULL, 0, scales + blk, 1, ops);
		}
		blk += len;
	}
	c->endW = c->startW + blk;

	s[0] = offset[1]; e[0] = s[0] + blk; s[1] = c->startW; e[1] = c->endW;
	t1 = ops->GetWtime();
	if (p->user_defined_multi_linear_solver == 2)
		ops->MultiLinearSolver(c->A, b, c->V, s, e, ops);
	if (p->user_defined_multi_linear_solver == 0 || p->user_defined_multi_linear_solver == 2) {
		/* scalar scratch of BlockPCG starts with the column scales ("user" tolerance type) */
		cg_ws[0] = c->ws0; cg_ws[1] = c->ws1; cg_ws[2] = c->ws2;
		g_shift_ctx = c;
		if (sigma != 0.0 && c->B != NULL && ops->MatAxpby != NULL) {
			ops->MatAxpby(sigma, c->B, 1.0, c->A, ops);
			MultiLinearSolverSetup_BlockPCG(p->compW_cg_max_iter, p->compW_cg_rate, p->compW_cg_tol,
					p->compW_cg_tol_type, cg_ws, c->scratch, c->iscratch, NULL, NULL, ops);
		} else {
			MultiLinearSolverSetup_BlockPCG(p->compW_cg_max_iter, p->compW_cg_rate, p->compW_cg_tol,
					p->compW_cg_tol_type, cg_ws, c->scratch, c->iscratch, NULL,
					sigma != 0.0 ? MatDotMultiVecShift : NULL, ops);
		}
	}
	if (p->user_defined_multi_linear_solver == 1) GCGE_SetLinearSolverShift(sigma, c->B);
	/* the column scales (lambda_j + sigma): what BlockPCG finds at the start of its scalar scratch for the "user"
	 * tolerance type (ops_lin_sol.c:186-192); a solver behind flag 1 has no such scratch and reads them here */
	if (p->user_defined_multi_linear_solver == 1) GCGE_SetLinearSolverUserScale(scales, blk);
	if (scaled_rhs) GCGE_SetLinearSolverRhsScale(scales);
	if (p->user_defined_multi_linear_solver == 1) {   /* the work blocks are idle until the orthonormalisation below */
		cg_ws[0] = c->ws0; cg_ws[1] = c->ws1; cg_ws[2] = c->ws2;
		GCGE_SetLinearSolverIdleBlocks(cg_ws, 3);
	}
	ops->MultiLinearSolver(c->A, b, c->V, s, e, ops);
	GCGE_SetLinearSolverIdleBlocks(NULL, 0);
	GCGE_SetLinearSolverUserScale(NULL, 0);
	if (scaled_rhs) GCGE_SetLinearSolverRhsScale(NULL);
	if (p->user_defined_multi_linear_solver == 1) GCGE_SetLinearSolverShift(0.0, NULL);
	if (sigma != 0.0 && c->B != NULL && ops->MatAxpby != NULL
			&& p->user_defined_multi_linear_solver != 1)
		ops->MatAxpby(-sigma, c->B, 1.0, c->A, ops);
	ops->MultiLinearSolver = saved_solver;
	ops->multi_linear_solver_workspace = saved_ws;
	g_timing.linsol += ops->GetWtime() - t1;

	/* B-orthonormalise W against [C X P] and itself; W may shrink */
	setup_orth(c, p->compW_orth_method, p->compW_orth_block_size, p->compW_orth_max_reorth,
			p->compW_orth_zero_tol, c->ws0, ops);
	ops->MultiVecOrth(c->V, c->startW, &c->endW, c->B, ops);
	c->sizeW = c->endW - c->startW;
	g_timing.compW += ops->GetWtime() - t0;
}

/* Second-order variant (-gcge_compW_cg_order 2; reference ComputeW12, ops_eig_sol_gcg.c:697-923): only the first
 * half of the unconverged columns gets search directions, but two of them each: W1 ~ (A + sigma B)^-1 (lambda + sigma) B x
 * started from x, and W2 = the same system solved again starting from W1 (a second leg of the Krylov iteration).
 * The reference refuses a user-defined solver here (:703); ours may be used (it is called twice in the same way). */
static void ComputeW12(Ctx *c, int *offset)
{
	struct OPS_ *ops = c->ops; GCGSolver *p = c->p; void **b = c->ritz;
	int s[2], e[2], idx, blk = 0, i, total = 0, half, pass;
	double sigma = 0.0, *scales = c->scratch, t0 = ops->GetWtime(), t1;
	void (*saved_solver)(void*, void**, void**, int*, int*, struct OPS_*) = ops->MultiLinearSolver;
	void *saved_ws = ops->multi_linear_solver_workspace;
	void **cg_ws[3];
	int use_axpby;
	if (p->compW_cg_auto_shift == 1)
		sigma = -c->ss_eval[c->sizeC] + (c->ss_eval[c->sizeC + 1] - c->ss_eval[c->sizeC]) * 0.01;
	p->sigma = sigma = p->compW_cg_shift + sigma;
	assert(p->compW_cg_auto_shift == 0 || p->user_defined_multi_linear_solver != 2);   /* as in ComputeW */
	use_axpby = sigma != 0.0 && c->B != NULL && ops->MatAxpby != NULL && p->user_defined_multi_linear_solver != 1;

	for (idx = 0; idx < offset[0]; ++idx) total += offset[idx * 2 + 2] - offset[idx * 2 + 1];
	half = total / 2;
	c->startW = c->endP;
	for (pass = 0; pass < 2; ++pass) {
		/* x (first pass only) and the right-hand side (lambda + sigma) B x for the first `half` unconverged columns;
		 * the shift wrapper of BlockPCG uses the rhs block as scratch, so it is rebuilt before the second solve */
		if (pass == 1 && !(sigma != 0.0 && c->B != NULL && !use_axpby && p->user_defined_multi_linear_solver != 1)) break;
		blk = 0;
		for (idx = 0; idx < offset[0] && blk < half; ++idx) {
			int lo = offset[idx * 2 + 1], len = offset[idx * 2 + 2] - lo;
			if (blk + len > half) len = half - blk;
			if (pass == 0) {
				s[0] = lo; e[0] = lo + len; s[1] = c->startW + blk; e[1] = s[1] + len;
				ops->MultiVecAxpby(1.0, c->ritz, 0.0, c->V, s, e, ops);
			}
			s[0] = lo; e[0] = lo + len; s[1] = offset[1] + blk; e[1] = s[1] + len;
			ops->MatDotMultiVec(c->B, c->V, b, s, e, ops);
			for (i = 0; i < len; ++i) scales[blk + i] = c->ss_eval[lo + i] + sigma;
			ops->MultiVecLinearComb(NULL, b, 0, s, e, NULL, 0, scales + blk, 1, ops);
			blk += len;
		}
		if (pass == 0) {
			t1 = ops->GetWtime();
			if (p->user_defined_multi_linear_solver == 0 || p->user_defined_multi_linear_solver == 2) {
				cg_ws[0] = c->ws0; cg_ws[1] = c->ws1; cg_ws[2] = c->ws2;
				g_shift_ctx = c;
				if (use_axpby) ops->MatAxpby(sigma, c->B, 1.0, c->A, ops);
				MultiLinearSolverSetup_BlockPCG(p->compW_cg_max_iter, p->compW_cg_rate, p->compW_cg_tol,
						p->compW_cg_tol_type, cg_ws, c->scratch, c->iscratch, NULL,
						(sigma != 0.0 && !use_axpby) ? MatDotMultiVecShift : NULL, ops);
			}
			if (p->user_defined_multi_linear_solver == 1) { GCGE_SetLinearSolverShift(sigma, c->B); GCGE_SetLinearSolverUserScale(scales, half); }
			s[0] = offset[1]; e[0] = s[0] + half; s[1] = c->startW; e[1] = s[1] + half;
			ops->MultiLinearSolver(c->A, b, c->V, s, e, ops);
			g_timing.linsol += ops->GetWtime() - t1;
			c->endW = c->startW + half;
			/* second leg starts from W1 */
			s[0] = c->startW; e[0] = c->endW; s[1] = c->endW; e[1] = c->endW + half;
			ops->MultiVecAxpby(1.0, c->V, 0.0, c->V, s, e, ops);
		}
	}
	t1 = ops->GetWtime();
	s[0] = offset[1]; e[0] = s[0] + half; s[1] = c->endW; e[1] = s[1] + half;
	ops->MultiLinearSolver(c->A, b, c->V, s, e, ops);
	g_timing.linsol += ops->GetWtime() - t1;
	if (p->user_defined_multi_linear_solver == 1) { GCGE_SetLinearSolverShift(0.0, NULL); GCGE_SetLinearSolverUserScale(NULL, 0); }
	c->endW += half;
	assert(c->endW - c->startW <= total);
	if (use_axpby) ops->MatAxpby(-sigma, c->B, 1.0, c->A, ops);
	ops->MultiLinearSolver = saved_solver;
	ops->multi_linear_solver_workspace = saved_ws;

	setup_orth(c, p->compW_orth_method, p->compW_orth_block_size, p->compW_orth_max_reorth,
			p->compW_orth_zero_tol, c->ws0, ops);
	ops->MultiVecOrth(c->V, c->startW, &c->endW, c->B, ops);
	c->sizeW = c->endW - c->startW;
	g_timing.compW += ops->GetWtime() - t0;
}

static void ComputeRayleighRitz(Ctx *c, int nevConv)
{
	struct OPS_ *ops = c->ops; GCGSolver *p = c->p;
	int N, XP, idx, s[2], e[2], info; double *PtAP = c->scratch, *work;
	double t0 = ops->GetWtime(), t1;
	if (c->sizeP > 0) {   /* P^T A P from the previous projected matrix (host) */
		int Nold = c->sizeV - c->sizeC;
		double *coefP = c->ss_evec + (size_t)Nold * (c->sizeX - c->sizeC);
		ops->DenseMatQtAP('L', 'S', Nold, Nold, c->sizeP, c->sizeP, 1.0, coefP, Nold,
				c->ss_matA, Nold, coefP, Nold, 0.0, PtAP, c->sizeP, PtAP + (size_t)c->sizeP * c->sizeP);
	}
	c->sizeV  = c->sizeX + c->sizeP + c->sizeW;
	c->startN = c->startN + (nevConv - c->sizeC);
	c->endN   = imin(c->endN + (nevConv - c->sizeC), c->endX);
	c->sizeN  = c->endN - c->startN;
	c->sizeC  = nevConv;
	rebase(c);
	N = c->sizeV - c->sizeC; XP = c->sizeX + c->sizeP - c->sizeC;

	t1 = ops->GetWtime();
	if (c->sizeW > 0) {   /* the only O(n) part: V[:, C..W)^T A W */
		double *dst = c->ss_matA + (size_t)N * XP;
		s[0] = c->startN; e[0] = c->endW; s[1] = c->startW; e[1] = c->endW;
		ops->MultiVecQtAP('S', 'N', c->V, c->A, c->V, 0, s, e, dst, N, c->ws0, ops);
		for (idx = 0; idx < c->sizeW; ++idx) {           /* mirror into the W rows */
			int r;
			for (r = 0; r < XP; ++r) c->ss_matA[(size_t)N * r + XP + idx] = dst[(size_t)N * idx + r];
		}
	}
	g_timing.rr_matW += ops->GetWtime() - t1;

	if (c->sizeX == c->sizeV) {   /* start-up: full X^T A X in strips of block_size columns */
		int len = c->sizeX - c->sizeC, bs = imin(p->block_size, len), c0 = c->sizeC;
		while (len > 0) {
			s[0] = c->sizeC; e[0] = c->sizeX; s[1] = c0; e[1] = c0 + bs;
			ops->MultiVecQtAP('S', 'N', c->V, c->A, c->V, 0, s, e,
					c->ss_matA + (size_t)N * (c0 - c->sizeC), N, c->ws0, ops);
			c0 += bs; len -= bs; bs = imin(bs, len);
		}
	} else {
		int r, q;
		for (q = 0; q < XP; ++q) memset(c->ss_matA + (size_t)N * q, 0, XP * sizeof(double));
		for (q = 0; q < c->sizeX - c->sizeC; ++q) c->ss_matA[(size_t)N * q + q] = c->ss_eval[c->sizeC + q];
		for (q = 0; q < c->sizeP; ++q) for (r = 0; r < c->sizeP; ++r)
			c->ss_matA[(size_t)N * (c->sizeX - c->sizeC + q) + (c->sizeX - c->sizeC + r)] = PtAP[(size_t)c->sizeP * q + r];
	}
	for (idx = 0; idx < N; ++idx) c->ss_diag[idx] = c->ss_matA[(size_t)N * idx + idx];
	if (p->compW_cg_shift != 0.0)
		for (idx = 0; idx < N; ++idx) c->ss_matA[(size_t)N * idx + idx] += p->compW_cg_shift;

	t1 = ops->GetWtime();
	work = c->scratch + (size_t)c->sizeP * c->sizeP;
	info = GCGE_SymEigFor((void*)ops->MultiVecLinearComb, 'U', N, c->ss_matA, N, c->ss_eval + c->sizeC, c->ss_evec, N, work);
	assert(info == 0); (void)info;
	g_timing.dsyevx += ops->GetWtime() - t1;

	for (idx = 0; idx < N; ++idx) c->ss_matA[(size_t)N * idx + idx] = c->ss_diag[idx];
	if (p->compW_cg_shift != 0.0)
		for (idx = 0; idx < N; ++idx) c->ss_eval[c->sizeC + idx] -= p->compW_cg_shift;
	g_timing.compRR += ops->GetWtime() - t0;
}

static void GCG(void *A, void *B, double *eval, void **evec, int nevGiven, int *nevConv, struct OPS_ *ops)
{
	GCGSolver *p = (GCGSolver*)ops->eigen_solver_workspace; Ctx ctx, *c = &ctx;
	int nevMax = p->nevMax, b = p->block_size, nevInit = p->nevInit, T = nevMax + 2 * b;
	int nev0, nev, numIterMax = p->numIterMax, numIter, numCheck, idx, *tmp;
	double t_start;
	memset(c, 0, sizeof(*c)); memset(&g_timing, 0, sizeof(g_timing));
	c->ops = ops; c->p = p; c->A = A; c->B = B;
	p->A = A; p->B = B; p->nevGiven = nevGiven; p->nevConv = *nevConv;
	assert(nevInit >= nevGiven);
	assert(nevInit <= nevMax);
	assert(nevInit >= 3 * b || nevInit == nevMax);
	assert(nevMax >= *nevConv + b);
	assert(nevMax <= *nevConv + nevInit);
	assert(p->multiMax <= b);

	c->sizeC = 0; c->sizeN = b; c->sizeX = nevInit; c->sizeP = 0; c->sizeW = 0;
	c->sizeV = c->sizeX; c->startN = 0; c->endN = c->sizeN; c->endX = c->sizeX;
	c->startP = c->endX; c->endP = c->startP; c->startW = c->endP; c->endW = c->startW;
	c->V = p->mv_ws[0]; c->ritz = evec;
	c->ws0 = p->mv_ws[1]; c->ws1 = p->mv_ws[2]; c->ws2 = p->mv_ws[3];
	c->ss_eval = p->dbl_ws;
	for (idx = 0; idx < T; ++idx) c->ss_eval[idx] = 1.0;
	c->ss_diag = c->ss_eval + T;
	rebase(c);
	p->length_dbl_ws = T + 2 * (nevInit + 2 * b) * (nevInit + 2 * b) + 10 * (nevInit + 2 * b) + nevMax * b;
	c->scratch_len = (size_t)4 * T * T + (size_t)64 * T + 1024;
	c->scratch  = (double*)malloc(c->scratch_len * sizeof(double));
	c->iscratch = (int*)malloc((size_t)(8 * T + 64) * sizeof(int));
	c->offsetP = p->int_ws; c->offsetW = c->offsetP + b + 3;
	c->offsetP[0] = 0; c->offsetW[0] = 0;
	t_start = ops->GetWtime();

	InitializeX(c, nevGiven);
	ComputeRayleighRitz(c, 0);
	for (idx = c->sizeV; idx < T; ++idx) c->ss_eval[idx] = c->ss_eval[c->sizeV - 1];
	ComputeRitzVec(c);

	*nevConv = imin(*nevConv, nevMax);
	nev0 = *nevConv; *nevConv = 0;
	nev  = imin(nevInit < nevMax ? 2 * b : nev0, nev0);
	numIter = 0;
	ops->Printf("------------------------------\n");
	ops->Printf("numIter\tnevConv\n");
	do {
		if (numIter <= 0) numCheck = 0;
		else numCheck = (c->startN + c->sizeN < c->endX) ? c->sizeN : (c->endX - c->startN);
		numCheck = imin(numCheck, p->check_conv_max_num);
		*nevConv = CheckConvergence(c, numCheck, c->offsetW);
		ops->Printf("%d\t%d\n", numIter, *nevConv);
		if (*nevConv >= nev) {
			if (*nevConv >= nev0) break;
			{   /* enough pairs locked: absorb P and W into X (only when nevInit < nevMax) */
				int s[2], e[2], N = c->sizeV - c->sizeC;
				nev = imin(nev + c->sizeP + c->sizeW, nev0);
				c->sizeX = imin(c->sizeX + c->sizeP + c->sizeW, nevMax);
				s[0] = c->startN; e[0] = c->endW; s[1] = c->endX; e[1] = c->sizeX;
				ops->MultiVecLinearComb(c->V, c->ritz, 0, s, e,
						c->ss_evec + (size_t)N * (c->endX - c->sizeC), N, NULL, 0, ops);
				c->sizeP = 0; c->sizeW = 0; c->sizeV = c->sizeX;
				c->startP = c->endX; c->endP = c->startP; c->startW = c->endP; c->endW = c->startW;
				c->endX = c->sizeX;
				c->endN = imin(c->startN + b, c->endX); c->sizeN = c->endN - c->startN;
				numIterMax -= numIter; numIter = 0;
			}
		}
		if (numIter == 0) { c->sizeP = 0; c->startP = c->endX; c->endP = c->startP; }
		else ComputeP(c, c->offsetP);
		ComputeX(c);
		if (p->compW_cg_order != 1) ComputeW12(c, c->offsetW);
		else ComputeW(c, c->offsetW);
		tmp = c->offsetP; c->offsetP = c->offsetW; c->offsetW = tmp;
		ComputeRayleighRitz(c, *nevConv);
		for (idx = c->sizeV; idx < T; ++idx) c->ss_eval[idx] = c->ss_eval[c->sizeV - 1];
		ComputeRitzVec(c);
		++numIter;
	} while (numIter < numIterMax);

	p->numIter = numIter + (p->numIterMax - numIterMax);
	p->sizeV = c->sizeV;
	memcpy(eval, c->ss_eval, c->sizeX * sizeof(double));
	g_timing.total = ops->GetWtime() - t_start;
	{
		const GCGE_Timing *t = &g_timing; double it = p->numIter > 0 ? p->numIter : 1;
		ops->Printf("|--GCG----------------------------\n");
		ops->Printf("|Total Time = %.2f, Avg Time per Iteration = %.2f\n", t->total, t->total / it);
		ops->Printf("|checkconv   compP   compRR   (rr_matW   dsyevx)   compRV   compW   (linsol)   compX   initX\n");
		ops->Printf("|%.2f\t%.2f\t%.2f\t(%.2f\t%.2f)\t%.2f\t%.2f\t(%.2f)\t%.2f\t%.2f\n",
				t->checkconv, t->compP, t->compRR, t->rr_matW, t->dsyevx, t->compRV,
				t->compW, t->linsol, t->compX, t->initX);
		ops->Printf("|--GCG----------------------------\n");
	}
	free(c->scratch); free(c->iscratch);
}

/* ------------------------------------------------------------ setup / parameters */
void EigenSolverSetup_GCG(int multiMax, double gapMin, int nevInit, int nevMax, int block_size,
		double tol[2], int numIterMax, int user_defined_multi_linear_solver,
		void **mv_ws[4], double *dbl_ws, int *int_ws, struct OPS_ *ops)
{
	static GCGSolver g;
	memset(&g, 0, sizeof(g));
	g.nevMax = nevMax; g.multiMax = multiMax; g.gapMin = gapMin;
	g.nevInit = nevInit; g.nevGiven = 0; g.block_size = block_size;
	g.tol[0] = tol[0]; g.tol[1] = tol[1]; g.numIterMax = numIterMax;
	g.user_defined_multi_linear_solver = user_defined_multi_linear_solver;
	g.mv_ws[0] = mv_ws[0]; g.mv_ws[1] = mv_ws[1]; g.mv_ws[2] = mv_ws[2]; g.mv_ws[3] = mv_ws[3];
	g.dbl_ws = dbl_ws; g.int_ws = int_ws;
	/* algorithm defaults (reference :1569-1598) */
	strcpy(g.initX_orth_method, "mgs"); g.initX_orth_block_size = -1; g.initX_orth_max_reorth = 1; g.initX_orth_zero_tol = 1e-14;
	strcpy(g.compP_orth_method, "mgs"); g.compP_orth_block_size = -1; g.compP_orth_max_reorth = 1; g.compP_orth_zero_tol = 1e-14;
	strcpy(g.compW_orth_method, "mgs"); g.compW_orth_block_size = -1; g.compW_orth_max_reorth = 1; g.compW_orth_zero_tol = 1e-14;
	g.compW_cg_max_iter = 40; g.compW_cg_rate = 1e-2; g.compW_cg_tol = 1e-8; strcpy(g.compW_cg_tol_type, "abs");
	g.compW_cg_auto_shift = 0; g.compW_cg_shift = 0.0; g.compW_cg_order = 1;
	g.compRR_min_gap = gapMin; g.compRR_min_num = -1; g.compRR_tol = 1e-16;
	g.check_conv_max_num = block_size;
	ops->eigen_solver_workspace = (void*)&g;
	ops->EigenSolver = GCG;
}

void EigenSolverCreateWorkspace_GCG(int nevInit, int nevMax, int block_size, void *mat,
		void ***mv_ws, double **dbl_ws, int **int_ws, struct OPS_ *ops)
{
	int T = nevMax + 2 * block_size, sizeV = nevInit + 2 * block_size, i;
	assert(mv_ws != NULL);
	ops->MultiVecCreateByMat(&mv_ws[0], T, mat, ops);
	ops->MultiVecSetRandomValue(mv_ws[0], 0, T, ops);
	for (i = 1; i < 4; ++i) {
		ops->MultiVecCreateByMat(&mv_ws[i], block_size, mat, ops);
		ops->MultiVecSetRandomValue(mv_ws[i], 0, block_size, ops);
	}
	if (dbl_ws != NULL) *dbl_ws = (double*)calloc((size_t)2 * sizeV * sizeV + 10 * sizeV + T + (size_t)nevMax * block_size, sizeof(double));
	if (int_ws != NULL) *int_ws = (int*)calloc((size_t)6 * sizeV + 2 * (block_size + 3), sizeof(int));
}

void EigenSolverDestroyWorkspace_GCG(int nevInit, int nevMax, int block_size, void *mat,
		void ***mv_ws, double **dbl_ws, int **int_ws, struct OPS_ *ops)
{
	int i;
	assert(mv_ws != NULL);
	ops->MultiVecDestroy(&mv_ws[0], nevMax + 2 * block_size, ops);
	for (i = 1; i < 4; ++i) ops->MultiVecDestroy(&mv_ws[i], block_size, ops);
	if (dbl_ws != NULL) { free(*dbl_ws); *dbl_ws = NULL; }
	if (int_ws != NULL) { free(*int_ws); *int_ws = NULL; }
}

/* negative / NULL arguments keep the current value (reference :1678-1735) */
void EigenSolverSetParameters_GCG(int check_conv_max_num,
		const char *initX_orth_method, int initX_orth_block_size, int initX_orth_max_reorth, double initX_orth_zero_tol,
		const char *compP_orth_method, int compP_orth_block_size, int compP_orth_max_reorth, double compP_orth_zero_tol,
		const char *compW_orth_method, int compW_orth_block_size, int compW_orth_max_reorth, double compW_orth_zero_tol,
		int compW_cg_max_iter, double compW_cg_rate, double compW_cg_tol, const char *compW_cg_tol_type,
		int compW_cg_auto_shift, int compRR_min_num, double compRR_min_gap, double compRR_tol,
		struct OPS_ *ops)
{
	GCGSolver *g = (GCGSolver*)ops->eigen_solver_workspace;
	if (check_conv_max_num > 0) g->check_conv_max_num = check_conv_max_num;
#define SET_ORTH(PH)                                                                \
	if (PH##_orth_method != NULL) { strncpy(g->PH##_orth_method, PH##_orth_method, 7); g->PH##_orth_method[7] = 0; } \
	if (PH##_orth_block_size > 0)  g->PH##_orth_block_size = PH##_orth_block_size;  \
	if (PH##_orth_max_reorth >= 0) g->PH##_orth_max_reorth = PH##_orth_max_reorth;  \
	if (PH##_orth_zero_tol > 0)    g->PH##_orth_zero_tol   = PH##_orth_zero_tol;
	SET_ORTH(initX) SET_ORTH(compP) SET_ORTH(compW)
#undef SET_ORTH
	if (compW_cg_max_iter > 0) g->compW_cg_max_iter = compW_cg_max_iter;
	if (compW_cg_rate > 0)     g->compW_cg_rate = compW_cg_rate;
	if (compW_cg_tol > 0)      g->compW_cg_tol = compW_cg_tol;
	if (compW_cg_tol_type != NULL) { strncpy(g->compW_cg_tol_type, compW_cg_tol_type, 7); g->compW_cg_tol_type[7] = 0; }
	g->compW_cg_auto_shift = compW_cg_auto_shift;
	if (compRR_min_gap > 0) g->compRR_min_gap = compRR_min_gap;
	if (compRR_min_num > 0) g->compRR_min_num = compRR_min_num;
	if (compRR_tol > 0)     g->compRR_tol = compRR_tol;
}

void EigenSolverSetParametersFromCommandLine_GCG(int argc, char *argv[], struct OPS_ *ops)
{
	GCGSolver *g = (GCGSolver*)ops->eigen_solver_workspace;
	struct { const char *name; char type; void *ptr; } opt[] = {
		{"-gcge_max_multi", 'i', &g->multiMax}, {"-gcge_min_gap", 'f', &g->gapMin},
		{"-gcge_given_nevec", 'i', &g->nevGiven}, {"-gcge_max_niter", 'i', &g->numIterMax},
		{"-gcge_abs_tol", 'f', &g->tol[0]}, {"-gcge_rel_tol", 'f', &g->tol[1]},
		{"-gcge_user_defined_multi_lin_sol", 'i', &g->user_defined_multi_linear_solver},
		{"-gcge_initX_orth_method", 's', g->initX_orth_method},
		{"-gcge_initX_orth_block_size", 'i', &g->initX_orth_block_size},
		{"-gcge_initX_orth_max_reorth", 'i', &g->initX_orth_max_reorth},
		{"-gcge_initX_orth_zero_tol", 'f', &g->initX_orth_zero_tol},
		{"-gcge_check_conv_max_num", 'i', &g->check_conv_max_num},
		{"-gcge_compP_orth_method", 's', g->compP_orth_method},
		{"-gcge_compP_orth_block_size", 'i', &g->compP_orth_block_size},
		{"-gcge_compP_orth_max_reorth", 'i', &g->compP_orth_max_reorth},
		{"-gcge_compP_orth_zero_tol", 'f', &g->compP_orth_zero_tol},
		{"-gcge_compW_orth_method", 's', g->compW_orth_method},
		{"-gcge_compW_orth_block_size", 'i', &g->compW_orth_block_size},
		{"-gcge_compW_orth_max_reorth", 'i', &g->compW_orth_max_reorth},
		{"-gcge_compW_orth_zero_tol", 'f', &g->compW_orth_zero_tol},
		{"-gcge_compW_cg_max_iter", 'i', &g->compW_cg_max_iter},
		{"-gcge_compW_cg_rate", 'f', &g->compW_cg_rate}, {"-gcge_compW_cg_tol", 'f', &g->compW_cg_tol},
		{"-gcge_compW_cg_tol_type", 's', g->compW_cg_tol_type},
		{"-gcge_compW_cg_auto_shift", 'i', &g->compW_cg_auto_shift},
		{"-gcge_compW_cg_shift", 'f', &g->compW_cg_shift}, {"-gcge_compW_cg_order", 'i', &g->compW_cg_order},
		{"-gcge_compRR_min_num", 'i', &g->compRR_min_num},
		/* the reference parses min_gap with type 'i' into a double and never matches compRR_tol
		 * (trailing blanks in the option name, :1804-1806); both are read as doubles here */
		{"-gcge_compRR_min_gap", 'f', &g->compRR_min_gap}, {"-gcge_compRR_tol", 'f', &g->compRR_tol},
	};
	int i, print_usage = 1;
	for (i = 0; i < (int)(sizeof(opt) / sizeof(opt[0])); ++i)
		ops->GetOptionFromCommandLine(opt[i].name, opt[i].type, opt[i].ptr, argc, argv, ops);
	ops->GetOptionFromCommandLine("-gcge_print_usage", 'i', &print_usage, argc, argv, ops);
	if (print_usage) {
		ops->Printf("\nUsage: %s [<options>]   (-gcge_<name> <value>)\n", argc > 0 ? argv[0] : "gcge");
		ops->Printf(" max_multi %d  min_gap %.2e  max_niter %d  given_nevec %d  abs_tol %.2e  rel_tol %.2e\n",
				g->multiMax, g->gapMin, g->numIterMax, g->nevGiven, g->tol[0], g->tol[1]);
		ops->Printf(" user_defined_multi_lin_sol %d  check_conv_max_num %d\n",
				g->user_defined_multi_linear_solver, g->check_conv_max_num);
		ops->Printf(" initX_orth: %s block %d reorth %d zero_tol %.2e\n", g->initX_orth_method,
				g->initX_orth_block_size, g->initX_orth_max_reorth, g->initX_orth_zero_tol);
		ops->Printf(" compP_orth: %s block %d reorth %d zero_tol %.2e\n", g->compP_orth_method,
				g->compP_orth_block_size, g->compP_orth_max_reorth, g->compP_orth_zero_tol);
		ops->Printf(" compW_orth: %s block %d reorth %d zero_tol %.2e\n", g->compW_orth_method,
				g->compW_orth_block_size, g->compW_orth_max_reorth, g->compW_orth_zero_tol);
		ops->Printf(" compW_cg: max_iter %d rate %.2e tol %.2e type %s order %d auto_shift %d shift %.2e\n",
				g->compW_cg_max_iter, g->compW_cg_rate, g->compW_cg_tol, g->compW_cg_tol_type,
				g->compW_cg_order, g->compW_cg_auto_shift, g->compW_cg_shift);
		ops->Printf(" compRR: min_num %d min_gap %.2e tol %.2e\n", g->compRR_min_num, g->compRR_min_gap, g->compRR_tol);
	}
}
