/* Block B-orthonormalisation against the operator table.
 *
 *   MultiVecOrth(x, start, &end, B):  columns [0,start) of x are B-orthonormal;
 *   make [start,end) B-orthonormal and B-orthogonal to them, drop numerically
 *   dependent columns, compact the survivors to the front and return the new end.
 *
 * Two schemes, selected by the Setup call (semantics of the reference's
 * src/ops_orth.c; restated, not copied):
 *   - block modified Gram–Schmidt   (ModifiedGramSchmidt :203-393, OrthSelf :45-118)
 *   - recursive-halving Gram–Schmidt (BinaryGramSchmidt :517-617, OrthBinary :415-515,
 *     leaf = Gram matrix + symmetric eigensolve, OrthSelfEVP :122-201)
 * Only ops->MultiVecQtAP / MultiVecLinearComb / MultiVecAxpby touch O(n) data, so
 * the same code drives the HIP back-end, the CPU oracle and the host dense table.
 */
#include <assert.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gcge_solver.h"

static double max_abs(const double *v, int n)
{
	double m = 0.0; int i;
	for (i = 0; i < n; ++i) if (fabs(v[i]) > m) m = fabs(v[i]);
	return m;
}
static void negate(double *v, int n) { int i; for (i = 0; i < n; ++i) v[i] = -v[i]; }

/* x[:, s1:e1) -= x[:, s0:e0) * (x[:, s0:e0)^T B x[:, s1:e1)), repeated up to 1+max_reorth
 * times.  Returns nothing; coef scratch needs (e0-s0)*(e1-s1) doubles.
 * lazy = 0: the reference's order (ops_orth.c:235-265: apply the update, then test its size) — what every
 * caller uses;
 * lazy = 1: a re-orthogonalisation pass whose coefficients are already below reorth_tol is not applied
 * (saves one n x k x m panel update per call).  NOT safe in general: the test is absolute, and when the
 * columns being projected are tiny (W blocks close to convergence are ~1e-9 of the basis vectors) a
 * coefficient of 1e-14 is a relative 1e-5 that the subsequent normalisation blows up — an 8-rank run on an
 * 8^3 grid stagnated at 7 of 8 pairs with it (tests/test_dist.py).
 * lazy = 2 (the Cholesky-QR scheme, B == NULL): every pass is applied as in the reference, but whether ANOTHER pass
 * follows is decided relative to the columns ("twice is enough", Kahan / Parlett): after pass p >= 1, if
 * ||c_j|| <= 0.1 ||w_j|| for every column (c_j this pass's coefficients, w_j the column as it entered the pass), what
 * the pass removed was a small fraction of the column, the update was computed without cancellation and the result is
 * orthogonal to the basis to rounding — a further pass would remove O(eps ||w_j||).  The reference's absolute test
 * (|c| < 50 eps) asks for a third Gram product + panel update whenever the second pass's coefficients exceed 1e-14 in
 * absolute terms, which they do for columns of norm ~1.  Costs one column-norm sweep after the first pass. */
static void project_out(void **x, int s0, int e0, int s1, int e1, void *B,
		int max_reorth, double reorth_tol, int lazy, void **mv_ws, double *coef, struct OPS_ *ops)
{
	int pass, start[2], end[2], k = e0 - s0, m = e1 - s1, i, j; double one = 1.0;
	const int kahan = lazy == 2 && B == NULL && getenv("GCGE_ORTH_ABSOLUTE_TEST") == NULL;
	double *wn = NULL;   /* squared norms of the columns as they enter the next pass */
	if (k <= 0 || m <= 0) return;
	if (kahan) wn = (double*)malloc((size_t)m * sizeof(double));   /* (not carved from coef: callers size that as k x m) */
	for (pass = 0; pass < 1 + max_reorth; ++pass) {
		start[0] = s0; end[0] = e0; start[1] = s1; end[1] = e1;
		ops->MultiVecQtAP('S', 'N', x, B, x, 0, start, end, coef, k, mv_ws, ops);
		if (lazy == 1 && pass > 0 && max_abs(coef, k * m) < reorth_tol) break;
		negate(coef, k * m);
		ops->MultiVecLinearComb(x, x, 0, start, end, coef, k, &one, 0, ops);
		if (max_abs(coef, k * m) < reorth_tol) break;
		if (kahan && pass + 1 < 1 + max_reorth) {
			if (pass >= 1) {                 /* was this pass a small correction for every column? */
				int small = 1;
				for (j = 0; j < m && small; ++j) {
					double c2 = 0.0;
					for (i = 0; i < k; ++i) c2 += coef[(size_t)k * j + i] * coef[(size_t)k * j + i];
					if (!(c2 <= 0.01 * wn[j])) small = 0;     /* also catches wn = 0 and NaN */
				}
				if (small) break;
			}
			/* the columns as the next pass will see them */
			start[0] = s1; end[0] = e1; start[1] = s1; end[1] = e1;
			ops->MultiVecInnerProd('D', x, x, 0, start, end, wn, 1, ops);
		}
	}
	free(wn);
}

/* Column-by-column modified Gram–Schmidt inside x[:, start:*end) in the B inner product. */
static void orth_self(void **x, int start_x, int *end_x, void *B, int max_reorth,
		double zero_tol, double reorth_tol, void **mv_ws, double *ws, struct OPS_ *ops)
{
	int k, start[2], end[2], pass; double one = 1.0;
	for (k = start_x; k < *end_x; ++k) {
		int tail = *end_x - (k + 1);
		double *g = ws;                    /* g[0] = x_k^T B x_k, g[1..] = x_j^T B x_k, j > k */
		double nrm, inv;
		start[0] = k; end[0] = *end_x; start[1] = k; end[1] = k + 1;
		ops->MultiVecQtAP('S', 'N', x, B, x, 0, start, end, g, *end_x - k, mv_ws, ops);
		nrm = sqrt(g[0]);
		if (!(nrm >= zero_tol)) {          /* dependent (or NaN): drop, pull the last column in */
			ops->Printf("r_[%d] = %6.4e\n", k, nrm);
			if (k < *end_x - 1) {
				start[0] = *end_x - 1; end[0] = *end_x; start[1] = k; end[1] = k + 1;
				ops->MultiVecAxpby(1.0, x, 0.0, x, start, end, ops);
			}
			--(*end_x); --k;
			continue;
		}
		inv = 1.0 / nrm;
		start[0] = k; end[0] = k + 1; start[1] = k; end[1] = k + 1;
		ops->MultiVecAxpby(0.0, NULL, inv, x, start, end, ops);          /* x_k /= ||x_k||_B */
		if (tail <= 0) continue;
		{
			int i; double *c = g + 1;
			for (i = 0; i < tail; ++i) c[i] *= -inv;                     /* -(q_k^T B x_j) */
			start[0] = k; end[0] = k + 1; start[1] = k + 1; end[1] = *end_x;
			ops->MultiVecLinearComb(x, x, 0, start, end, c, 1, &one, 0, ops);
			for (pass = 1; pass < max_reorth - 1; ++pass) {
				start[0] = k + 1; end[0] = *end_x; start[1] = k; end[1] = k + 1;
				ops->MultiVecQtAP('S', 'N', x, B, x, 0, start, end, c, tail, mv_ws, ops);
				negate(c, tail);
				start[0] = k; end[0] = k + 1; start[1] = k + 1; end[1] = *end_x;
				ops->MultiVecLinearComb(x, x, 0, start, end, c, 1, &one, 0, ops);
				if (max_abs(c, tail) < reorth_tol) break;
			}
		}
	}
}

/* Project the freshly orthonormalised block [b0,b1) out of the columns [b1,e) that follow
 * it.  On the first pass B*block is formed in mv_ws (side effect of QtAP); later passes
 * re-use it ('T' output keeps the coefficient matrix in the layout LinearComb wants). */
static void project_block_from_rest(void **x, int b0, int b1, int e, void *B,
		int max_reorth, double reorth_tol, int lazy, void **mv_ws, double *coef, struct OPS_ *ops)
{
	int pass, start[2], end[2], nb = b1 - b0, rem = e - b1; double one = 1.0;
	if (nb <= 0 || rem <= 0) return;
	for (pass = 0; pass < 1 + max_reorth; ++pass) {
		if (B != NULL && pass > 0) {
			start[0] = b1; end[0] = e; start[1] = 0; end[1] = nb;
			ops->MultiVecQtAP('S', 'T', x, NULL, mv_ws, 0, start, end, coef, nb, mv_ws, ops);
		} else {
			start[0] = b1; end[0] = e; start[1] = b0; end[1] = b1;
			ops->MultiVecQtAP('S', 'T', x, B, x, 0, start, end, coef, nb, mv_ws, ops);
		}
		if (lazy && pass > 0 && max_abs(coef, nb * rem) < reorth_tol) break;
		negate(coef, nb * rem);
		start[0] = b0; end[0] = b1; start[1] = b1; end[1] = e;
		ops->MultiVecLinearComb(x, x, 0, start, end, coef, nb, &one, 0, ops);
		if (max_abs(coef, nb * rem) < reorth_tol) break;
	}
}

static void ModifiedGramSchmidt(void **x, int start_x, int *end_x, void *B, struct OPS_ *ops)
{
	ModifiedGramSchmidtOrth *p = (ModifiedGramSchmidtOrth*)ops->orth_workspace;
	double *coef = p->dbl_ws;
	int block, b0, b1, start[2], end[2];
	if (*end_x <= start_x) return;
	project_out(x, 0, start_x, start_x, *end_x, B, p->max_reorth, p->reorth_tol, 0, p->mv_ws, coef, ops);

	b0 = start_x;
	block = p->block_size;
	if (block <= 0) block = ((*end_x - b0) / 2 > 2) ? (*end_x - b0) / 2 : 2;
	if (block > *end_x - b0) block = *end_x - b0;
	while (block > 0) {
		int dropped, refill;
		b1 = b0 + block;
		orth_self(x, b0, &b1, B, p->max_reorth, p->orth_zero_tol, p->reorth_tol, p->mv_ws, coef, ops);
		dropped = block - (b1 - b0);
		refill  = *end_x - (b0 + block);            /* columns not yet visited */
		if (refill > dropped) refill = dropped;
		if (refill > 0) {                          /* move tail columns into the holes */
			start[0] = *end_x - refill; end[0] = *end_x; start[1] = b1; end[1] = b1 + refill;
			ops->MultiVecAxpby(1.0, x, 0.0, x, start, end, ops);
		}
		*end_x -= dropped;
		if (b1 < *end_x && b0 < b1)
			project_block_from_rest(x, b0, b1, *end_x, B, p->max_reorth, p->reorth_tol, 0, p->mv_ws, coef, ops);
		b0 = b1;
		if (block > *end_x - b0) block = *end_x - b0;
	}
}

void MultiVecOrthSetup_ModifiedGramSchmidt(int block_size, int max_reorth, double orth_zero_tol,
		void **mv_ws, double *dbl_ws, struct OPS_ *ops)
{
	static ModifiedGramSchmidtOrth mgs;
	mgs.block_size = block_size; mgs.max_reorth = max_reorth;
	mgs.orth_zero_tol = orth_zero_tol; mgs.reorth_tol = 50 * DBL_EPSILON;
	mgs.mv_ws = mv_ws; mgs.dbl_ws = dbl_ws;
	ops->orth_workspace = (void*)&mgs;
	ops->MultiVecOrth   = ModifiedGramSchmidt;
}

/* ---------------------------------------------------------------- Cholesky-QR scheme
 * Same contract and the same outer structure as the block MGS above (project out the
 * leading columns, then sweep the new columns block by block, dropping dependent ones and
 * projecting every finished block out of the rest), but a block is orthonormalised among
 * itself with BLOCK operations only:  G = X^T B X  ->  G = R^T R  ->  X <- X R^-1, repeated
 * until G = I to rounding (Cholesky-QR2/3).  In exact arithmetic the result equals the
 * Gram-Schmidt columns (QR is unique); on a GPU it replaces ~4 single-column passes per
 * column (orth_self) by 2-3 Gram + panel-update passes per block.
 * Rank decision: a column whose Cholesky pivot falls below chol_drop^2 of its squared norm
 * in the first pass is dependent (relative 1e-7, the resolution of a Gram matrix in FP64;
 * orth_self decides at orth_zero_tol on the absolute norm).  If a factorisation breaks down
 * later the block falls back to orth_self. */
static int chol_upper(int m, const double *G, int ldg, double *R, double drop2, int first_pass, int *bad)
{
	/* R upper triangular (column-major, ld m) with G = R^T R; returns 0, or 1 with *bad = column */
	int i, j, k;
	for (j = 0; j < m; ++j) {
		double piv = G[(size_t)ldg * j + j];
		for (i = 0; i < j; ++i) {
			double v = G[(size_t)ldg * j + i];
			for (k = 0; k < i; ++k) v -= R[(size_t)m * i + k] * R[(size_t)m * j + k];
			v /= R[(size_t)m * i + i];
			R[(size_t)m * j + i] = v;
			piv -= v * v;
		}
		if (!(piv > (first_pass ? drop2 * G[(size_t)ldg * j + j] : 0.0)) || !(G[(size_t)ldg * j + j] > 0.0)) { *bad = j; return 1; }
		R[(size_t)m * j + j] = sqrt(piv);
		for (i = j + 1; i < m; ++i) R[(size_t)m * j + i] = 0.0;
	}
	return 0;
}
static void invert_upper(int m, const double *R, double *Ri)
{
	int i, j, k;
	for (j = 0; j < m; ++j) {
		for (i = 0; i < m; ++i) Ri[(size_t)m * j + i] = 0.0;
		Ri[(size_t)m * j + j] = 1.0 / R[(size_t)m * j + j];
		for (i = j - 1; i >= 0; --i) {
			double v = 0.0;
			for (k = i + 1; k <= j; ++k) v += R[(size_t)m * k + i] * Ri[(size_t)m * j + k];
			Ri[(size_t)m * j + i] = -v / R[(size_t)m * i + i];
		}
	}
}
/* orthonormalise x[:, b0:*b1) among themselves; dependent columns are replaced by the block's
 * last column and *b1 shrinks.  ws: 3 m^2 doubles. */
static void orth_self_chol(void **x, int b0, int *b1, void *B, int max_reorth, double zero_tol,
		double reorth_tol, void **mv_ws, double *ws, struct OPS_ *ops)
{
	const double drop = 1e-7;
	int pass = 0, start[2], end[2];
	while (*b1 > b0 && pass < 4) {
		int m = *b1 - b0, bad = -1, i, j; double dev = 0.0;
		double *G = ws, *R = G + (size_t)m * m, *Ri = R + (size_t)m * m;
		start[0] = b0; end[0] = *b1; start[1] = b0; end[1] = *b1;
		ops->MultiVecQtAP('S', 'S', x, B, x, 0, start, end, G, m, mv_ws, ops);
		for (j = 0; j < m; ++j) for (i = 0; i < m; ++i) {
			double d = fabs(G[(size_t)m * j + i] - (i == j ? 1.0 : 0.0));
			if (d > dev) dev = d;
		}
		if (dev < reorth_tol) return;                       /* already B-orthonormal */
		if (chol_upper(m, G, m, R, drop * drop, pass == 0, &bad)) {
			if (pass == 0) {                                 /* dependent column: pull the last one in, retry */
				ops->Printf("chol: column %d dependent (pivot/norm^2 below %.1e)\n", b0 + bad, drop * drop);
				if (b0 + bad < *b1 - 1) {
					start[0] = *b1 - 1; end[0] = *b1; start[1] = b0 + bad; end[1] = b0 + bad + 1;
					ops->MultiVecAxpby(1.0, x, 0.0, x, start, end, ops);
				}
				--(*b1);
				continue;
			}
			/* breakdown after a successful pass: finish with the column-wise scheme */
			orth_self(x, b0, b1, B, max_reorth, zero_tol, reorth_tol, mv_ws, ws, ops);
			return;
		}
		invert_upper(m, R, Ri);
		if (m <= GCGE_InplaceLinearCombCols((void*)ops->MultiVecLinearComb)) {       /* X = X R^-1 row by row, in place */
			start[0] = b0; end[0] = *b1; start[1] = b0; end[1] = *b1;
			ops->MultiVecLinearComb(x, x, 0, start, end, Ri, m, NULL, 0, ops);
		} else {
			start[0] = b0; end[0] = *b1; start[1] = 0; end[1] = m;
			ops->MultiVecLinearComb(x, mv_ws, 0, start, end, Ri, m, NULL, 0, ops);      /* ws = X R^-1 */
			start[0] = 0; end[0] = m; start[1] = b0; end[1] = *b1;
			ops->MultiVecAxpby(1.0, mv_ws, 0.0, x, start, end, ops);
		}
		++pass;
		if (dev < 1e-8) return;   /* deviation after this pass is O(dev^2 + eps) */
	}
}

static void CholeskyQR(void **x, int start_x, int *end_x, void *B, struct OPS_ *ops)
{
	ModifiedGramSchmidtOrth *p = (ModifiedGramSchmidtOrth*)ops->orth_workspace;
	double *coef = p->dbl_ws;
	int block, b0, b1, start[2], end[2];
	if (*end_x <= start_x) return;
	project_out(x, 0, start_x, start_x, *end_x, B, p->max_reorth, p->reorth_tol, 2, p->mv_ws, coef, ops);
	b0 = start_x;
	block = p->block_size;
	if (block <= 0) block = *end_x - b0;
	if (block > *end_x - b0) block = *end_x - b0;
	while (block > 0) {
		int dropped, refill;
		b1 = b0 + block;
		orth_self_chol(x, b0, &b1, B, p->max_reorth, p->orth_zero_tol, p->reorth_tol, p->mv_ws, coef, ops);
		dropped = block - (b1 - b0);
		refill  = *end_x - (b0 + block);
		if (refill > dropped) refill = dropped;
		if (refill > 0) {
			start[0] = *end_x - refill; end[0] = *end_x; start[1] = b1; end[1] = b1 + refill;
			ops->MultiVecAxpby(1.0, x, 0.0, x, start, end, ops);
		}
		*end_x -= dropped;
		if (b1 < *end_x && b0 < b1)
			project_block_from_rest(x, b0, b1, *end_x, B, p->max_reorth, p->reorth_tol, 0, p->mv_ws, coef, ops);
		b0 = b1;
		if (block > *end_x - b0) block = *end_x - b0;
	}
}

void MultiVecOrthSetup_CholeskyQR(int block_size, int max_reorth, double orth_zero_tol,
		void **mv_ws, double *dbl_ws, struct OPS_ *ops)
{
	static ModifiedGramSchmidtOrth cqr;
	cqr.block_size = block_size; cqr.max_reorth = max_reorth;
	cqr.orth_zero_tol = orth_zero_tol; cqr.reorth_tol = 50 * DBL_EPSILON;
	cqr.mv_ws = mv_ws; cqr.dbl_ws = dbl_ws;
	ops->orth_workspace = (void*)&cqr;
	ops->MultiVecOrth   = CholeskyQR;
}

/* ---------------------------------------------------------------- binary scheme */
/* Leaf: G = X^T B X, G = U diag(w) U^T, X <- X U diag(w^-1/2) for w above the threshold
 * (a threshold on the SQUARED norm, unlike orth_self).  ws: N*N + 3N doubles. */
static void orth_self_evp(void **x, int start_x, int *end_x, void *B, int max_reorth,
		double zero_tol, double reorth_tol, void **mv_ws, double *ws, struct OPS_ *ops)
{
	int pass, start[2], end[2];
	for (pass = 0; pass < 1 + max_reorth; ++pass) {
		int N = *end_x - start_x, k, dep = 0, info; double sum = 0.0;
		double *G, *U, *w, *work;
		if (N <= 0) return;
		G = ws; U = G + (size_t)N * N; w = U + (size_t)N * N; work = w + N;
		start[0] = start_x; end[0] = *end_x; start[1] = start_x; end[1] = *end_x;
		ops->MultiVecQtAP('S', 'S', x, B, x, 0, start, end, G, N, mv_ws, ops);
		info = GCGE_SymEigFor((void*)ops->MultiVecLinearComb, 'L', N, G, N, w, U, N, work);
		assert(info == 0); (void)info;
		for (k = 0; k < N; ++k) {
			assert(w[k] > -zero_tol);
			if (fabs(w[k]) > zero_tol) w[k] = 1.0 / sqrt(w[k]);
			else ++dep;
		}
		if (dep > 0) ops->Printf("There has %d linear dependent vec\n", dep);
		start[0] = start_x; end[0] = *end_x; start[1] = 0; end[1] = N - dep;
		ops->MultiVecLinearComb(x, mv_ws, 0, start, end, U + (size_t)N * dep, N, NULL, 0, ops);
		ops->MultiVecLinearComb(NULL, mv_ws, 0, start, end, NULL, 0, w + dep, 1, ops);
		*end_x -= dep;
		start[0] = 0; end[0] = N - dep; start[1] = start_x; end[1] = *end_x;
		ops->MultiVecAxpby(1.0, mv_ws, 0.0, x, start, end, ops);
		for (k = 0; k < N; ++k) sum += fabs(w[k]);
		if (dep == 0 && fabs(sum - N) < reorth_tol) break;
	}
}

static void orth_binary(void **x, int start_x, int *end_x, void *B, char leaf, int block_size,
		int max_reorth, double zero_tol, double reorth_tol, void **mv_ws, double *ws, struct OPS_ *ops)
{
	int ncols = *end_x - start_x, s0, e0, s1, e1, dropped, move, start[2], end[2];
	if (ncols <= 0) return;
	if (ncols <= block_size) {
		if (leaf == 'E') orth_self_evp(x, start_x, end_x, B, max_reorth, zero_tol, reorth_tol, mv_ws, ws, ops);
		else orth_self(x, start_x, end_x, B, max_reorth, zero_tol, reorth_tol, mv_ws, ws, ops);
		return;
	}
	s0 = start_x; e0 = start_x + ncols / 2; s1 = e0; e1 = *end_x;
	orth_binary(x, s0, &e0, B, leaf, block_size, max_reorth, zero_tol, reorth_tol, mv_ws, ws, ops);
	{   /* second half minus the span of the first; 'T' layout + B*X0 cached in mv_ws */
		int pass, nb = e0 - s0, rem = e1 - s1; double one = 1.0;
		for (pass = 0; pass < 1 + max_reorth && nb > 0 && rem > 0; ++pass) {
			if (B != NULL && pass > 0) {
				start[0] = s1; end[0] = e1; start[1] = 0; end[1] = nb;
				ops->MultiVecQtAP('S', 'T', x, NULL, mv_ws, 0, start, end, ws, nb, mv_ws, ops);
			} else {
				start[0] = s1; end[0] = e1; start[1] = s0; end[1] = e0;
				ops->MultiVecQtAP('S', 'T', x, B, x, 0, start, end, ws, nb, mv_ws, ops);
			}
			negate(ws, nb * rem);
			start[0] = s0; end[0] = e0; start[1] = s1; end[1] = e1;
			ops->MultiVecLinearComb(x, x, 0, start, end, ws, nb, &one, 0, ops);
			if (max_abs(ws, nb * rem) < reorth_tol) break;
		}
	}
	orth_binary(x, s1, &e1, B, leaf, block_size, max_reorth, zero_tol, reorth_tol, mv_ws, ws, ops);
	dropped = s1 - e0;                     /* holes left by the first half */
	*end_x  = e1 - dropped;
	move    = dropped < e1 - s1 ? dropped : e1 - s1;
	if (move > 0) {
		start[0] = e1 - move; end[0] = e1; start[1] = e0; end[1] = e0 + move;
		ops->MultiVecAxpby(1.0, x, 0.0, x, start, end, ops);
	}
}

static void BinaryGramSchmidt(void **x, int start_x, int *end_x, void *B, struct OPS_ *ops)
{
	BinaryGramSchmidtOrth *p = (BinaryGramSchmidtOrth*)ops->orth_workspace;
	int n, block = p->block_size; char leaf;
	if (*end_x <= start_x) return;
	project_out(x, 0, start_x, start_x, *end_x, B, p->max_reorth, p->reorth_tol, 0, p->mv_ws, p->dbl_ws, ops);
	n = *end_x - start_x;
	if (n < 16) { if (block <= 0) block = 4; leaf = 'M'; }
	else { if (block <= 0 || block > n / 4) block = n / 4; leaf = 'E'; }
	orth_binary(x, start_x, end_x, B, leaf, block, p->max_reorth, p->orth_zero_tol, p->reorth_tol,
			p->mv_ws, p->dbl_ws, ops);
}

void MultiVecOrthSetup_BinaryGramSchmidt(int block_size, int max_reorth, double orth_zero_tol,
		void **mv_ws, double *dbl_ws, struct OPS_ *ops)
{
	static BinaryGramSchmidtOrth bgs;
	bgs.block_size = block_size; bgs.max_reorth = max_reorth;
	bgs.orth_zero_tol = orth_zero_tol; bgs.reorth_tol = 50 * DBL_EPSILON;
	bgs.mv_ws = mv_ws; bgs.dbl_ws = dbl_ws;
	ops->orth_workspace = (void*)&bgs;
	ops->MultiVecOrth   = BinaryGramSchmidt;
}
