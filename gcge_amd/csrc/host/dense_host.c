/* Host dense back-end: column-major blocks in host memory (GCGE_DENSE).
 *
 * This is the table the solver reaches through ops->lapack_ops for its SMALL
 * matrices — the (V-C) x b coefficient block orthonormalised in ComputeP
 * (reference ops_eig_sol_gcg.c:373-414), P^T A P in the Rayleigh–Ritz step
 * (:936-949) — and it doubles as a complete OPS back-end for dense test problems.
 * Behaviour follows app/app_lapack.c (DenseMatQtAP :24-227, MultiVecAxpby :334-395,
 * MultiVecLinearComb :463-534, MultiVecQtAP :535-584, MultiVecSetRandomValue
 * :322-333, DenseMatOrth :653-699); BLAS/LAPACK calls are replaced by plain C loops
 * (no BLAS is guaranteed on the target box) and dgeqp3/dorgqr by an in-file
 * Householder QR with column pivoting.
 */
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gcge_ops.h"

typedef GCGE_DENSE DM;

static double dotn(int n, const double *x, const double *y)
{
	double s0 = 0, s1 = 0, s2 = 0, s3 = 0; int i = 0;
	for (; i + 4 <= n; i += 4) {
		s0 += x[i] * y[i]; s1 += x[i+1] * y[i+1]; s2 += x[i+2] * y[i+2]; s3 += x[i+3] * y[i+3];
	}
	for (; i < n; ++i) s0 += x[i] * y[i];
	return (s0 + s1) + (s2 + s3);
}
static void axpyn(int n, double a, const double *x, double *y)
{
	int i; for (i = 0; i < n; ++i) y[i] += a * x[i];
}
static void scaln(int n, double b, double *y)
{
	int i;
	if (b == 0.0) memset(y, 0, (size_t)n * sizeof(double));
	else if (b != 1.0) for (i = 0; i < n; ++i) y[i] *= b;
}

/* C = alpha Q^T A P + beta C   (host, column-major; see header comment) */
static void DenseMatQtAP(char ntluA, char nsdC, int nrowsA, int ncolsA, int nrowsC, int ncolsC,
		double alpha, double *Q, int ldQ, double *A, int ldA, double *P, int ldP,
		double beta, double *C, int ldC, double *dbl_ws)
{
	int i, j;
	if (nrowsC == 0 || ncolsC == 0) return;
	if (nrowsA == 0 && ncolsA == 0) {             /* empty inner dimension: C = 0 */
		if (nsdC == 'D') for (j = 0; j < ncolsC; ++j) C[(size_t)ldC * j] = 0.0;
		else for (j = 0; j < ncolsC; ++j) memset(C + (size_t)ldC * j, 0, nrowsC * sizeof(double));
		return;
	}
	if (A == NULL) {
		assert(nrowsA == ncolsA);
		if (nsdC == 'D') {
			assert(nrowsC == ncolsC);
			for (j = 0; j < ncolsC; ++j) {
				double v = (alpha != 0.0) ? alpha * dotn(nrowsA, Q + (size_t)ldQ * j, P + (size_t)ldP * j) : 0.0;
				double *c = C + (size_t)ldC * j;
				*c = (beta == 0.0) ? v : v + beta * (*c);
			}
		} else if (nsdC == 'S') {                  /* lower triangle, then mirror */
			assert(nrowsC == ncolsC);
			for (j = 0; j < ncolsC; ++j) {
				for (i = j; i < nrowsC; ++i) {
					double v = alpha * dotn(nrowsA, Q + (size_t)ldQ * i, P + (size_t)ldP * j);
					double *c = C + (size_t)ldC * j + i;
					*c = (beta == 0.0) ? v : v + beta * (*c);
				}
				for (i = j + 1; i < nrowsC; ++i) C[(size_t)ldC * i + j] = C[(size_t)ldC * j + i];
			}
		} else {
			for (j = 0; j < ncolsC; ++j) for (i = 0; i < nrowsC; ++i) {
				double v = alpha * dotn(nrowsA, Q + (size_t)ldQ * i, P + (size_t)ldP * j);
				double *c = C + (size_t)ldC * j + i;
				*c = (beta == 0.0) ? v : v + beta * (*c);
			}
		}
		return;
	}
	/* W = op(A) P into dbl_ws (nrowsA x ncolsC), then C = alpha Q^T W + beta C */
	{
		double *W = dbl_ws; int ldW = nrowsA, k;
		for (j = 0; j < ncolsC; ++j) {
			double *w = W + (size_t)ldW * j; const double *p = P + (size_t)ldP * j;
			memset(w, 0, nrowsA * sizeof(double));
			if (ntluA == 'L' || ntluA == 'U') {      /* symmetric A, one triangle stored */
				for (k = 0; k < nrowsA; ++k) {
					const double *ak = A + (size_t)ldA * k; double s = 0.0;
					if (ntluA == 'L') {
						for (i = k + 1; i < nrowsA; ++i) { w[i] += ak[i] * p[k]; s += ak[i] * p[i]; }
					} else {
						for (i = 0; i < k; ++i) { w[i] += ak[i] * p[k]; s += ak[i] * p[i]; }
					}
					w[k] += ak[k] * p[k] + s;
				}
			} else if (ntluA == 'T') {
				for (i = 0; i < nrowsA; ++i) w[i] = dotn(ncolsA, A + (size_t)ldA * i, p);
			} else {
				for (k = 0; k < ncolsA; ++k) axpyn(nrowsA, p[k], A + (size_t)ldA * k, w);
			}
		}
		DenseMatQtAP(ntluA, nsdC, nrowsA, nrowsA, nrowsC, ncolsC, alpha, Q, ldQ, NULL, ldA,
				W, ldW, beta, C, ldC, NULL);
	}
}

/* ------------------------------------------------------------ multivectors */
static void mv_alloc(DM **v, int nrows, int ncols, int ldd)
{
	*v = (DM*)malloc(sizeof(DM));
	(*v)->nrows = nrows; (*v)->ncols = ncols; (*v)->ldd = ldd;
	(*v)->data = (double*)calloc((size_t)ldd * (ncols > 0 ? ncols : 1), sizeof(double));
}
static void D_MultiVecCreateByMat(void ***mv, int num_vec, void *mat, struct OPS_ *ops)
{ DM *m = (DM*)mat; mv_alloc((DM**)mv, m->ncols, num_vec, m->ncols); }
static void D_MultiVecCreateByVec(void ***mv, int num_vec, void *vec, struct OPS_ *ops)
{ DM *s = (DM*)vec; mv_alloc((DM**)mv, s->nrows, num_vec, s->ldd); }
static void D_MultiVecCreateByMultiVec(void ***mv, int num_vec, void **src, struct OPS_ *ops)
{ DM *s = (DM*)src; mv_alloc((DM**)mv, s->nrows, num_vec, s->ldd); }
static void D_MultiVecDestroy(void ***mv, int num_vec, struct OPS_ *ops)
{ DM *v = *(DM**)mv; if (v) { free(v->data); free(v); } *mv = NULL; }
static void D_GetVecFromMultiVec(void **mv, int col, void **vec, struct OPS_ *ops)
{
	DM *m = (DM*)mv, *v = (DM*)malloc(sizeof(DM));
	v->nrows = m->nrows; v->ncols = 1; v->ldd = m->ldd; v->data = m->data + (size_t)m->ldd * col;
	*vec = v;
}
static void D_RestoreVecForMultiVec(void **mv, int col, void **vec, struct OPS_ *ops)
{ free(*vec); *vec = NULL; }
static void D_MultiVecView(void **x, int start, int end, struct OPS_ *ops)
{
	DM *v = (DM*)x; int r, c;
	for (r = 0; r < v->nrows; ++r) {
		for (c = start; c < end; ++c) ops->Printf("%6.4e\t", v->data[(size_t)v->ldd * c + r]);
		ops->Printf("\n");
	}
}
static void D_MultiVecLocalInnerProd(char nsdIP, void **x, void **y, int is_vec,
		int *start, int *end, double *ip, int ldIP, struct OPS_ *ops)
{
	DM *vx = (DM*)x, *vy = (DM*)y; int k = end[0] - start[0], m = end[1] - start[1];
	if (k <= 0 || m <= 0) return;
	DenseMatQtAP('S', nsdIP, vx->nrows, vy->nrows, k, m, 1.0,
			vx->data + (size_t)vx->ldd * start[0], vx->ldd, NULL, 0,
			vy->data + (size_t)vy->ldd * start[1], vy->ldd, 0.0, ip, ldIP, NULL);
}
static void D_MultiVecSetRandomValue(void **x, int start, int end, struct OPS_ *ops)
{
	DM *v = (DM*)x; int r, c;
	for (c = start; c < end; ++c) {
		double *d = v->data + (size_t)v->ldd * c;
		for (r = 0; r < v->nrows; ++r) d[r] = ((double)rand()) / ((double)RAND_MAX + 1);
	}
}
static void D_MultiVecAxpby(double alpha, void **x, double beta, void **y,
		int *start, int *end, struct OPS_ *ops)
{
	DM *vx = (DM*)x, *vy = (DM*)y; int c, m = end[1] - start[1];
	assert(end[0] - start[0] == m);
	if (m <= 0 || vy->nrows == 0) return;
	for (c = 0; c < m; ++c) {
		double *d = vy->data + (size_t)vy->ldd * (start[1] + c);
		scaln(vy->nrows, beta, d);
		if (vx != NULL) axpyn(vy->nrows, alpha, vx->data + (size_t)vx->ldd * (start[0] + c), d);
	}
}
static void D_MultiVecLinearComb(void **x, void **y, int is_vec, int *start, int *end,
		double *coef, int ldc, double *beta, int incb, struct OPS_ *ops)
{
	DM *vx = (DM*)x, *vy = (DM*)y; int k = end[0] - start[0], m = end[1] - start[1], i, j;
	if (k == 0 || m == 0 || vy->nrows == 0) return;
	if (vx != NULL && coef != NULL && vx->data == vy->data) {
		/* in-place (disjoint column ranges): results must not feed back — same as dgemm on
		 * disjoint panels, which holds because source and destination columns differ */
	}
	for (j = 0; j < m; ++j) {
		double *d = vy->data + (size_t)vy->ldd * (start[1] + j);
		double b = (beta == NULL) ? 0.0 : (incb == 0 ? *beta : beta[(size_t)j * incb]);
		if (vx != NULL && coef != NULL) {
			scaln(vy->nrows, b, d);
			for (i = 0; i < k; ++i)
				axpyn(vy->nrows, coef[(size_t)ldc * j + i], vx->data + (size_t)vx->ldd * (start[0] + i), d);
		} else if (beta != NULL) {
			if (b != 1.0) { int r; for (r = 0; r < vy->nrows; ++r) d[r] *= b; }
		}
	}
}
static void D_MatDotMultiVec(void *mat, void **x, void **y, int *start, int *end, struct OPS_ *ops)
{
	DM *A = (DM*)mat, *vx = (DM*)x, *vy = (DM*)y; int c, k, m = end[1] - start[1];
	assert(end[0] - start[0] == m);
	for (c = 0; c < m; ++c) {
		const double *xs = vx->data + (size_t)vx->ldd * (start[0] + c);
		double *yd = vy->data + (size_t)vy->ldd * (start[1] + c);
		if (A == NULL) { memcpy(yd, xs, vy->nrows * sizeof(double)); continue; }
		memset(yd, 0, vy->nrows * sizeof(double));
		for (k = 0; k < A->ncols; ++k) axpyn(A->nrows, xs[k], A->data + (size_t)A->ldd * k, yd);
	}
}
static void D_MatTransDotMultiVec(void *mat, void **x, void **y, int *start, int *end, struct OPS_ *ops)
{
	DM *A = (DM*)mat, *vx = (DM*)x, *vy = (DM*)y; int c, k, m = end[1] - start[1];
	for (c = 0; c < m; ++c) {
		const double *xs = vx->data + (size_t)vx->ldd * (start[0] + c);
		double *yd = vy->data + (size_t)vy->ldd * (start[1] + c);
		for (k = 0; k < A->ncols; ++k) yd[k] = dotn(A->nrows, A->data + (size_t)A->ldd * k, xs);
	}
}
static void D_MultiVecQtAP(char ntsA, char ntsd, void **mvQ, void *matA, void **mvP, int is_vec,
		int *start, int *end, double *qAp, int ldQAP, void **mv_ws, struct OPS_ *ops)
{
	DM *q = (DM*)mvQ, *p = (DM*)mvP, *A = (DM*)matA, *ws = (DM*)mv_ws;
	int k = end[0] - start[0], m = end[1] - start[1], i, j;
	if (k <= 0 || m <= 0) return;
	if (ntsA == 'S') ntsA = 'L';
	if (ntsd == 'T') {
		double *tmp = (double*)malloc((size_t)k * m * sizeof(double));
		DenseMatQtAP(ntsA, 'N', q->nrows, p->nrows, k, m, 1.0, q->data + (size_t)q->ldd * start[0], q->ldd,
				A ? A->data : NULL, A ? A->ldd : 0, p->data + (size_t)p->ldd * start[1], p->ldd,
				0.0, tmp, k, ws ? ws->data : NULL);
		for (i = 0; i < k; ++i) for (j = 0; j < m; ++j) qAp[(size_t)ldQAP * i + j] = tmp[(size_t)k * j + i];
		free(tmp);
	} else {
		DenseMatQtAP(ntsA, ntsd, q->nrows, p->nrows, k, m, 1.0, q->data + (size_t)q->ldd * start[0], q->ldd,
				A ? A->data : NULL, A ? A->ldd : 0, p->data + (size_t)p->ldd * start[1], p->ldd,
				0.0, qAp, ldQAP, ws ? ws->data : NULL);
	}
}

/* ------------------------------------------------------------ DenseMatOrth
 * Columns [0,start) of mat are orthonormal.  Remove their span from columns
 * [start,*end) (two passes), then QR with column pivoting; the numerical rank is
 * the number of leading |r_ii| > tol; on exit columns [start, new end) hold an
 * orthonormal basis.  dbl_ws: >= start*(n) + 3n doubles; int_ws unused here. */
static void DenseMatOrth(double *mat, int nrows, int ldm, int start, int *end,
		double orth_zero_tol, double *dbl_ws, int length, int *int_ws)
{
	int n = *end - start, m = nrows, i, j, k, pass, rank;
	double *a = mat + (size_t)ldm * start;
	if (n <= 0) return;
	for (pass = 0; pass < 2 && start > 0; ++pass) {
		for (j = 0; j < n; ++j) {
			double *c = dbl_ws;            /* start coefficients for this column */
			for (i = 0; i < start; ++i) c[i] = dotn(m, mat + (size_t)ldm * i, a + (size_t)ldm * j);
			for (i = 0; i < start; ++i) axpyn(m, -c[i], mat + (size_t)ldm * i, a + (size_t)ldm * j);
		}
	}
	{
		int kmax = m < n ? m : n;
		double *tau = dbl_ws, *nrm = dbl_ws + n, *w = dbl_ws + 2 * n;
		assert(length >= 3 * n + m);
		for (j = 0; j < n; ++j) nrm[j] = sqrt(dotn(m, a + (size_t)ldm * j, a + (size_t)ldm * j));
		for (k = 0; k < kmax; ++k) {
			int piv = k; double *ak, alpha, beta, t;
			for (j = k + 1; j < n; ++j) if (nrm[j] > nrm[piv]) piv = j;
			if (piv != k) {
				double *p1 = a + (size_t)ldm * k, *p2 = a + (size_t)ldm * piv;
				for (i = 0; i < m; ++i) { t = p1[i]; p1[i] = p2[i]; p2[i] = t; }
				t = nrm[k]; nrm[k] = nrm[piv]; nrm[piv] = t;
			}
			ak = a + (size_t)ldm * k;
			alpha = ak[k];
			beta = sqrt(dotn(m - k, ak + k, ak + k));
			if (beta == 0.0) { tau[k] = 0.0; continue; }
			if (alpha > 0) beta = -beta;
			tau[k] = (beta - alpha) / beta;
			t = 1.0 / (alpha - beta);
			for (i = k + 1; i < m; ++i) ak[i] *= t;      /* v = [1; ak[k+1:]] */
			ak[k] = beta;                                 /* r_kk */
			for (j = k + 1; j < n; ++j) {                 /* apply H_k to the trailing columns */
				double *aj = a + (size_t)ldm * j, s = aj[k];
				for (i = k + 1; i < m; ++i) s += ak[i] * aj[i];
				s *= tau[k];
				aj[k] -= s;
				for (i = k + 1; i < m; ++i) aj[i] -= s * ak[i];
				nrm[j] = sqrt(dotn(m - k - 1, aj + k + 1, aj + k + 1));
			}
		}
		for (rank = kmax; rank > 0; --rank)
			if (fabs(a[(size_t)ldm * (rank - 1) + (rank - 1)]) > orth_zero_tol) break;
		/* form the first `rank` columns of Q = H_0 ... H_{kmax-1}: backward accumulation */
		for (j = rank - 1; j >= 0; --j) {
			double *qj = a + (size_t)ldm * j;
			/* w = e_j, apply H_j..H_0?  (Q e_j = H_0 H_1 ... H_j e_j since H_k e_j = e_j for k > j) */
			for (i = 0; i < m; ++i) w[i] = 0.0;
			w[j] = 1.0;
			for (k = j; k >= 0; --k) {
				double *ak = a + (size_t)ldm * k, s;
				if (tau[k] == 0.0) continue;
				s = w[k];
				for (i = k + 1; i < m; ++i) s += ak[i] * w[i];
				s *= tau[k];
				w[k] -= s;
				for (i = k + 1; i < m; ++i) w[i] -= s * ak[i];
			}
			/* column j's reflector vector is no longer needed by columns < j ... but IS by them:
			 * columns k < j use only reflectors 0..k, so overwriting column j is safe */
			memcpy(qj, w, m * sizeof(double));
		}
		*end = start + rank;
	}
}

/* ------------------------------------------------------------ single vectors */
static void D_VecCreateByMat(void **v, void *mat, struct OPS_ *ops) { D_MultiVecCreateByMat((void***)v, 1, mat, ops); }
static void D_VecCreateByVec(void **v, void *src, struct OPS_ *ops) { D_MultiVecCreateByVec((void***)v, 1, src, ops); }
static void D_VecDestroy(void **v, struct OPS_ *ops) { D_MultiVecDestroy((void***)v, 1, ops); }
static void D_VecView(void *x, struct OPS_ *ops) { D_MultiVecView((void**)x, 0, 1, ops); }
static void D_VecInnerProd(void *x, void *y, double *ip, struct OPS_ *ops)
{ int s[2] = {0,0}, e[2] = {1,1}; D_MultiVecLocalInnerProd('S', (void**)x, (void**)y, 0, s, e, ip, 1, ops); }
static void D_VecSetRandomValue(void *x, struct OPS_ *ops) { D_MultiVecSetRandomValue((void**)x, 0, 1, ops); }
static void D_VecAxpby(double a, void *x, double b, void *y, struct OPS_ *ops)
{ int s[2] = {0,0}, e[2] = {1,1}; D_MultiVecAxpby(a, (void**)x, b, (void**)y, s, e, ops); }
static void D_MatDotVec(void *mat, void *x, void *y, struct OPS_ *ops)
{ int s[2] = {0,0}, e[2] = {1,1}; D_MatDotMultiVec(mat, (void**)x, (void**)y, s, e, ops); }
static void D_MatTransDotVec(void *mat, void *x, void *y, struct OPS_ *ops)
{ int s[2] = {0,0}, e[2] = {1,1}; D_MatTransDotMultiVec(mat, (void**)x, (void**)y, s, e, ops); }
static void D_MatView(void *mat, struct OPS_ *ops)
{ DM *m = (DM*)mat; D_MultiVecView((void**)mat, 0, m->ncols, ops); }

/* ------------------------------------------------------------ multigrid (dense toy)
 * app_lapack.c:863-955: a fixed 1-D hierarchy for testing the multigrid machinery on dense matrices — level l + 1 has
 * (rows - 1) / 2 rows, P_l is linear interpolation (1 at row 2c + 1, 1/2 at its two neighbours), A_{l+1} = P_l^T A_l P_l (B alike). */
static DM *dm_new(int nrows, int ncols)
{
	DM *m = (DM*)malloc(sizeof(DM));
	m->nrows = nrows; m->ncols = ncols; m->ldd = nrows;
	m->data = (double*)calloc((size_t)nrows * (ncols > 0 ? ncols : 1), sizeof(double));
	return m;
}
static void dm_ptap(const DM *P, const DM *A, DM *C)      /* C = P^T A P */
{
	const int nf = P->nrows, nc = P->ncols; int i, j, k;
	double *t = (double*)calloc((size_t)nf * nc, sizeof(double));
	for (j = 0; j < nc; ++j)
		for (k = 0; k < nf; ++k) {
			const double pkj = P->data[(size_t)P->ldd * j + k];
			if (pkj != 0.0) axpyn(nf, pkj, A->data + (size_t)A->ldd * k, t + (size_t)nf * j);
		}
	for (j = 0; j < nc; ++j)
		for (i = 0; i < nc; ++i) C->data[(size_t)C->ldd * j + i] = dotn(nf, P->data + (size_t)P->ldd * i, t + (size_t)nf * j);
	free(t);
}
static void D_MultiGridCreate(void ***A_array, void ***B_array, void ***P_array, int *num_levels, void *A, void *B, struct OPS_ *ops)
{
	int level, nrows = ((DM*)A)->nrows, ncols = (nrows - 1) / 2, col;
	ops->Printf("Just a test, P is fixed\n");
	*A_array = (void**)calloc(*num_levels, sizeof(void*));
	*P_array = (void**)calloc(*num_levels > 1 ? *num_levels - 1 : 1, sizeof(void*));
	if (B != NULL) *B_array = (void**)calloc(*num_levels, sizeof(void*));
	(*A_array)[0] = A;
	if (B != NULL) (*B_array)[0] = B;
	for (level = 1; level < *num_levels; ++level) {
		DM *P = dm_new(nrows, ncols), *Ac = dm_new(ncols, ncols);
		for (col = 0; col < ncols; ++col) {
			P->data[(size_t)nrows * col + 2 * col + 1] = 1.0;
			P->data[(size_t)nrows * col + 2 * col] = 0.5;
			P->data[(size_t)nrows * col + 2 * col + 2] = 0.5;
		}
		dm_ptap(P, (DM*)(*A_array)[level - 1], Ac);
		(*P_array)[level - 1] = P; (*A_array)[level] = Ac;
		if (B != NULL) { DM *Bc = dm_new(ncols, ncols); dm_ptap(P, (DM*)(*B_array)[level - 1], Bc); (*B_array)[level] = Bc; }
		nrows = ncols; ncols = (nrows - 1) / 2;
	}
}
static void D_MultiGridDestroy(void ***A_array, void ***B_array, void ***P_array, int *num_levels, struct OPS_ *ops)
{
	int level;
	for (level = 1; level < *num_levels; ++level) {
		DM *a = (DM*)(*A_array)[level], *p = (DM*)(*P_array)[level - 1];
		free(a->data); free(a); free(p->data); free(p);
		if (B_array != NULL && *B_array != NULL) { DM *b = (DM*)(*B_array)[level]; free(b->data); free(b); }
	}
	free(*A_array); *A_array = NULL;
	free(*P_array); *P_array = NULL;
	if (B_array != NULL && *B_array != NULL) { free(*B_array); *B_array = NULL; }
}

void OPS_DENSE_Set(struct OPS_ *ops)
{
	ops->Printf                   = DefaultPrintf;
	ops->GetWtime                 = DefaultGetWtime;
	ops->GetOptionFromCommandLine = DefaultGetOptionFromCommandLine;
	ops->MatView                  = D_MatView;
	ops->VecCreateByMat           = D_VecCreateByMat;
	ops->VecCreateByVec           = D_VecCreateByVec;
	ops->VecDestroy               = D_VecDestroy;
	ops->VecView                  = D_VecView;
	ops->VecInnerProd             = D_VecInnerProd;
	ops->VecLocalInnerProd        = D_VecInnerProd;
	ops->VecSetRandomValue        = D_VecSetRandomValue;
	ops->VecAxpby                 = D_VecAxpby;
	ops->MatDotVec                = D_MatDotVec;
	ops->MatTransDotVec           = D_MatTransDotVec;
	ops->MultiVecCreateByMat      = D_MultiVecCreateByMat;
	ops->MultiVecCreateByVec      = D_MultiVecCreateByVec;
	ops->MultiVecCreateByMultiVec = D_MultiVecCreateByMultiVec;
	ops->MultiVecDestroy          = D_MultiVecDestroy;
	ops->GetVecFromMultiVec       = D_GetVecFromMultiVec;
	ops->RestoreVecForMultiVec    = D_RestoreVecForMultiVec;
	ops->MultiVecView             = D_MultiVecView;
	ops->MultiVecLocalInnerProd   = D_MultiVecLocalInnerProd;
	ops->MultiVecInnerProd        = D_MultiVecLocalInnerProd;   /* host data is replicated: no reduction */
	ops->MultiVecSetRandomValue   = D_MultiVecSetRandomValue;
	ops->MultiVecAxpby            = D_MultiVecAxpby;
	ops->MultiVecLinearComb       = D_MultiVecLinearComb;
	ops->MatDotMultiVec           = D_MatDotMultiVec;
	ops->MatTransDotMultiVec      = D_MatTransDotMultiVec;
	ops->MultiVecQtAP             = D_MultiVecQtAP;
	ops->DenseMatQtAP             = DenseMatQtAP;
	ops->DenseMatOrth             = DenseMatOrth;
	ops->MultiGridCreate          = D_MultiGridCreate;
	ops->MultiGridDestroy         = D_MultiGridDestroy;
}
