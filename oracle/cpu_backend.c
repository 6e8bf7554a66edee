/* TEST INFRASTRUCTURE — the CPU oracle.  Never linked into, imported by or called
 * from the product path (gcge_amd/); only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it.
 *
 * Plain-C restatement of the reference's built-in CPU back-end for the GCG hot
 * path: the CCS sparse matrix with column-major dense blocks of vectors
 *   app/app_ccs.c     MultiVecCreateByMat :40-49, MatDotMultiVec :50-139 (scatter
 *                     form, OpenMP over the block columns :117-131), OPS_CCS_Set :213-249
 *   app/app_lapack.c  DenseMatQtAP (matA == NULL branch) :64-183, MultiVecLocalInnerProd
 *                     :299-313, MultiVecSetRandomValue :322-333, MultiVecAxpby :334-395,
 *                     MultiVecLinearComb :463-534
 *   src/ops_multi_vec.c  DefaultMultiVecQtAP :351-411 (A P staged in mv_ws)
 * BLAS calls are written out as loops (deliberately naive: this is the checker).
 *
 * Parity pin: tests/test_oracle_golden.py compares every slot and whole GCG runs
 * with the real reference compiled under oracle/_ref (here, where /root/reference
 * exists) and tests/golden/ holds the vectors produced by it for the GPU box.
 *
 * Struct layouts equal the reference's LAPACKVEC (app_lapack.h:17-20) and CCSMAT
 * (app_ccs.h:20-24) on purpose: either solver stack can drive either back-end.
 */
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "gcge_ops.h"
#include "gcge_multigrid.h"
#include "oracle.h"

typedef ORACLE_VEC VEC;
typedef ORACLE_CCS CCS;

static int g_threads = 1;
void oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int  oracle_get_threads(void) { return g_threads; }

/* Row-partitioned variant (one rank per slab; the counterpart of the reference's MPI back-ends,
 * app/app_slepc.c:49-60, app/app_phg.c:292-359): a slab is stored as CSR with LOCAL column indices,
 * nrows owned rows and ncols = nrows + nghost; blocks of vectors carry nghost extra rows (ldd) that
 * are filled by the halo exchange callback right before a product. */
static int is_dist(const CCS *m) { return m != NULL && m->ncols > m->nrows; }
static ORACLE_HALO g_halo;    /* one partition per process */
void oracle_set_halo(const ORACLE_HALO *h) { g_halo = *h; }

/* app_ccs.c:40-49 — zero-filled n x num_vec block, ldd = nrows = mat->ncols */
static void O_MultiVecCreateByMat(void ***mv, int num_vec, void *mat, struct OPS_ *ops)
{
	VEC *v = (VEC*)malloc(sizeof(VEC)); CCS *m = (CCS*)mat;
	v->nrows = is_dist(m) ? m->nrows : m->ncols; v->ncols = num_vec; v->ldd = m->ncols;
	v->data = (double*)calloc((size_t)v->ldd * (num_vec > 0 ? num_vec : 1), sizeof(double));
	*mv = (void**)v;
}
static void O_MultiVecCreateByMultiVec(void ***mv, int num_vec, void **src, struct OPS_ *ops)
{
	VEC *v = (VEC*)malloc(sizeof(VEC)), *s = (VEC*)src;
	v->nrows = s->nrows; v->ncols = num_vec; v->ldd = s->ldd;
	v->data = (double*)calloc((size_t)v->ldd * (num_vec > 0 ? num_vec : 1), sizeof(double));
	*mv = (void**)v;
}
/* app_lapack.c:260-268 */
static void O_MultiVecDestroy(void ***mv, int num_vec, struct OPS_ *ops)
{
	VEC *v = *(VEC**)mv;
	if (v != NULL) { free(v->data); free(v); }
	*mv = NULL;
}
static void O_MultiVecView(void **x, int start, int end, struct OPS_ *ops)
{
	VEC *v = (VEC*)x; int r, c;
	for (r = 0; r < v->nrows; ++r) {
		for (c = start; c < end; ++c) ops->Printf("%6.4e\t", v->data[(size_t)v->ldd * c + r]);
		ops->Printf("\n");
	}
}
/* row partition of this rank (row-partitioned runs only): lets every rank draw the GLOBAL rand() sequence and
 * keep its own rows, so the start block does not depend on the number of ranks (identical local streams would
 * make the global block periodic in the slab index and starve the solver of whole families of modes) */
static long g_part_begin = 0, g_part_nglobal = 0;
void oracle_set_partition(long row_begin, long nglobal) { g_part_begin = row_begin; g_part_nglobal = nglobal; }
/* app_lapack.c:322-333 — glibc rand(), columns outer, rows inner */
static void O_MultiVecSetRandomValue(void **x, int start, int end, struct OPS_ *ops)
{
	VEC *v = (VEC*)x; int r, c; long k;
	const long before = g_part_nglobal > 0 ? g_part_begin : 0;
	const long after = g_part_nglobal > 0 ? g_part_nglobal - g_part_begin - v->nrows : 0;
	for (c = start; c < end; ++c) {
		for (k = 0; k < before; ++k) (void)rand();
		for (r = 0; r < v->nrows; ++r)
			v->data[(size_t)v->ldd * c + r] = ((double)rand()) / ((double)RAND_MAX + 1);
		for (k = 0; k < after; ++k) (void)rand();
	}
}
/* app_lapack.c:334-395 — beta == 0 zeroes y (never multiplies), x == NULL scales only */
static void O_MultiVecAxpby(double alpha, void **x, double beta, void **y,
		int *start, int *end, struct OPS_ *ops)
{
	VEC *vx = (VEC*)x, *vy = (VEC*)y; int c, m = end[1] - start[1];
	assert(end[0] - start[0] == m);
	if (m <= 0 || vy->nrows == 0) return;
#pragma omp parallel for schedule(static) num_threads(g_threads)
	for (c = 0; c < m; ++c) {
		double *d = vy->data + (size_t)vy->ldd * (start[1] + c); int r;
		if (beta == 0.0) for (r = 0; r < vy->nrows; ++r) d[r] = 0.0;
		else if (beta != 1.0) for (r = 0; r < vy->nrows; ++r) d[r] *= beta;
		if (vx != NULL) {
			const double *s = vx->data + (size_t)vx->ldd * (start[0] + c);
			for (r = 0; r < vy->nrows; ++r) d[r] += alpha * s[r];
		}
	}
}
/* app_lapack.c:463-534 — y_j = sum_i x_i coef(i,j) + beta_j y_j */
static void O_MultiVecLinearComb(void **x, void **y, int is_vec, int *start, int *end,
		double *coef, int ldc, double *beta, int incb, struct OPS_ *ops)
{
	VEC *vx = (VEC*)x, *vy = (VEC*)y; int k = end[0] - start[0], m = end[1] - start[1], j;
	if (k == 0 || m == 0 || vy->nrows == 0) return;
#pragma omp parallel for schedule(static) num_threads(g_threads)
	for (j = 0; j < m; ++j) {
		double *d = vy->data + (size_t)vy->ldd * (start[1] + j); int r, i;
		double g = (beta == NULL) ? 0.0 : (incb == 0 ? *beta : beta[(size_t)incb * j]);
		if (beta != NULL && g != 1.0) for (r = 0; r < vy->nrows; ++r) d[r] *= g;
		if (vx != NULL && coef != NULL) {
			if (beta == NULL) for (r = 0; r < vy->nrows; ++r) d[r] = 0.0;   /* dgemm with beta = 0 */
			for (i = 0; i < k; ++i) {
				const double *s = vx->data + (size_t)vx->ldd * (start[0] + i);
				double cij = coef[(size_t)ldc * j + i];
				for (r = 0; r < vy->nrows; ++r) d[r] += cij * s[r];
			}
		}
	}
}
/* app_lapack.c:299-313 -> DenseMatQtAP :64-183 with matA == NULL */
static void O_MultiVecLocalInnerProd(char nsd, void **x, void **y, int is_vec,
		int *start, int *end, double *ip, int ldIP, struct OPS_ *ops)
{
	VEC *vx = (VEC*)x, *vy = (VEC*)y; int k = end[0] - start[0], m = end[1] - start[1], j;
	if (k <= 0 || m <= 0) return;
	assert(vx->nrows == vy->nrows);
	if (nsd == 'D' || nsd == 'S') assert(k == m);
#pragma omp parallel for schedule(static) num_threads(g_threads)
	for (j = 0; j < m; ++j) {
		const double *pj = vy->data + (size_t)vy->ldd * (start[1] + j); int i, r;
		int i0 = (nsd == 'D') ? j : (nsd == 'S' ? j : 0), i1 = (nsd == 'D') ? j + 1 : k;
		for (i = i0; i < i1; ++i) {
			const double *qi = vx->data + (size_t)vx->ldd * (start[0] + i); double s = 0.0;
			for (r = 0; r < vx->nrows; ++r) s += qi[r] * pj[r];
			if (nsd == 'D') ip[(size_t)ldIP * j] = s;
			else ip[(size_t)ldIP * j + i] = s;
		}
	}
	if (nsd == 'S')   /* lower triangle computed, mirrored (dcopy at :127-129) */
		for (j = 0; j < m; ++j) { int i; for (i = j + 1; i < k; ++i) ip[(size_t)ldIP * i + j] = ip[(size_t)ldIP * j + i]; }
}
/* app_ccs.c:50-139 — y = 0, then for each column j of A: y[i_row] += a * x[j]
 * (CCS scatter; equals A x for the symmetric matrices GCGE handles); mat == NULL copies */
static void O_MatDotMultiVec(void *mat, void **x, void **y, int *start, int *end, struct OPS_ *ops)
{
	CCS *A = (CCS*)mat; VEC *vx = (VEC*)x, *vy = (VEC*)y; int c, m = end[0] - start[0];
	assert(m == end[1] - start[1]);
	if (m <= 0) return;
	if (is_dist(A)) {   /* slab: fetch halo rows, then row-wise (gather) product with local indices */
		int nloc = A->nrows, ng = A->ncols - A->nrows, i, j, k;
		double *sb = (double*)malloc((size_t)(g_halo.nsend > 0 ? g_halo.nsend : 1) * m * sizeof(double));
		double *rb = (double*)malloc((size_t)(ng > 0 ? ng : 1) * m * sizeof(double));
		assert(vx->ldd >= A->ncols);
		for (i = 0; i < g_halo.nsend; ++i) for (j = 0; j < m; ++j)
			sb[(size_t)i * m + j] = vx->data[(size_t)vx->ldd * (start[0] + j) + g_halo.send_rows[i]];
		g_halo.exchange(sb, rb, m, g_halo.ctx);
		for (i = 0; i < ng; ++i) for (j = 0; j < m; ++j)
			vx->data[(size_t)vx->ldd * (start[0] + j) + nloc + i] = rb[(size_t)i * m + j];
		free(sb); free(rb);
		for (c = 0; c < m; ++c) {
			const double *xs = vx->data + (size_t)vx->ldd * (start[0] + c);
			double *yd = vy->data + (size_t)vy->ldd * (start[1] + c);
			for (i = 0; i < nloc; ++i) {   /* j_col doubles as the CSR row pointer of the slab */
				double acc = 0.0;
				for (k = A->j_col[i]; k < A->j_col[i + 1]; ++k) acc += A->data[k] * xs[A->i_row[k]];
				yd[i] = acc;
			}
		}
		return;
	}
	if (A != NULL) assert(vx->nrows == vx->ldd && vy->nrows == vy->ldd);   /* app_ccs.c:54-55 */
#pragma omp parallel for schedule(static) num_threads(g_threads)
	for (c = 0; c < m; ++c) {
		const double *xs = vx->data + (size_t)vx->ldd * (start[0] + c);
		double *yd = vy->data + (size_t)vy->ldd * (start[1] + c); int j, i;
		if (A == NULL) { memcpy(yd, xs, (size_t)vy->nrows * sizeof(double)); continue; }
		memset(yd, 0, (size_t)vy->nrows * sizeof(double));
		for (j = 0; j < A->ncols; ++j)
			for (i = A->j_col[j]; i < A->j_col[j + 1]; ++i) yd[A->i_row[i]] += A->data[i] * xs[j];
	}
}
/* app_ccs.c:140-150 — symmetric matrices: the product itself.  EXTENSION (the reference asserts a square matrix there, so its
 * CCS back-end cannot restrict through P^T: only its dense and external back-ends run BlockAMG): a rectangular CCS matrix — a
 * prolongation P of O_MultiGridCreate — is applied transposed in gather form, y[j] = sum over column j of a * x[i_row]. */
static void O_MatTransDotMultiVec(void *mat, void **x, void **y, int *start, int *end, struct OPS_ *ops)
{
	CCS *A = (CCS*)mat; VEC *vx = (VEC*)x, *vy = (VEC*)y; int c, m = end[0] - start[0];
	if (A == NULL || A->nrows == A->ncols) { O_MatDotMultiVec(mat, x, y, start, end, ops); return; }
	assert(m == end[1] - start[1] && vx->nrows == A->nrows && vy->nrows == A->ncols);
#pragma omp parallel for schedule(static) num_threads(g_threads)
	for (c = 0; c < m; ++c) {
		const double *xs = vx->data + (size_t)vx->ldd * (start[0] + c);
		double *yd = vy->data + (size_t)vy->ldd * (start[1] + c); int j, i;
		for (j = 0; j < A->ncols; ++j) {
			double acc = 0.0;
			for (i = A->j_col[j]; i < A->j_col[j + 1]; ++i) acc += A->data[i] * xs[A->i_row[i]];
			yd[j] = acc;
		}
	}
}
/* ops.h:134-139 — the hierarchy behind BlockAMG.  The reference's CCS back-end has none (app_ccs.c:213-249 leaves the slot
 * NULL); this one wraps the aggregation hierarchy of include/gcge_multigrid.h (the same one the HIP back-end uploads) into CCS
 * matrices: A_l symmetric (CSR arrays == CCS arrays), P_l (rows of level l x rows of level l + 1) as the CCS triple whose
 * column pointers are the row pointers of P_l^T. */
typedef struct { void **A_array; GCGE_MG mg; CCS *mats; } O_MG_HOLD;
static O_MG_HOLD g_mg_hold[8]; static int g_mg_nhold = 0;
static void O_MultiGridCreate(void ***A_array, void ***B_array, void ***P_array, int *num_levels, void *A, void *B, struct OPS_ *ops)
{
	CCS *mA = (CCS*)A, *mB = (CCS*)B; GCGE_CSR cA, cB; O_MG_HOLD *h; int l, L;
	assert(g_mg_nhold < 8 && mA->nrows == mA->ncols);
	h = &g_mg_hold[g_mg_nhold++];
	memset(&cA, 0, sizeof cA); memset(&cB, 0, sizeof cB);
	cA.nrows = cA.ncols = mA->nrows; cA.rowptr = mA->j_col; cA.colidx = mA->i_row; cA.val = mA->data; cA.nnz = mA->j_col[mA->nrows];
	if (mB != NULL) { cB.nrows = cB.ncols = mB->nrows; cB.rowptr = mB->j_col; cB.colidx = mB->i_row; cB.val = mB->data; cB.nnz = mB->j_col[mB->nrows]; }
	if (gcge_mg_build(&cA, mB != NULL ? &cB : NULL, *num_levels, 0, 0.0, &h->mg) != 0) { fprintf(stderr, "O_MultiGridCreate: out of memory\n"); abort(); }
	L = h->mg.num_levels;
	h->mats = (CCS*)calloc(3 * (size_t)L, sizeof(CCS));
	*A_array = (void**)calloc(L, sizeof(void*));
	*P_array = (void**)calloc(L > 1 ? L - 1 : 1, sizeof(void*));
	if (B_array != NULL) *B_array = (void**)calloc(L, sizeof(void*));
	for (l = 0; l < L; ++l) {
		CCS *a = &h->mats[3 * l], *b = &h->mats[3 * l + 1], *p = &h->mats[3 * l + 2];
		if (l == 0) (*A_array)[0] = A;
		else { a->nrows = a->ncols = h->mg.A[l].nrows; a->j_col = h->mg.A[l].rowptr; a->i_row = h->mg.A[l].colidx; a->data = h->mg.A[l].val; (*A_array)[l] = a; }
		if (B_array != NULL) {
			if (l == 0 || mB == NULL) (*B_array)[l] = l == 0 ? B : NULL;
			else { b->nrows = b->ncols = h->mg.B[l].nrows; b->j_col = h->mg.B[l].rowptr; b->i_row = h->mg.B[l].colidx; b->data = h->mg.B[l].val; (*B_array)[l] = b; }
		}
		if (l + 1 < L) {
			p->nrows = h->mg.P[l].nrows; p->ncols = h->mg.P[l].ncols;
			p->j_col = h->mg.PT[l].rowptr; p->i_row = h->mg.PT[l].colidx; p->data = h->mg.PT[l].val;
			(*P_array)[l] = p;
		}
	}
	h->A_array = *A_array;
	*num_levels = L;
}
static void O_MultiGridDestroy(void ***A_array, void ***B_array, void ***P_array, int *num_levels, struct OPS_ *ops)
{
	int i;
	for (i = 0; i < g_mg_nhold; ++i) if (g_mg_hold[i].A_array == *A_array) {
		gcge_mg_free(&g_mg_hold[i].mg); free(g_mg_hold[i].mats);
		g_mg_hold[i] = g_mg_hold[--g_mg_nhold];
		break;
	}
	free(*A_array); *A_array = NULL;
	free(*P_array); *P_array = NULL;
	if (B_array != NULL && *B_array != NULL) { free(*B_array); *B_array = NULL; }
}
/* ops_multi_vec.c:351-411 */
static void O_MultiVecQtAP(char ntsA, char ntsd, void **mvQ, void *matA, void **mvP, int is_vec,
		int *startQP, int *endQP, double *qAp, int ldQAP, void **mv_ws, struct OPS_ *ops)
{
	int s[2], e[2], k = endQP[0] - startQP[0], m = endQP[1] - startQP[1];
	if (k <= 0 || m <= 0) return;
	if (matA == NULL) {
		if (ntsd == 'T') {
			s[0] = startQP[1]; e[0] = endQP[1]; s[1] = startQP[0]; e[1] = endQP[0];
			ops->MultiVecInnerProd('N', mvP, mvQ, is_vec, s, e, qAp, ldQAP, ops);
		} else ops->MultiVecInnerProd(ntsd, mvQ, mvP, is_vec, startQP, endQP, qAp, ldQAP, ops);
		return;
	}
	s[0] = startQP[1]; e[0] = endQP[1]; s[1] = 0; e[1] = m;
	O_MatDotMultiVec(matA, mvP, mv_ws, s, e, ops);
	if (ntsd == 'T') {
		s[0] = 0; e[0] = m; s[1] = startQP[0]; e[1] = endQP[0];
		ops->MultiVecInnerProd('N', mv_ws, mvQ, is_vec, s, e, qAp, ldQAP, ops);
	} else {
		s[0] = startQP[0]; e[0] = endQP[0]; s[1] = 0; e[1] = m;
		ops->MultiVecInnerProd(ntsd, mvQ, mv_ws, is_vec, s, e, qAp, ldQAP, ops);
	}
}

/* app_ccs.c:213-249 */
void OPS_ORACLE_Set(struct OPS_ *ops)
{
	ops->Printf                   = DefaultPrintf;
	ops->GetWtime                 = DefaultGetWtime;
	ops->GetOptionFromCommandLine = DefaultGetOptionFromCommandLine;
	ops->MultiVecCreateByMat      = O_MultiVecCreateByMat;
	ops->MultiVecCreateByMultiVec = O_MultiVecCreateByMultiVec;
	ops->MultiVecDestroy          = O_MultiVecDestroy;
	ops->MultiVecView             = O_MultiVecView;
	ops->MultiVecLocalInnerProd   = O_MultiVecLocalInnerProd;
	ops->MultiVecInnerProd        = NULL;   /* OPS_Setup: local part + sum over ranks (src/ops_multi_vec.c:202-230) */
	ops->MultiVecSetRandomValue   = O_MultiVecSetRandomValue;
	ops->MultiVecAxpby            = O_MultiVecAxpby;
	ops->MultiVecLinearComb       = O_MultiVecLinearComb;
	ops->MatDotMultiVec           = O_MatDotMultiVec;
	ops->MatTransDotMultiVec      = O_MatTransDotMultiVec;
	ops->MultiVecQtAP             = O_MultiVecQtAP;
	ops->MultiGridCreate          = O_MultiGridCreate;
	ops->MultiGridDestroy         = O_MultiGridDestroy;
}
