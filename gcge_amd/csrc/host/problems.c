/* Deterministic symmetric model matrices (host CSR) — see include/gcge_problems.h.
 *
 * Reference counterparts: the 1-D pair restates test/test_app_ccs.c:142-184
 * (CreateMatrixCCS); the 3-D generators are the synthetic inputs SURVEY.md §8(d)
 * defines for BASELINE.json's configs (the reference's own 3-D matrices come
 * from PHG / PETSc binary files that are not available here).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gcge_problems.h"

void gcge_csr_free(GCGE_CSR *A)
{
	if (A == NULL) return;
	free(A->rowptr); free(A->colidx); free(A->val);
	memset(A, 0, sizeof(*A));
}

static int csr_alloc(GCGE_CSR *A, int64_t nrows, int64_t ncols, int64_t row_begin, int64_t cap)
{
	memset(A, 0, sizeof(*A));
	A->nrows = (int)nrows; A->ncols = (int)ncols; A->row_begin = (int)row_begin;
	A->rowptr = (int*)malloc((size_t)(nrows + 1) * sizeof(int));
	A->colidx = (int*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(int));
	A->val    = (double*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(double));
	if (!A->rowptr || !A->colidx || !A->val) { gcge_csr_free(A); return -1; }
	A->rowptr[0] = 0;
	return 0;
}

double gcge_uniform(uint64_t seed, uint64_t index)
{
	uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	z =  z ^ (z >> 31);
	return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

/* ---- generic constant-coefficient stencil on an N^3 grid, Dirichlet truncation ---- */
typedef struct { int di, dj, dk; double w; } StencilPt;

static int cmp_stencil(const void *a, const void *b)
{
	const StencilPt *p = (const StencilPt*)a, *q = (const StencilPt*)b;
	if (p->dk != q->dk) return p->dk < q->dk ? -1 : 1;
	if (p->dj != q->dj) return p->dj < q->dj ? -1 : 1;
	if (p->di != q->di) return p->di < q->di ? -1 : 1;
	return 0;
}

/* st must be sorted by (dk,dj,di) so that column indices ascend inside a row */
static int stencil_box(int Nx, int Ny, int Nz, const StencilPt *st, int npt, double scale,
		int64_t row_begin, int64_t row_end, GCGE_CSR *A)
{
	int64_t n = (int64_t)Nx * Ny * Nz, r; int s;
	if (row_begin < 0) row_begin = 0;
	if (row_end > n || row_end < 0) row_end = n;
	if (n > 2147483647LL) return -2;
	if (csr_alloc(A, row_end - row_begin, n, row_begin, (row_end - row_begin) * npt)) return -1;
	int64_t p = 0;
	for (r = row_begin; r < row_end; ++r) {
		int i = (int)(r % Nx), j = (int)((r / Nx) % Ny), k = (int)(r / ((int64_t)Nx * Ny));
		for (s = 0; s < npt; ++s) {
			int ii = i + st[s].di, jj = j + st[s].dj, kk = k + st[s].dk;
			if (ii < 0 || ii >= Nx || jj < 0 || jj >= Ny || kk < 0 || kk >= Nz) continue;
			A->colidx[p] = (int)(ii + (int64_t)Nx * (jj + (int64_t)Ny * kk));
			A->val[p]    = scale * st[s].w;
			++p;
		}
		if (p > 2147483647LL) { gcge_csr_free(A); return -2; }
		A->rowptr[r - row_begin + 1] = (int)p;
	}
	A->nnz = p;
	return 0;
}

static int stencil_rows(int N, const StencilPt *st, int npt, double scale,
		int64_t row_begin, int64_t row_end, GCGE_CSR *A)
{
	return stencil_box(N, N, N, st, npt, scale, row_begin, row_end, A);
}

int gcge_problem_lap3d_box(int Nx, int Ny, int Nz, int64_t row_begin, int64_t row_end, GCGE_CSR *A)
{
	StencilPt st[7] = {
		{0,0,-1,-1.0},{0,-1,0,-1.0},{-1,0,0,-1.0},{0,0,0,6.0},
		{1,0,0,-1.0},{0,1,0,-1.0},{0,0,1,-1.0}};
	qsort(st, 7, sizeof(StencilPt), cmp_stencil);
	return stencil_box(Nx, Ny, Nz, st, 7, 1.0, row_begin, row_end, A);
}

int gcge_problem_lap3d(int N, int64_t row_begin, int64_t row_end, GCGE_CSR *A)
{
	return gcge_problem_lap3d_box(N, N, N, row_begin, row_end, A);
}

int gcge_problem_fe1d(int n, GCGE_CSR *A, GCGE_CSR *B)
{
	double h = 1.0 / (n + 1); int r; int64_t p = 0;
	if (csr_alloc(A, n, n, 0, 3 * (int64_t)n)) return -1;
	for (r = 0; r < n; ++r) {
		if (r > 0)     { A->colidx[p] = r - 1; A->val[p] = -1.0 / h; ++p; }
		                 A->colidx[p] = r;     A->val[p] = +2.0 / h; ++p;
		if (r < n - 1) { A->colidx[p] = r + 1; A->val[p] = -1.0 / h; ++p; }
		A->rowptr[r + 1] = (int)p;
	}
	A->nnz = p;
	if (B != NULL) {
		if (csr_alloc(B, n, n, 0, n)) { gcge_csr_free(A); return -1; }
		for (r = 0; r < n; ++r) {
			B->colidx[r] = r; B->val[r] = 1.0 * h; B->rowptr[r + 1] = r + 1;
		}
		B->nnz = n;
	}
	return 0;
}

int gcge_problem_fe3d(int M, int64_t row_begin, int64_t row_end, GCGE_CSR *A, GCGE_CSR *B)
{
	double h = 1.0 / (M + 1); int rc;
	StencilPt sa[7] = {
		{0,0,-1,-1.0},{0,-1,0,-1.0},{-1,0,0,-1.0},{0,0,0,6.0},
		{1,0,0,-1.0},{0,1,0,-1.0},{0,0,1,-1.0}};
	StencilPt sb[15] = {
		{0,0,0,0.4},
		{1,0,0,1.0/20},{-1,0,0,1.0/20},{0,1,0,1.0/20},{0,-1,0,1.0/20},
		{0,0,1,1.0/20},{0,0,-1,1.0/20},{1,1,1,1.0/20},{-1,-1,-1,1.0/20},
		{1,1,0,1.0/30},{-1,-1,0,1.0/30},{1,0,1,1.0/30},{-1,0,-1,1.0/30},
		{0,1,1,1.0/30},{0,-1,-1,1.0/30}};
	qsort(sa, 7, sizeof(StencilPt), cmp_stencil);
	qsort(sb, 15, sizeof(StencilPt), cmp_stencil);
	rc = stencil_rows(M, sa, 7, h, row_begin, row_end, A);
	if (rc) return rc;
	if (B != NULL) {
		rc = stencil_rows(M, sb, 15, h * h * h, row_begin, row_end, B);
		if (rc) { gcge_csr_free(A); return rc; }
	}
	return 0;
}

/* ---- SiO2-like: 37-point 12th-order -Laplacian + Gaussian rank-one "atoms" ---- */
typedef struct { int col; double v; } Entry;
static int cmp_entry(const void *a, const void *b)
{
	int x = ((const Entry*)a)->col, y = ((const Entry*)b)->col;
	return x < y ? -1 : (x > y);
}

int gcge_problem_sio2_like(int G, int K, double R0, double R1, uint64_t seed,
		int64_t row_begin, int64_t row_end, GCGE_CSR *A)
{
	static const double c12[7] = { 5369.0/1800, -12.0/7, 15.0/56, -10.0/189,
		1.0/112, -2.0/1925, 1.0/16632 };
	int64_t n = (int64_t)G * G * G, r, p, cap;
	int a, d, axis;
	if (n > 2147483647LL) return -2;
	if (row_begin < 0) row_begin = 0;
	if (row_end > n || row_end < 0) row_end = n;

	/* atom supports: the FULL atom list is generated on every slab so that slabs agree */
	int     *sup_ptr = (int*)malloc((size_t)(K + 1) * sizeof(int));
	int     *sup_col = NULL; double *sup_u = NULL; int sup_cap = 0;
	sup_ptr[0] = 0;
	for (a = 0; a < K; ++a) {
		double u0 = gcge_uniform(seed, 6ULL * a + 0), u1 = gcge_uniform(seed, 6ULL * a + 1);
		double u2 = gcge_uniform(seed, 6ULL * a + 2), u3 = gcge_uniform(seed, 6ULL * a + 3);
		double u4 = gcge_uniform(seed, 6ULL * a + 4), u5 = gcge_uniform(seed, 6ULL * a + 5);
		int cx = (int)(u0 * G), cy = (int)(u1 * G), cz = (int)(u2 * G);
		double R = R0 + R1 * u3 * u4, w = 0.5 + u5;
		int ir = (int)floor(R), dx, dy, dz, cnt = sup_ptr[a];
		for (dz = -ir; dz <= ir; ++dz) for (dy = -ir; dy <= ir; ++dy) for (dx = -ir; dx <= ir; ++dx) {
			double r2 = (double)(dx * dx + dy * dy + dz * dz);
			int x = cx + dx, y = cy + dy, z = cz + dz;
			if (r2 > R * R) continue;
			if (x < 0 || x >= G || y < 0 || y >= G || z < 0 || z >= G) continue;
			if (cnt >= sup_cap) {
				sup_cap = sup_cap ? 2 * sup_cap : 4096;
				sup_col = (int*)realloc(sup_col, (size_t)sup_cap * sizeof(int));
				sup_u   = (double*)realloc(sup_u, (size_t)sup_cap * sizeof(double));
			}
			sup_col[cnt] = (int)(x + (int64_t)G * (y + (int64_t)G * z));
			sup_u[cnt]   = w * exp(-r2 / (R * R));
			++cnt;
		}
		sup_ptr[a + 1] = cnt;
	}
	/* per-row list of (atom, position inside its support) for rows of this slab */
	int64_t nloc = row_end - row_begin;
	int *cov_ptr = (int*)calloc((size_t)(nloc + 1), sizeof(int));
	for (a = 0; a < K; ++a) for (d = sup_ptr[a]; d < sup_ptr[a + 1]; ++d) {
		int64_t q = sup_col[d];
		if (q >= row_begin && q < row_end) ++cov_ptr[q - row_begin + 1];
	}
	for (r = 0; r < nloc; ++r) cov_ptr[r + 1] += cov_ptr[r];
	int *cov_atom = (int*)malloc((size_t)(cov_ptr[nloc] > 0 ? cov_ptr[nloc] : 1) * sizeof(int));
	int *cov_pos  = (int*)malloc((size_t)(cov_ptr[nloc] > 0 ? cov_ptr[nloc] : 1) * sizeof(int));
	int *fill     = (int*)calloc((size_t)(nloc > 0 ? nloc : 1), sizeof(int));
	for (a = 0; a < K; ++a) for (d = sup_ptr[a]; d < sup_ptr[a + 1]; ++d) {
		int64_t q = sup_col[d];
		if (q >= row_begin && q < row_end) {
			int at = cov_ptr[q - row_begin] + fill[q - row_begin]++;
			cov_atom[at] = a; cov_pos[at] = d;
		}
	}
	free(fill);

	cap = nloc * 64 + 1024;
	if (csr_alloc(A, nloc, n, row_begin, cap)) return -1;
	Entry *tmp = NULL; int tmp_cap = 0;
	p = 0;
	for (r = row_begin; r < row_end; ++r) {
		int i = (int)(r % G), j = (int)((r / G) % G), k = (int)(r / ((int64_t)G * G));
		int cnt = 0, need = 37, t, m;
		for (t = cov_ptr[r - row_begin]; t < cov_ptr[r - row_begin + 1]; ++t)
			need += sup_ptr[cov_atom[t] + 1] - sup_ptr[cov_atom[t]];
		if (need > tmp_cap) { tmp_cap = 2 * need; tmp = (Entry*)realloc(tmp, (size_t)tmp_cap * sizeof(Entry)); }
		tmp[cnt].col = (int)r; tmp[cnt].v = 3.0 * c12[0]; ++cnt;
		for (axis = 0; axis < 3; ++axis) for (d = 1; d <= 6; ++d) {
			int s;
			for (s = -1; s <= 1; s += 2) {
				int ii = i + (axis == 0 ? s * d : 0), jj = j + (axis == 1 ? s * d : 0);
				int kk = k + (axis == 2 ? s * d : 0);
				if (ii < 0 || ii >= G || jj < 0 || jj >= G || kk < 0 || kk >= G) continue;
				tmp[cnt].col = (int)(ii + (int64_t)G * (jj + (int64_t)G * kk));
				tmp[cnt].v = c12[d]; ++cnt;
			}
		}
		for (t = cov_ptr[r - row_begin]; t < cov_ptr[r - row_begin + 1]; ++t) {
			int at = cov_atom[t]; double up = sup_u[cov_pos[t]];
			for (d = sup_ptr[at]; d < sup_ptr[at + 1]; ++d) {
				tmp[cnt].col = sup_col[d]; tmp[cnt].v = up * sup_u[d]; ++cnt;
			}
		}
		qsort(tmp, (size_t)cnt, sizeof(Entry), cmp_entry);
		for (t = 0, m = 0; t < cnt; ++t) {  /* merge duplicates */
			if (m > 0 && tmp[m - 1].col == tmp[t].col) tmp[m - 1].v += tmp[t].v;
			else tmp[m++] = tmp[t];
		}
		if (p + m > cap) {
			cap = (p + m) * 2;
			A->colidx = (int*)realloc(A->colidx, (size_t)cap * sizeof(int));
			A->val    = (double*)realloc(A->val, (size_t)cap * sizeof(double));
		}
		for (t = 0; t < m; ++t) { A->colidx[p] = tmp[t].col; A->val[p] = tmp[t].v; ++p; }
		if (p > 2147483647LL) { gcge_csr_free(A); p = -1; break; }
		A->rowptr[r - row_begin + 1] = (int)p;
	}
	free(tmp); free(cov_ptr); free(cov_atom); free(cov_pos);
	free(sup_ptr); free(sup_col); free(sup_u);
	if (p < 0) return -2;
	A->nnz = p;
	return 0;
}

/* The same operator on a BALL inside the G^3 box — the domain of the real-space DFT matrices behind BASELINE config 5 (PARSEC:
 * grid points inside a sphere, numbered in scan order, x fastest; test/submit.sh:9-15 of the reference): the principal submatrix
 * of gcge_problem_sio2_like over the grid points with (x - c)^2 + (y - c)^2 + (z - c)^2 <= (G / 2)^2, c = (G - 1) / 2, i.e. the
 * stencil and the atom blocks truncated at the sphere (Dirichlet), still symmetric positive definite.  Rows are the points
 * inside in scan order.  box_of_row (may be NULL): for every row of the slab its box index x + G (y + G z). */
static int in_ball(int G, int x, int y, int z)
{
	/* doubled coordinates keep it in integers: (2x - (G-1))^2 + ... <= G^2 */
	long a = 2L * x - (G - 1), b = 2L * y - (G - 1), c = 2L * z - (G - 1);
	return a * a + b * b + c * c <= (long)G * G;
}
int64_t gcge_problem_sio2_ball_rows(int G)
{
	int64_t n = 0; int x, y, z;
	for (z = 0; z < G; ++z) for (y = 0; y < G; ++y) for (x = 0; x < G; ++x) n += in_ball(G, x, y, z);
	return n;
}
int gcge_problem_sio2_ball(int G, int K, double R0, double R1, uint64_t seed,
		int64_t row_begin, int64_t row_end, GCGE_CSR *A, int **box_of_row)
{
	int64_t nbox = (int64_t)G * G * G, nball = 0, q, b0 = -1, b1 = -1, r, p = 0;
	int *idx, x, y, z, rc;
	GCGE_CSR Bx;
	if (nbox > 2147483647LL) return -2;
	idx = (int*)malloc((size_t)nbox * sizeof(int));
	if (idx == NULL) return -1;
	for (z = 0, q = 0; z < G; ++z) for (y = 0; y < G; ++y) for (x = 0; x < G; ++x, ++q)
		idx[q] = in_ball(G, x, y, z) ? (int)nball++ : -1;
	if (row_begin < 0) row_begin = 0;
	if (row_end > nball || row_end < 0) row_end = nball;
	/* the box rows that enclose the slab (scan order is kept by the numbering) */
	for (q = 0; q < nbox; ++q) {
		if (idx[q] == row_begin && b0 < 0) b0 = q;
		if (idx[q] >= 0 && idx[q] < row_end) b1 = q;
	}
	if (row_end <= row_begin) { b0 = 0; b1 = -1; }
	memset(&Bx, 0, sizeof Bx);
	rc = gcge_problem_sio2_like(G, K, R0, R1, seed, b0, b1 + 1, &Bx);
	if (rc != 0) { free(idx); return rc; }
	if (csr_alloc(A, row_end - row_begin, (int)nball, row_begin, Bx.nnz > 0 ? Bx.nnz : 1)) { gcge_csr_free(&Bx); free(idx); return -1; }
	if (box_of_row != NULL) *box_of_row = (int*)malloc((size_t)(row_end - row_begin > 0 ? row_end - row_begin : 1) * sizeof(int));
	A->rowptr[0] = 0;
	for (r = 0, q = b0; q <= b1; ++q) {
		int k;
		if (idx[q] < 0) continue;
		for (k = Bx.rowptr[q - b0]; k < Bx.rowptr[q - b0 + 1]; ++k) {
			const int c = idx[Bx.colidx[k]];
			if (c < 0) continue;                      /* a neighbour outside the sphere: dropped (Dirichlet) */
			A->colidx[p] = c; A->val[p] = Bx.val[k]; ++p;
		}
		if (box_of_row != NULL) (*box_of_row)[r] = (int)q;
		A->rowptr[++r] = (int)p;
	}
	A->nnz = p;
	gcge_csr_free(&Bx); free(idx);
	return 0;
}

/* ---- row-partition helpers ---------------------------------------------------------- */
static int cmp_int(const void *a, const void *b)
{
	int x = *(const int*)a, y = *(const int*)b;
	return x < y ? -1 : (x > y);
}
void gcge_free_ints(int *p) { free(p); }

int gcge_dist_ghosts(const GCGE_CSR *A, int **ghosts_out, int *nghost_out)
{
	int64_t k, cnt = 0, cap = 1024, lo = A->row_begin, hi = (int64_t)A->row_begin + A->nrows;
	int *g = (int*)malloc((size_t)cap * sizeof(int)), last = -1, m = 0, i;
	if (g == NULL) return -1;
	for (k = 0; k < A->nnz; ++k) {
		int c = A->colidx[k];
		if (c >= lo && c < hi) continue;
		if (c == last) continue;               /* cheap filter for runs of equal columns */
		if (cnt == cap) { cap *= 2; g = (int*)realloc(g, (size_t)cap * sizeof(int)); if (!g) return -1; }
		g[cnt++] = c; last = c;
	}
	qsort(g, (size_t)cnt, sizeof(int), cmp_int);
	for (i = 0; i < cnt; ++i) if (m == 0 || g[m - 1] != g[i]) g[m++] = g[i];
	*ghosts_out = g; *nghost_out = m;
	return 0;
}

int gcge_dist_localize(GCGE_CSR *A, const int *ghosts, int nghost)
{
	int64_t k, lo = A->row_begin, hi = (int64_t)A->row_begin + A->nrows;
	for (k = 0; k < A->nnz; ++k) {
		int c = A->colidx[k];
		if (c >= lo && c < hi) { A->colidx[k] = (int)(c - lo); continue; }
		{   /* binary search in the ghost list */
			int a = 0, b = nghost - 1, pos = -1;
			while (a <= b) { int mid = (a + b) / 2; if (ghosts[mid] == c) { pos = mid; break; } if (ghosts[mid] < c) a = mid + 1; else b = mid - 1; }
			if (pos < 0) return -1;
			A->colidx[k] = A->nrows + pos;
		}
	}
	A->ncols = A->nrows + nghost;
	return 0;
}

/* Halo plan of a row slab over ANY transport (one planner for every back-end and every transport: RCCL inside the HIP back-end,
 * torch.distributed / gloo in the tests).  What the reference's distributed back-ends get from PETSc / hypre / PHG when a matrix
 * is assembled (app/app_phg.c:292-359 uses the result: phgMapScatterBegin/End).  See include/gcge_problems.h. */
int gcge_dist_plan_halo(const long *part, const int *ghosts, int nghost, const GCGE_PLAN_TRANSPORT *t,
		int *recv_cnt, int *send_cnt, int **send_rows_out, int *nsend_out)
{
	const int world = t->size, rank = t->rank;
	int q, i, *all = NULL, *want = NULL, *gsend = NULL; long ns = 0;
	*send_rows_out = NULL; *nsend_out = 0;
	for (q = 0; q < world; ++q) { recv_cnt[q] = 0; send_cnt[q] = 0; }
	/* who owns each of my halo rows (ascending global ids: grouped by owner, owners ascending) */
	for (i = 0, q = 0; i < nghost; ++i) {
		if (i > 0 && ghosts[i] <= ghosts[i - 1]) return -2;                 /* not ascending / not unique */
		while (q < world && (long)ghosts[i] >= part[q + 1]) ++q;
		if (q >= world || (long)ghosts[i] < part[q] || q == rank) return -3;  /* outside the partition, or one of my own rows */
		++recv_cnt[q];
	}
	if (world == 1) return nghost == 0 ? 0 : -3;
	/* 1. counts: need[p][q] = rows slab p needs from slab q; I ship need[p][me] rows to p */
	all = (int*)malloc((size_t)world * world * sizeof(int));
	if (all == NULL) return -1;
	t->allgather_int(recv_cnt, world, all, t->ctx);
	for (q = 0; q < world; ++q) { send_cnt[q] = all[(size_t)q * world + rank]; ns += send_cnt[q]; }
	free(all);
	if (send_cnt[rank] != 0) return -4;
	/* 2. index lists: every owner learns which of its rows (global ids, ascending) I need */
	want = (int*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(int));
	gsend = (int*)malloc((size_t)(nghost > 0 ? nghost : 1) * sizeof(int));
	if (want == NULL || gsend == NULL) { free(want); free(gsend); return -1; }
	memcpy(gsend, ghosts, (size_t)nghost * sizeof(int));
	t->exchange_int(gsend, recv_cnt, want, send_cnt, t->ctx);            /* (I SEND recv_cnt[q] ids to q and RECEIVE send_cnt[q] from it) */
	free(gsend);
	for (i = 0; i < ns; ++i) {
		if ((long)want[i] < part[rank] || (long)want[i] >= part[rank + 1]) { free(want); return -5; }   /* a peer asked for a row I do not own */
		want[i] -= (int)part[rank];
	}
	*send_rows_out = want; *nsend_out = (int)ns;
	return 0;
}

/* ---------------------------------------------------------------- ingestion */
#include <stdio.h>

static uint32_t be32(const unsigned char *q) { return ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; }
static double be64f(const unsigned char *q)
{
	uint64_t u = 0; double d; int i;
	for (i = 0; i < 8; ++i) u = (u << 8) | q[i];
	memcpy(&d, &u, 8);
	return d;
}
static int read_be32(FILE *f, int32_t *dst, int64_t count)
{
	unsigned char buf[4096]; int64_t done = 0;
	while (done < count) {
		int64_t chunk = count - done, i; if (chunk > 1024) chunk = 1024;
		if (fread(buf, 4, (size_t)chunk, f) != (size_t)chunk) return -1;
		if (dst) for (i = 0; i < chunk; ++i) dst[done + i] = (int32_t)be32(buf + 4 * i);
		done += chunk;
	}
	return 0;
}

int gcge_load_petsc_binary(const char *path, int64_t row_begin, int64_t row_end, GCGE_CSR *A)
{
	FILE *f = fopen(path, "rb");
	unsigned char hdr[16], buf[4096];
	int32_t rows, cols, nnz, *len = NULL;
	int64_t r, first = 0, mine = 0, skip_after, i;
	if (f == NULL) return -1;
	if (fread(hdr, 1, 16, f) != 16) { fclose(f); return -1; }
	if (be32(hdr) != 1211216u) { fclose(f); return -2; }     /* MAT_FILE_CLASSID */
	rows = (int32_t)be32(hdr + 4); cols = (int32_t)be32(hdr + 8); nnz = (int32_t)be32(hdr + 12);
	if (rows < 0 || cols < 0 || nnz < 0) { fclose(f); return -2; }   /* nnz = -1 marks a dense file */
	if (row_end < 0 || row_end > rows) row_end = rows;
	if (row_begin < 0 || row_begin > row_end) { fclose(f); return -2; }
	len = (int32_t*)malloc((size_t)(rows > 0 ? rows : 1) * sizeof(int32_t));
	if (len == NULL) { fclose(f); return -3; }
	if (read_be32(f, len, rows)) { free(len); fclose(f); return -1; }
	for (r = 0; r < row_begin; ++r) first += len[r];
	for (r = row_begin; r < row_end; ++r) mine += len[r];
	skip_after = (int64_t)nnz - first - mine;
	if (skip_after < 0 || csr_alloc(A, row_end - row_begin, cols, row_begin, mine)) { free(len); fclose(f); return skip_after < 0 ? -2 : -3; }
	A->rowptr[0] = 0;
	for (r = row_begin; r < row_end; ++r) A->rowptr[r - row_begin + 1] = A->rowptr[r - row_begin] + len[r];
	A->nnz = mine;
	free(len);
	/* column indices: skip `first`, read `mine`, skip the rest; then the values likewise */
	if (fseek(f, (long)(4 * first), SEEK_CUR) || read_be32(f, A->colidx, mine) || fseek(f, (long)(4 * skip_after), SEEK_CUR)
			|| fseek(f, (long)(8 * first), SEEK_CUR)) { gcge_csr_free(A); fclose(f); return -1; }
	for (i = 0; i < mine; ) {
		int64_t chunk = mine - i, k; if (chunk > 512) chunk = 512;
		if (fread(buf, 8, (size_t)chunk, f) != (size_t)chunk) { gcge_csr_free(A); fclose(f); return -1; }
		for (k = 0; k < chunk; ++k) A->val[i + k] = be64f(buf + 8 * k);
		i += chunk;
	}
	fclose(f);
	for (i = 0; i < mine; ++i) if (A->colidx[i] < 0 || A->colidx[i] >= cols) { gcge_csr_free(A); return -2; }
	return 0;
}

int gcge_save_petsc_binary(const char *path, const GCGE_CSR *A)
{
	FILE *f = fopen(path, "wb");
	unsigned char q[8]; int64_t i; int r, k;
	uint32_t hdr[4];
	if (f == NULL) return -1;
	hdr[0] = 1211216u; hdr[1] = (uint32_t)A->nrows; hdr[2] = (uint32_t)A->ncols; hdr[3] = (uint32_t)A->nnz;
	for (k = 0; k < 4; ++k) { q[0] = hdr[k] >> 24; q[1] = hdr[k] >> 16; q[2] = hdr[k] >> 8; q[3] = hdr[k]; fwrite(q, 1, 4, f); }
	for (r = 0; r < A->nrows; ++r) { uint32_t v = (uint32_t)(A->rowptr[r + 1] - A->rowptr[r]); q[0] = v >> 24; q[1] = v >> 16; q[2] = v >> 8; q[3] = v; fwrite(q, 1, 4, f); }
	for (i = 0; i < A->nnz; ++i) { uint32_t v = (uint32_t)A->colidx[i]; q[0] = v >> 24; q[1] = v >> 16; q[2] = v >> 8; q[3] = v; fwrite(q, 1, 4, f); }
	for (i = 0; i < A->nnz; ++i) { uint64_t u; memcpy(&u, &A->val[i], 8); for (k = 0; k < 8; ++k) q[k] = (unsigned char)(u >> (56 - 8 * k)); fwrite(q, 1, 8, f); }
	return fclose(f) ? -1 : 0;
}

int gcge_csr_from_ccs(int nrows, int ncols, const int *j_col, const int *i_row, const double *data,
		int one_based, GCGE_CSR *A)
{
	const int ob = one_based ? 1 : 0;
	int64_t nnz = (int64_t)j_col[ncols] - ob, k; int c, r;
	int *fill;
	if (nnz < 0 || csr_alloc(A, nrows, ncols, 0, nnz)) return -3;
	A->nnz = nnz;
	memset(A->rowptr, 0, ((size_t)nrows + 1) * sizeof(int));
	for (k = 0; k < nnz; ++k) { r = i_row[k] - ob; if (r < 0 || r >= nrows) { gcge_csr_free(A); return -2; } ++A->rowptr[r + 1]; }
	for (r = 0; r < nrows; ++r) A->rowptr[r + 1] += A->rowptr[r];
	fill = (int*)malloc((size_t)(nrows > 0 ? nrows : 1) * sizeof(int));
	if (fill == NULL) { gcge_csr_free(A); return -3; }
	memcpy(fill, A->rowptr, (size_t)nrows * sizeof(int));
	for (c = 0; c < ncols; ++c)          /* columns ascending => column indices ascending inside every row */
		for (k = j_col[c] - ob; k < j_col[c + 1] - ob; ++k) {
			r = i_row[k] - ob;
			A->colidx[fill[r]] = c; A->val[fill[r]] = data[k]; ++fill[r];
		}
	free(fill);
	return 0;
}

/* ---- Matrix Market coordinate files (the form the SuiteSparse collection ships the PARSEC matrices of the reference's
 * test/submit.sh:9-15 in: "%%MatrixMarket matrix coordinate real symmetric", 1-based, lower triangle).  general / symmetric /
 * skew-symmetric, real / integer / pattern (pattern: every entry 1.0); entries in any order, duplicates summed; the result is
 * the full matrix in CSR with ascending columns.  0 ok, -1 cannot open / short file, -2 not such a file (or complex / array
 * format, more than 2^31 - 1 entries), -3 out of memory. */
typedef struct { int r, c; double v; } mm_entry;
static int mm_cmp(const void *a, const void *b)
{
	const mm_entry *x = (const mm_entry*)a, *y = (const mm_entry*)b;
	if (x->r != y->r) return x->r < y->r ? -1 : 1;
	return x->c < y->c ? -1 : x->c > y->c;
}
int gcge_load_matrix_market(const char *path, GCGE_CSR *A)
{
	FILE *f = fopen(path, "r");
	char line[1024], obj[64], fmt[64], field[64], sym[64];
	long rows, cols, nz, k, m = 0, out;
	int symmetric = 0, skew = 0, pattern = 0;
	mm_entry *e;
	if (f == NULL) return -1;
	if (fgets(line, sizeof line, f) == NULL) { fclose(f); return -1; }
	if (sscanf(line, "%%%%MatrixMarket %63s %63s %63s %63s", obj, fmt, field, sym) != 4) { fclose(f); return -2; }
	for (k = 0; obj[k]; ++k) if (obj[k] >= 'A' && obj[k] <= 'Z') obj[k] += 32;
	for (k = 0; fmt[k]; ++k) if (fmt[k] >= 'A' && fmt[k] <= 'Z') fmt[k] += 32;
	for (k = 0; field[k]; ++k) if (field[k] >= 'A' && field[k] <= 'Z') field[k] += 32;
	for (k = 0; sym[k]; ++k) if (sym[k] >= 'A' && sym[k] <= 'Z') sym[k] += 32;
	if (strcmp(obj, "matrix") != 0 || strcmp(fmt, "coordinate") != 0) { fclose(f); return -2; }
	if (strcmp(field, "pattern") == 0) pattern = 1;
	else if (strcmp(field, "real") != 0 && strcmp(field, "integer") != 0 && strcmp(field, "double") != 0) { fclose(f); return -2; }
	if (strcmp(sym, "symmetric") == 0) symmetric = 1;
	else if (strcmp(sym, "skew-symmetric") == 0) { symmetric = 1; skew = 1; }
	else if (strcmp(sym, "general") != 0) { fclose(f); return -2; }
	do { if (fgets(line, sizeof line, f) == NULL) { fclose(f); return -1; } } while (line[0] == '%' || line[0] == '\n' || line[0] == '\r');
	if (sscanf(line, "%ld %ld %ld", &rows, &cols, &nz) != 3 || rows < 0 || cols < 0 || nz < 0 || rows > 2147483647L || cols > 2147483647L) { fclose(f); return -2; }
	e = (mm_entry*)malloc((size_t)(nz > 0 ? (symmetric ? 2 * nz : nz) : 1) * sizeof(mm_entry));
	if (e == NULL) { fclose(f); return -3; }
	for (k = 0; k < nz; ++k) {
		long r, c; double v = 1.0; int got;
		if (fgets(line, sizeof line, f) == NULL) { free(e); fclose(f); return -1; }
		if (line[0] == '%' || line[0] == '\n' || line[0] == '\r') { --k; continue; }
		got = pattern ? sscanf(line, "%ld %ld", &r, &c) : sscanf(line, "%ld %ld %lf", &r, &c, &v);
		if (got != (pattern ? 2 : 3) || r < 1 || r > rows || c < 1 || c > cols) { free(e); fclose(f); return -2; }
		e[m].r = (int)(r - 1); e[m].c = (int)(c - 1); e[m].v = v; ++m;
		if (symmetric && r != c) { e[m].r = (int)(c - 1); e[m].c = (int)(r - 1); e[m].v = skew ? -v : v; ++m; }
	}
	fclose(f);
	qsort(e, (size_t)m, sizeof(mm_entry), mm_cmp);
	for (k = 0, out = 0; k < m; ++k) {                  /* duplicates: summed (the format allows them in an assembled file) */
		if (out > 0 && e[out - 1].r == e[k].r && e[out - 1].c == e[k].c) e[out - 1].v += e[k].v;
		else e[out++] = e[k];
	}
	if (out > 2147483647L || csr_alloc(A, rows, (int)cols, 0, out)) { free(e); return out > 2147483647L ? -2 : -3; }
	A->nnz = out;
	memset(A->rowptr, 0, ((size_t)rows + 1) * sizeof(int));
	for (k = 0; k < out; ++k) { ++A->rowptr[e[k].r + 1]; A->colidx[k] = e[k].c; A->val[k] = e[k].v; }
	for (k = 0; k < rows; ++k) A->rowptr[k + 1] += A->rowptr[k];
	free(e);
	return 0;
}

/* writer of the same format (tests, data exchange): symmetric != 0 writes the lower triangle only ("real symmetric") */
int gcge_save_matrix_market(const char *path, const GCGE_CSR *A, int symmetric)
{
	FILE *f = fopen(path, "w");
	int64_t cnt = 0; int r, k;
	if (f == NULL) return -1;
	for (r = 0; r < A->nrows; ++r)
		for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) if (!symmetric || A->colidx[k] <= r + A->row_begin) ++cnt;
	fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n%% written by gcge_save_matrix_market\n%d %d %lld\n",
			symmetric ? "symmetric" : "general", A->nrows, A->ncols, (long long)cnt);
	for (r = 0; r < A->nrows; ++r)
		for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k)
			if (!symmetric || A->colidx[k] <= r + A->row_begin)
				fprintf(f, "%d %d %.17g\n", r + A->row_begin + 1, A->colidx[k] + 1, A->val[k]);
	return fclose(f) ? -1 : 0;
}
