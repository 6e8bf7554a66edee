/* Aggregation multigrid hierarchy on host CSR arrays — what our back-ends' MultiGridCreate slots are built from
 * (include/gcge_multigrid.h).  The reference leaves the hierarchy to the back-end (src/ops.h:134-139): app/app_slepc.c:648-728
 * extracts it from PETSc GAMG, app/app_hypre.c from BoomerAMG, app/app_lapack.c:863-929 builds a fixed 1-D toy.  Neither library
 * exists here, so the hierarchy is our own: plain aggregation, Galerkin coarse operators (optionally rescaled), transfers that
 * are one 1.0 per row.  Consumers: BlockAMG (lin_sol.c; reference src/ops_lin_sol.c:466-715), DefaultMultiVecFromItoJ
 * (ops_table.c; reference src/ops_multi_grid.c:69-117).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "gcge_multigrid.h"

static double g_scale = 0.5; static int g_min_rows = 64; static double g_theta = 0.25;
void gcge_mg_set_defaults(double scale, int min_rows, double theta)
{
	if (scale > 0.0) g_scale = scale;
	if (min_rows > 0) g_min_rows = min_rows;
	if (theta >= 0.0) g_theta = theta;
}
void gcge_mg_get_defaults(double *scale, int *min_rows, double *theta)
{
	if (scale) *scale = g_scale;
	if (min_rows) *min_rows = g_min_rows;
	if (theta) *theta = g_theta;
}

/* ---------------------------------------------------------------- grid detection */
static int cmp_int(const void *a, const void *b) { int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }

int gcge_mg_detect_grid(const GCGE_CSR *A, int dims[3], int *arm_out)
{
	/* a whole matrix, or a slab of its rows [row_begin, row_begin + nrows) with GLOBAL columns (ncols = the global size) */
	const int nloc = A->nrows, n = A->ncols > A->nrows ? A->ncols : A->nrows; const long rb = A->row_begin;
	int *off = NULL, noff = 0, cap = 0, *freq = NULL, nfreq = 0, i, arm, nx, nxy = 0;
	long sampled = 0, step;
	if (nloc < 8 || (A->ncols != nloc && A->ncols < rb + nloc)) return 0;
	/* positive column offsets of a sample of rows (all of them up to 2^20 rows), with their counts */
	step = nloc > (1 << 20) ? nloc / (1 << 20) : 1;
	{
		/* offsets are collected row by row into a list, sorted, run-length counted */
		long r; size_t tot = 0;
		for (r = 0; r < nloc; r += step) tot += (size_t)(A->rowptr[r + 1] - A->rowptr[r]);
		off = (int*)malloc((tot ? tot : 1) * sizeof(int));
		if (off == NULL) return 0;
		for (r = 0; r < nloc; r += step, ++sampled) {
			int k;
			for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k)
				if (A->colidx[k] > rb + r) off[noff++] = (int)(A->colidx[k] - (rb + r));
		}
		qsort(off, noff, sizeof(int), cmp_int);
		cap = 64; freq = (int*)malloc(cap * sizeof(int));
		for (i = 0; i < noff && freq != NULL; ) {
			int j = i; while (j < noff && off[j] == off[i]) ++j;
			/* "frequent": a quarter of the sampled rows carry it (a face of the box loses a 1 / N share of any offset) */
			if ((long)(j - i) * 4 >= sampled) {
				if (nfreq == cap) { cap *= 2; freq = (int*)realloc(freq, cap * sizeof(int)); if (freq == NULL) break; }
				freq[nfreq++] = off[i];
			}
			i = j;
		}
		free(off);
		if (freq == NULL) return 0;
	}
	if (nfreq == 0 || freq[0] != 1) { free(freq); return 0; }
	for (arm = 1; arm < nfreq && freq[arm] == arm + 1; ++arm) ;
	if (arm == nfreq) {          /* couplings along one line only: a 1-D grid */
		free(freq);
		dims[0] = n; dims[1] = 1; dims[2] = 1;
		if (arm_out) *arm_out = arm;
		return 1;
	}
	nx = freq[arm];
	if (nx <= arm || n % nx != 0) { free(freq); return 0; }
	for (i = arm + 1; i < nfreq; ++i)
		if (freq[i] % nx == 0 && freq[i] / nx > arm) { nxy = freq[i]; break; }
	free(freq);
	/* the line length must show in the rows: the last point of a line has no + 1 neighbour (Dirichlet truncation / the
	 * next line starts there), every other sampled point has one */
	{
		long r, bad = 0, seen = 0;
		for (r = 0; r < nloc; r += step) {
			int k, has = 0;
			for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) has |= A->colidx[k] == rb + r + 1;
			if ((rb + r + 1) % nx == 0) { ++seen; bad += has; }
		}
		if (seen > 0 && bad * 10 > seen) return 0;
	}
	if (nxy == 0) { dims[0] = nx; dims[1] = n / nx; dims[2] = 1; }
	else {
		if (n % nxy != 0) return 0;
		dims[0] = nx; dims[1] = nxy / nx; dims[2] = n / nxy;
	}
	if (arm_out) *arm_out = arm;
	return 1;
}

int gcge_mg_aggregate_grid(const int dims[3], int *agg, int cdims[3])
{
	const int nx = dims[0], ny = dims[1], nz = dims[2];
	const int cx = (nx + 1) / 2, cy = (ny + 1) / 2, cz = (nz + 1) / 2;
	long z;
	cdims[0] = cx; cdims[1] = cy; cdims[2] = cz;
#pragma omp parallel for schedule(static)
	for (z = 0; z < nz; ++z) {
		int y, x;
		for (y = 0; y < ny; ++y) {
			int *row = agg + ((size_t)z * ny + y) * nx;
			const int base = cx * ((y / 2) + cy * (int)(z / 2));
			for (x = 0; x < nx; ++x) row[x] = base + x / 2;
		}
	}
	return cx * cy * cz;
}

int gcge_mg_aggregate_graph(const GCGE_CSR *A, double theta, int *agg)
{
	const int n = A->nrows;
	int nc = 0, r, k;
	double *thr = (double*)malloc((n > 0 ? n : 1) * sizeof(double));
	if (thr == NULL) return -3;
	for (r = 0; r < n; ++r) {
		double mx = 0.0;
		for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k)
			if (A->colidx[k] != r && fabs(A->val[k]) > mx) mx = fabs(A->val[k]);
		thr[r] = theta * mx;
		agg[r] = -1;
	}
	/* pass 1: a node whose strong neighbourhood is entirely free becomes the root of an aggregate */
	for (r = 0; r < n; ++r) {
		int free_all = 1, any = 0;
		if (agg[r] != -1) continue;
		for (k = A->rowptr[r]; k < A->rowptr[r + 1] && free_all; ++k) {
			const int c = A->colidx[k];
			if (c == r || c >= n || fabs(A->val[k]) < thr[r] || A->val[k] == 0.0) continue;
			any = 1;
			if (agg[c] != -1) free_all = 0;
		}
		if (!free_all || !any) continue;
		agg[r] = nc;
		for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) {
			const int c = A->colidx[k];
			if (c != r && c < n && fabs(A->val[k]) >= thr[r] && A->val[k] != 0.0) agg[c] = nc;
		}
		++nc;
	}
	/* pass 2: the rest joins the aggregate (of pass 1) it is coupled to most strongly; marked - 2 - id first so that a node
	 * attached in this pass does not attract others */
	for (r = 0; r < n; ++r) {
		double best = -1.0; int to = -1;
		if (agg[r] != -1) continue;
		for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) {
			const int c = A->colidx[k];
			if (c == r || c >= n || agg[c] < 0) continue;
			if (fabs(A->val[k]) > best) { best = fabs(A->val[k]); to = agg[c]; }
		}
		if (to >= 0 && best > 0.0) agg[r] = -2 - to;
	}
	for (r = 0; r < n; ++r) if (agg[r] <= -2) agg[r] = -2 - agg[r];
	/* pass 3: whatever is left (rows without couplings, islands of pass-2 leftovers) */
	for (r = 0; r < n; ++r) {
		if (agg[r] != -1) continue;
		agg[r] = nc;
		for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) {
			const int c = A->colidx[k];
			if (c != r && c < n && agg[c] == -1 && fabs(A->val[k]) >= thr[r] && A->val[k] != 0.0) agg[c] = nc;
		}
		++nc;
	}
	free(thr);
	return nc;
}

/* ---------------------------------------------------------------- transfers and coarse operators */
/* members of every aggregate in ascending fine-row order: ptr[nc + 1], mem[nf] */
static int aggregate_members(const int *agg, int nf, int nc, int **ptr_out, int **mem_out)
{
	int *ptr = (int*)calloc((size_t)nc + 1, sizeof(int)), *mem = (int*)malloc((nf > 0 ? nf : 1) * sizeof(int)), r;
	if (ptr == NULL || mem == NULL) { free(ptr); free(mem); return -3; }
	for (r = 0; r < nf; ++r) ++ptr[agg[r] + 1];
	for (r = 0; r < nc; ++r) ptr[r + 1] += ptr[r];
	for (r = 0; r < nf; ++r) mem[ptr[agg[r]]++] = r;
	for (r = nc; r > 0; --r) ptr[r] = ptr[r - 1];
	ptr[0] = 0;
	*ptr_out = ptr; *mem_out = mem;
	return 0;
}

int gcge_mg_prolongation(const int *agg, int nf, int nc, GCGE_CSR *P, GCGE_CSR *PT)
{
	int r;
	memset(P, 0, sizeof *P); memset(PT, 0, sizeof *PT);
	P->nrows = nf; P->ncols = nc; P->nnz = nf;
	P->rowptr = (int*)malloc(((size_t)nf + 1) * sizeof(int));
	P->colidx = (int*)malloc((nf > 0 ? nf : 1) * sizeof(int));
	P->val = (double*)malloc((nf > 0 ? nf : 1) * sizeof(double));
	PT->nrows = nc; PT->ncols = nf; PT->nnz = nf;
	PT->val = (double*)malloc((nf > 0 ? nf : 1) * sizeof(double));
	if (!P->rowptr || !P->colidx || !P->val || !PT->val || aggregate_members(agg, nf, nc, &PT->rowptr, &PT->colidx) != 0) {
		gcge_csr_free(P); gcge_csr_free(PT); return -3;
	}
	for (r = 0; r < nf; ++r) { P->rowptr[r] = r; P->colidx[r] = agg[r]; P->val[r] = 1.0; PT->val[r] = 1.0; }
	P->rowptr[nf] = nf;
	return 0;
}

int gcge_mg_galerkin(const GCGE_CSR *A, const int *agg, int nc, double scale, GCGE_CSR *Ac)
{
	const int nf = A->nrows;
	int *ptr = NULL, *mem = NULL, *cnt = NULL, rc = 0, I;
	int64_t tot = 0;
	memset(Ac, 0, sizeof *Ac);
	if (aggregate_members(agg, nf, nc, &ptr, &mem) != 0) return -3;
	cnt = (int*)calloc((size_t)nc + 1, sizeof(int));
	if (cnt == NULL) { free(ptr); free(mem); return -3; }
	/* two passes over the coarse rows with a marker array per thread: count the distinct coarse columns, then fill */
#pragma omp parallel
	{
		int *mark = (int*)malloc((nc > 0 ? nc : 1) * sizeof(int)), J;
		if (mark == NULL) {
#pragma omp atomic write
			rc = -3;
		} else {
			for (J = 0; J < nc; ++J) mark[J] = -1;
#pragma omp for schedule(dynamic, 256)
			for (I = 0; I < nc; ++I) {
				int q, k, c = 0;
				for (q = ptr[I]; q < ptr[I + 1]; ++q) {
					const int r = mem[q];
					for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) {
						const int col = A->colidx[k];
						if (col >= nf) continue;                       /* (halo columns of a slab: not coarsened here) */
						if (mark[agg[col]] != I) { mark[agg[col]] = I; ++c; }
					}
				}
				cnt[I + 1] = c;
			}
			free(mark);
		}
	}
	if (rc != 0) { free(ptr); free(mem); free(cnt); return rc; }
	for (I = 0; I < nc; ++I) { tot += cnt[I + 1]; if (tot > 2147483647LL) { free(ptr); free(mem); free(cnt); return -2; } cnt[I + 1] = (int)tot; }
	Ac->nrows = nc; Ac->ncols = nc; Ac->row_begin = 0; Ac->nnz = tot;
	Ac->rowptr = cnt;
	Ac->colidx = (int*)malloc((tot > 0 ? (size_t)tot : 1) * sizeof(int));
	Ac->val = (double*)malloc((tot > 0 ? (size_t)tot : 1) * sizeof(double));
	if (Ac->colidx == NULL || Ac->val == NULL) { free(ptr); free(mem); gcge_csr_free(Ac); return -3; }
#pragma omp parallel
	{
		int *pos = (int*)malloc((nc > 0 ? nc : 1) * sizeof(int)), J;
		if (pos == NULL) {
#pragma omp atomic write
			rc = -3;
		} else {
			for (J = 0; J < nc; ++J) pos[J] = -1;
#pragma omp for schedule(dynamic, 256)
			for (I = 0; I < nc; ++I) {
				const int base = Ac->rowptr[I];
				int q, k, c = 0, a, b;
				/* distinct columns first, sorted; then the sums in fine-row / storage order (deterministic whatever the thread count) */
				for (q = ptr[I]; q < ptr[I + 1]; ++q) {
					const int r = mem[q];
					for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) {
						const int col = A->colidx[k];
						if (col >= nf) continue;
						if (pos[agg[col]] < base || pos[agg[col]] >= base + c || Ac->colidx[pos[agg[col]]] != agg[col]) {
							pos[agg[col]] = base + c; Ac->colidx[base + c] = agg[col]; ++c;
						}
					}
				}
				qsort(Ac->colidx + base, c, sizeof(int), cmp_int);
				for (a = 0; a < c; ++a) { pos[Ac->colidx[base + a]] = base + a; Ac->val[base + a] = 0.0; }
				for (q = ptr[I]; q < ptr[I + 1]; ++q) {
					const int r = mem[q];
					for (k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) {
						const int col = A->colidx[k];
						if (col >= nf) continue;
						Ac->val[pos[agg[col]]] += A->val[k];
					}
				}
				if (scale != 1.0) for (b = 0; b < c; ++b) Ac->val[base + b] *= scale;
			}
			free(pos);
		}
	}
	free(ptr); free(mem);
	if (rc != 0) { gcge_csr_free(Ac); return rc; }
	return 0;
}

/* ---------------------------------------------------------------- the hierarchy */
void gcge_mg_free(GCGE_MG *mg)
{
	int l;
	if (mg == NULL) return;
	for (l = 1; l < mg->num_levels; ++l) {
		if (mg->A) gcge_csr_free(&mg->A[l]);
		if (mg->B) gcge_csr_free(&mg->B[l]);
	}
	for (l = 0; l + 1 < mg->num_levels; ++l) {
		if (mg->P) gcge_csr_free(&mg->P[l]);
		if (mg->PT) gcge_csr_free(&mg->PT[l]);
	}
	free(mg->A); free(mg->B); free(mg->P); free(mg->PT); free(mg->dims);
	memset(mg, 0, sizeof *mg);
}

int gcge_mg_build(const GCGE_CSR *A, const GCGE_CSR *B, int max_levels, int min_rows, double scale, GCGE_MG *mg)
{
	int l, have_grid, dims[3] = {0, 0, 0};
	memset(mg, 0, sizeof *mg);
	if (max_levels < 1) max_levels = 1;
	if (min_rows <= 0) min_rows = g_min_rows;
	if (scale <= 0.0) scale = g_scale;
	mg->A = (GCGE_CSR*)calloc(max_levels, sizeof(GCGE_CSR));
	mg->P = (GCGE_CSR*)calloc(max_levels, sizeof(GCGE_CSR));
	mg->PT = (GCGE_CSR*)calloc(max_levels, sizeof(GCGE_CSR));
	mg->dims = (int (*)[3])calloc(max_levels, sizeof(int[3]));
	if (B != NULL) mg->B = (GCGE_CSR*)calloc(max_levels, sizeof(GCGE_CSR));
	if (!mg->A || !mg->P || !mg->PT || !mg->dims || (B != NULL && !mg->B)) { gcge_mg_free(mg); return -3; }
	mg->A[0] = *A;
	if (B != NULL) mg->B[0] = *B;
	mg->num_levels = 1;
	have_grid = gcge_mg_detect_grid(A, dims, NULL);
	for (l = 0; l + 1 < max_levels; ++l) {
		const GCGE_CSR *Af = &mg->A[l];
		const int nf = Af->nrows;
		int nc, cdims[3] = {0, 0, 0}, *agg, rc;
		if (nf <= min_rows) break;
		agg = (int*)malloc((size_t)nf * sizeof(int));
		if (agg == NULL) { gcge_mg_free(mg); return -3; }
		if (have_grid) {
			mg->dims[l][0] = dims[0]; mg->dims[l][1] = dims[1]; mg->dims[l][2] = dims[2];
			nc = gcge_mg_aggregate_grid(dims, agg, cdims);
		} else {
			nc = gcge_mg_aggregate_graph(Af, g_theta, agg);
		}
		if (nc < 1 || (long)nc * 3 > (long)nf * 2) { free(agg); break; }     /* coarsening stalled */
		rc = gcge_mg_galerkin(Af, agg, nc, scale, &mg->A[l + 1]);
		if (rc == 0 && B != NULL) rc = gcge_mg_galerkin(&mg->B[l], agg, nc, 1.0, &mg->B[l + 1]);
		if (rc == 0) rc = gcge_mg_prolongation(agg, nf, nc, &mg->P[l], &mg->PT[l]);
		free(agg);
		if (rc != 0) { mg->num_levels = l + 2; gcge_mg_free(mg); return rc; }
		mg->num_levels = l + 2;
		if (have_grid) { dims[0] = cdims[0]; dims[1] = cdims[1]; dims[2] = cdims[2]; }
	}
	if (have_grid) { l = mg->num_levels - 1; mg->dims[l][0] = dims[0]; mg->dims[l][1] = dims[1]; mg->dims[l][2] = dims[2]; }
	return 0;
}

/* ---------------------------------------------------------------- row slabs (one rank per GPU)
 * A slab that is whole grid planes coarsens by itself: every rank pairs ITS OWN planes from its first one (z0, z0 + 1), (z0 + 2,
 * z0 + 3), ... — an odd plane count ends in a thinner last cell, like the last plane of an odd grid — so the 2 x 2 x 2 cells of its
 * rows lie inside it, P is local (owned fine rows x owned coarse rows, no halo) and the coarse slab = scale * sum over the cells'
 * rows, columns mapped to GLOBAL coarse indices through the grid and the SHARED partition (a column in a neighbour's planes belongs
 * to the cell the neighbour's pairing puts it in).  With cuts on even plane numbers this is the whole-matrix hierarchy, row for row;
 * with a cut on an odd plane the cells next to it differ from that hierarchy's — still a partition of the rows, still Galerkin
 * (A_{l+1} = scale P^T A_l P with P = the ranks' prolongations stacked diagonally): a hierarchy as good, cut where the non-zeros
 * balance.  The coarse grid has sum_r ceil(planes_r / 2) planes.  The coarse slab goes through the back-end's slab constructor like
 * any row-partitioned matrix (ghost list, halo plan).  Every rank evaluates the SAME stopping rule from the shared partition: a level
 * is coarsened while every cut lies on a plane boundary, every slab holds a plane and some slab holds two.
 * Reference: the MPI back-ends get this from PETSc GAMG / BoomerAMG (app/app_slepc.c:648-728). */
typedef struct { int col; int idx; double val; } SlabEnt;
static int cmp_ent(const void *a, const void *b)
{
	const SlabEnt *x = (const SlabEnt*)a, *y = (const SlabEnt*)b;
	if (x->col != y->col) return (x->col > y->col) - (x->col < y->col);
	return (x->idx > y->idx) - (x->idx < y->idx);
}
/* global fine row -> global coarse row: x, y halved on the grid, z through the owner's pairing (zf[r]: first fine plane of rank r,
 * zc[r]: its first coarse plane; zf[world] = the plane count) */
static inline long slab_agg(long g, const int d[3], int cx, int cy, const long *zf, const long *zc, int world)
{
	const long x = g % d[0], y = (g / d[0]) % d[1], z = g / ((long)d[0] * d[1]);
	int lo = 0, hi = world - 1;
	while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (zf[mid] <= z) lo = mid; else hi = mid - 1; }
	return (x / 2) + (long)cx * ((y / 2) + (long)cy * (zc[lo] + (z - zf[lo]) / 2));
}
static int slab_coarsenable(const int d[3], const long *part, int world)
{
	const long plane = (long)d[0] * d[1]; int r, two = 0;
	if (d[2] < 2) return 0;
	for (r = 0; r < world; ++r) {
		if (part[r] % plane != 0 || part[r + 1] % plane != 0) return 0;
		if (part[r + 1] <= part[r]) return 0;
		if (part[r + 1] - part[r] >= 2 * plane) two = 1;
	}
	return two;
}
int gcge_mg_build_slab(const GCGE_CSR *A, const int dims[3], const long *part, int rank, int world, int max_levels, double scale,
		GCGE_MG *mg, long **part_levels_out)
{
	int l, d[3] = {dims[0], dims[1], dims[2]};
	long *parts, *zf, *zc;
	memset(mg, 0, sizeof *mg);
	if (max_levels < 1) max_levels = 1;
	if (scale <= 0.0) scale = g_scale;
	if (A->row_begin != part[rank] || A->nrows != (int)(part[rank + 1] - part[rank]) || (long)d[0] * d[1] * d[2] != part[world]) return -2;
	mg->A = (GCGE_CSR*)calloc(max_levels, sizeof(GCGE_CSR));
	mg->P = (GCGE_CSR*)calloc(max_levels, sizeof(GCGE_CSR));
	mg->PT = (GCGE_CSR*)calloc(max_levels, sizeof(GCGE_CSR));
	mg->dims = (int (*)[3])calloc(max_levels, sizeof(int[3]));
	parts = (long*)calloc((size_t)max_levels * (world + 1), sizeof(long));
	zf = (long*)calloc(2 * ((size_t)world + 1), sizeof(long)); zc = zf ? zf + world + 1 : NULL;
	if (!mg->A || !mg->P || !mg->PT || !mg->dims || !parts || !zf) { gcge_mg_free(mg); free(parts); free(zf); return -3; }
	mg->A[0] = *A; mg->num_levels = 1;
	memcpy(parts, part, (size_t)(world + 1) * sizeof(long));
	mg->dims[0][0] = d[0]; mg->dims[0][1] = d[1]; mg->dims[0][2] = d[2];
	for (l = 0; l + 1 < max_levels; ++l) {
		const GCGE_CSR *Af = &mg->A[l];
		const long *pf = parts + (size_t)l * (world + 1); long *pc = parts + (size_t)(l + 1) * (world + 1);
		const int cx = (d[0] + 1) / 2, cy = (d[1] + 1) / 2, nf = Af->nrows; int cz;
		const long plane = (long)d[0] * d[1], cplane = (long)cx * cy, rb = pf[rank];
		GCGE_CSR *Ac = &mg->A[l + 1];
		int r, nc, *agg, *ptr = NULL, *mem = NULL, I, rc = 0; int64_t tot = 0;
		if (!slab_coarsenable(d, pf, world)) break;
		zc[0] = 0;
		for (r = 0; r <= world; ++r) zf[r] = pf[r] / plane;
		for (r = 0; r < world; ++r) zc[r + 1] = zc[r] + (zf[r + 1] - zf[r] + 1) / 2;       /* every rank pairs its own planes */
		for (r = 0; r <= world; ++r) pc[r] = zc[r] * cplane;
		cz = (int)zc[world];
		if (zf[world] != d[2] || cz >= d[2]) { rc = -2; goto fail; }
		nc = (int)(pc[rank + 1] - pc[rank]);
		agg = (int*)malloc((size_t)(nf > 0 ? nf : 1) * sizeof(int));       /* LOCAL coarse row of every owned fine row */
		if (agg == NULL) { rc = -3; goto fail; }
		for (r = 0; r < nf; ++r) agg[r] = (int)(slab_agg(rb + r, d, cx, cy, zf, zc, world) - pc[rank]);
		if ((rc = gcge_mg_prolongation(agg, nf, nc, &mg->P[l], &mg->PT[l])) != 0) { free(agg); goto fail; }
		if (aggregate_members(agg, nf, nc, &ptr, &mem) != 0) { free(agg); rc = -3; goto fail; }
		/* coarse slab: per coarse row the entries of its members, columns -> global coarse index, merged in arrival order */
		Ac->nrows = nc; Ac->ncols = (int)pc[world]; Ac->row_begin = (int)pc[rank];
		Ac->rowptr = (int*)calloc((size_t)nc + 1, sizeof(int));
		{
			int64_t cap = 0; SlabEnt *buf; int maxlen = 0;
			for (I = 0; I < nc; ++I) { int q, len = 0; for (q = ptr[I]; q < ptr[I + 1]; ++q) len += Af->rowptr[mem[q] + 1] - Af->rowptr[mem[q]]; cap += len; if (len > maxlen) maxlen = len; }
			Ac->colidx = (int*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(int));       /* upper bound: no merging */
			Ac->val = (double*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(double));
			buf = (SlabEnt*)malloc((size_t)(maxlen > 0 ? maxlen : 1) * sizeof(SlabEnt));
			if (!Ac->rowptr || !Ac->colidx || !Ac->val || !buf) { free(buf); free(agg); free(ptr); free(mem); rc = -3; goto fail; }
			for (I = 0; I < nc; ++I) {
				int q, k, len = 0, a;
				for (q = ptr[I]; q < ptr[I + 1]; ++q) {
					const int fr = mem[q];
					for (k = Af->rowptr[fr]; k < Af->rowptr[fr + 1]; ++k) {
						buf[len].col = (int)slab_agg(Af->colidx[k], d, cx, cy, zf, zc, world); buf[len].idx = len; buf[len].val = Af->val[k]; ++len;
					}
				}
				qsort(buf, len, sizeof(SlabEnt), cmp_ent);
				for (a = 0; a < len; ) {
					int b = a; double sum = 0.0;
					while (b < len && buf[b].col == buf[a].col) { sum += buf[b].val; ++b; }
					Ac->colidx[tot] = buf[a].col; Ac->val[tot] = sum * scale; ++tot;
					a = b;
				}
				Ac->rowptr[I + 1] = (int)tot;
			}
			free(buf);
		}
		Ac->nnz = tot;
		free(agg); free(ptr); free(mem);
		mg->num_levels = l + 2;
		d[0] = cx; d[1] = cy; d[2] = cz;
		mg->dims[l + 1][0] = cx; mg->dims[l + 1][1] = cy; mg->dims[l + 1][2] = cz;
		continue;
fail:
		mg->num_levels = l + 2; gcge_mg_free(mg); free(parts); free(zf); return rc;
	}
	free(zf);
	*part_levels_out = parts;
	return 0;
}
