/* Small dense symmetric eigensolver (host) — the Rayleigh–Ritz kernel K7.
 *
 * Takes the place of LAPACK dsyevx('V','A','U') at ops_eig_sol_gcg.c:1201-1203 and
 * of dsyev('V','L') at ops_orth.c:144 in the reference (the reference pins no
 * LAPACK version and none is guaranteed on the target box — SURVEY.md §8c).
 *
 * Method: Householder reduction to tridiagonal form, implicit-shift QL iterations
 * on the tridiagonal matrix, rotations applied to the accumulated basis, ascending
 * sort.  All eigenpairs, orthonormal eigenvectors to O(n eps).
 *
 * N reaches 512 at BASELINE config 3 (nevMax 256 + 2 x 128), where an unblocked
 * single-thread version costs 0.4 s per GCG iteration — more than all device work.
 * Therefore every O(n^2) inner step runs over contiguous columns (6x faster than the
 * textbook row-oriented form at n = 512: 64 ms on the MI355X host) and can be shared among
 * host threads (OpenMP, GCGE_EIG_THREADS; default 1 — on the GPU boxes of this pool extra
 * threads made it slower, profiles/r01_dense/06_host_eig_threads.log):
 *   - the reduction works on the FULL symmetric matrix (both triangles), so the
 *     matrix-vector product and the rank-2 update are column sweeps;
 *   - the rotations of one QL sweep are recorded and then applied by every thread to
 *     its own block of rows of the column-major basis (rows are independent; the inner
 *     loop runs down two contiguous column segments and vectorises).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "gcge_solver.h"

static int eig_threads(int n)
{
	static int cached = 0;
	if (cached == 0) {
		const char *s = getenv("GCGE_EIG_THREADS");
		int t = s ? atoi(s) : 1;
#ifdef _OPENMP
		if (t > omp_get_num_procs()) t = omp_get_num_procs();
#else
		t = 1;
#endif
		cached = t < 1 ? 1 : t;
	}
	return n < 96 ? 1 : cached;   /* fork/join costs more than it saves on tiny problems */
}

/* m: full symmetric n x n, column-major, ld n (destroyed).  d, e: diagonal and sub-diagonal
 * (e[k] couples k and k+1, e[n-1] = 0).  q: n x n column-major orthogonal Q with A = Q T Q^T.
 * v, p: work vectors of length n. */
static void tridiagonalise(int n, double *m, double *d, double *e, double *q, double *v, double *p, int nt)
{
#define M(i,j) m[(size_t)(j) * n + (i)]
	int k, i, j;
	double *beta = e;   /* e[k] is written after beta[k] has been consumed: keep them apart */
	double *betas = (double*)malloc((size_t)n * sizeof(double));
	for (k = 0; k < n - 2; ++k) {
		const int len = n - k - 1;            /* reflector acts on rows/cols k+1 .. n-1 */
		double *x = &M(k + 1, k);
		double sigma = 0.0, scale = 0.0, x0, mu, vtv, K;
		for (i = 0; i < len; ++i) scale += fabs(x[i]);
		d[k] = M(k, k);
		if (scale == 0.0 || len == 1) {       /* nothing to annihilate */
			e[k] = x[0]; betas[k] = 0.0; M(k + 1, k) = 0.0;
			if (len == 1) { /* keep the trailing 2x2 as it is */ }
			for (i = 0; i < len; ++i) M(k + 1 + i, k) = 0.0;   /* v = 0 */
			continue;
		}
		for (i = 0; i < len; ++i) { v[i] = x[i] / scale; sigma += v[i] * v[i]; }
		x0 = v[0];
		mu = (x0 >= 0.0) ? -sqrt(sigma) : sqrt(sigma);
		e[k] = scale * mu;
		v[0] = x0 - mu;
		vtv = sigma - 2.0 * x0 * mu + mu * mu;  /* = sigma - x0^2 + (x0-mu)^2 */
		vtv = sigma - x0 * x0 + v[0] * v[0];
		betas[k] = 2.0 / vtv;
		for (i = 0; i < len; ++i) M(k + 1 + i, k) = v[i];      /* keep v in the annihilated column */
		/* p = beta * A22 v  (column dots, A22 symmetric and fully stored) */
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
		for (j = 0; j < len; ++j) {
			const double *col = &M(k + 1, k + 1 + j);
			double s = 0.0; int ii;
			for (ii = 0; ii < len; ++ii) s += col[ii] * v[ii];
			p[j] = betas[k] * s;
		}
		K = 0.0;
		for (i = 0; i < len; ++i) K += v[i] * p[i];
		K *= 0.5 * betas[k];
		for (i = 0; i < len; ++i) p[i] -= K * v[i];            /* w */
		/* A22 -= v w^T + w v^T */
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
		for (j = 0; j < len; ++j) {
			double *col = &M(k + 1, k + 1 + j);
			const double wj = p[j], vj = v[j]; int ii;
			for (ii = 0; ii < len; ++ii) col[ii] -= v[ii] * wj + p[ii] * vj;
		}
	}
	if (n >= 2) { d[n - 2] = M(n - 2, n - 2); e[n - 2] = M(n - 1, n - 2); }
	d[n - 1] = M(n - 1, n - 1); e[n - 1] = 0.0;
	(void)beta;
	/* Q = H_0 H_1 ... H_{n-3}: start from I and apply H_k from the left, k = n-3 .. 0 (column sweeps) */
	for (j = 0; j < n; ++j) { memset(q + (size_t)j * n, 0, (size_t)n * sizeof(double)); q[(size_t)j * n + j] = 1.0; }
	for (k = n - 3; k >= 0; --k) {
		const int len = n - k - 1;
		const double *vk = &M(k + 1, k);
		const double bk = betas[k];
		if (bk == 0.0) continue;
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
		for (j = 0; j < len; ++j) {
			double *col = q + (size_t)(k + 1 + j) * n + (k + 1);
			double t = 0.0; int ii;
			for (ii = 0; ii < len; ++ii) t += vk[ii] * col[ii];
			t *= bk;
			for (ii = 0; ii < len; ++ii) col[ii] -= t * vk[ii];
		}
	}
	free(betas);
#undef M
}

/* implicit QL on (d,e) with e[k] coupling k and k+1; the rotations of a sweep are recorded and then applied to
 * columns i, i+1 of the column-major q, each thread on its own block of rows.  Returns 0, or l+1 if
 * eigenvalue l failed to converge in 60 sweeps. */
static int ql_implicit(int n, double *d, double *e, double *q, double *cs, int nt)
{
	int i, l, m, iter;
	double s, r, p, g, f, dd, c, b;
	for (l = 0; l < n; ++l) {
		iter = 0;
		do {
			for (m = l; m < n - 1; ++m) {
				dd = fabs(d[m]) + fabs(d[m + 1]);
				if (fabs(e[m]) <= DBL_EPSILON * dd) break;
			}
			if (m != l) {
				int first, last, blk;
				const int rb = ((n + nt - 1) / nt + 7) / 8 * 8, nblk = (n + rb - 1) / rb;
				if (iter++ == 60) return l + 1;
				g = (d[l + 1] - d[l]) / (2.0 * e[l]);
				r = hypot(g, 1.0);
				g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
				s = c = 1.0; p = 0.0;
				first = m - 1; last = m;     /* rotations recorded for i = first down to last (last > first: none) */
				for (i = m - 1; i >= l; --i) {
					f = s * e[i]; b = c * e[i];
					e[i + 1] = r = hypot(f, g);
					if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
					s = f / r; c = g / r;
					g = d[i + 1] - p;
					r = (d[i] - g) * s + 2.0 * c * b;
					d[i + 1] = g + (p = s * r);
					g = c * r - b;
					cs[2 * i] = c; cs[2 * i + 1] = s; last = i;
				}
				if (last <= first) {
					/* a sweep is (first-last+1) rotations x n rows x 6 flops: only long sweeps repay a fork/join */
					const int par = nt > 1 && (long)(first - last + 1) * n > 150000;
#pragma omp parallel for num_threads(nt) schedule(static) if (par)
					for (blk = 0; blk < nblk; ++blk) {
						const int r0 = blk * rb, r1 = (r0 + rb < n) ? r0 + rb : n; int ii, rr;
						for (ii = first; ii >= last; --ii) {
							const double cc = cs[2 * ii], ss = cs[2 * ii + 1];
							double *zi = q + (size_t)ii * n, *zi1 = zi + n;
							for (rr = r0; rr < r1; ++rr) {
								const double ff = zi1[rr];
								zi1[rr] = ss * zi[rr] + cc * ff;
								zi[rr]  = cc * zi[rr] - ss * ff;
							}
						}
					}
				}
				if (r == 0.0 && i >= l) continue;
				d[l] -= p; e[l] = g; e[m] = 0.0;
			}
		} while (m != l);
	}
	return 0;
}

/* All eigenpairs of the symmetric n x n matrix a (column-major, ld lda; only the
 * triangle named by uplo is read).  w: eigenvalues ascending; z (ld ldz): the
 * matching orthonormal eigenvectors.  a is NOT modified.  work: >= 2n doubles.
 * Returns 0 on success (-1: out of memory). */
/* A back-end may offer the same computation on its device for n >= min_n (GCGE_SetSymEigHook, gcge_solver.h): the HIP
 * back-end registers csrc/hip/eig_device.hip.  The hook belongs to ONE operator table: it is keyed on `owner` (the table's
 * MultiVecLinearComb slot, as the other capabilities of gcge_ops.h), and only solver calls made on behalf of that table
 * (GCGE_SymEigFor) reach the device — the CPU oracle and the host-only entry point GCGE_SymEig never do, whatever was
 * registered in the process.  GCGE_EIG_HOST=1 keeps everything on the host. */
static GCGE_SYMEIG_FN g_eig_hook = NULL; static int g_eig_hook_min_n = 0; static void *g_eig_owner = NULL;
void GCGE_SetSymEigHook(GCGE_SYMEIG_FN fn, int min_n, void *owner) { g_eig_hook = fn; g_eig_hook_min_n = min_n; g_eig_owner = owner; }
int GCGE_SymEigHost(char uplo, int n, const double *a, int lda, double *w, double *z, int ldz, double *work);

int GCGE_SymEig(char uplo, int n, const double *a, int lda, double *w,
		double *z, int ldz, double *work)
{
	return GCGE_SymEigHost(uplo, n, a, lda, w, z, ldz, work);
}

int GCGE_SymEigFor(void *owner, char uplo, int n, const double *a, int lda, double *w,
		double *z, int ldz, double *work)
{
	if (g_eig_hook != NULL && owner != NULL && owner == g_eig_owner && n >= g_eig_hook_min_n && getenv("GCGE_EIG_HOST") == NULL) {
		const int info = g_eig_hook(uplo, n, a, lda, w, z, ldz);
		if (info == 0) return 0;            /* otherwise: fall through to the host solver */
	}
	return GCGE_SymEigHost(uplo, n, a, lda, w, z, ldz, work);
}

int GCGE_SymEigHost(char uplo, int n, const double *a, int lda, double *w,
		double *z, int ldz, double *work)
{
	int i, j, info, nt;
	double *m, *q, *e, *v, *p, *cs;
	int *perm;
	const int upper = (uplo == 'U' || uplo == 'u');
	if (n <= 0) return 0;
	if (n == 1) { w[0] = a[0]; z[0] = 1.0; return 0; }
	(void)work;
	m  = (double*)malloc(((size_t)2 * n * n + (size_t)6 * n) * sizeof(double));
	perm = (int*)malloc((size_t)n * sizeof(int));
	if (m == NULL || perm == NULL) { free(m); free(perm); return -1; }
	q = m + (size_t)n * n; e = q + (size_t)n * n; v = e + n; p = v + n; cs = p + n;   /* cs: 2n */
	/* full symmetric copy from the referenced triangle */
	for (j = 0; j < n; ++j)
		for (i = j; i < n; ++i) {
			const double t = upper ? a[(size_t)i * lda + j] : a[(size_t)j * lda + i];
			m[(size_t)j * n + i] = t; m[(size_t)i * n + j] = t;
		}
	nt = eig_threads(n);
	tridiagonalise(n, m, w, e, q, v, p, nt);
	info = ql_implicit(n, w, e, q, cs, nt);
	if (info == 0) {
		/* ascending order (stable insertion on an index permutation), then gather the columns into z */
		for (i = 0; i < n; ++i) perm[i] = i;
		for (i = 1; i < n; ++i) {
			const int pi = perm[i]; const double key = w[pi];
			for (j = i - 1; j >= 0 && w[perm[j]] > key; --j) perm[j + 1] = perm[j];
			perm[j + 1] = pi;
		}
		for (i = 0; i < n; ++i) v[i] = w[perm[i]];
		memcpy(w, v, (size_t)n * sizeof(double));
		for (j = 0; j < n; ++j) {
			memcpy(z + (size_t)j * ldz, q + (size_t)perm[j] * n, (size_t)n * sizeof(double));
		}
	}
	free(m); free(perm);
	return info;
}
