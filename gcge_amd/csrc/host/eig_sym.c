/* Small dense symmetric eigensolver (host) — the Rayleigh–Ritz kernel K7.
 *
 * Takes the place of LAPACK dsyevx('V','A','U') at ops_eig_sol_gcg.c:1201-1203 and
 * of dsyev('V','L') at ops_orth.c:144 in the reference (the reference pins no
 * LAPACK version and none is guaranteed on the target box — SURVEY.md §8c).
 *
 * Method: Householder reduction to tridiagonal form with accumulation of the
 * transformation, then implicit-shift QL iterations on the tridiagonal matrix
 * with the rotations applied to the accumulated basis, then an ascending sort.
 * All eigenpairs, orthonormal eigenvectors to O(n eps).  N <= a few hundred here
 * ((V-C) <= 656), so an O(n^3) unblocked host implementation is adequate; the
 * inner loops run over contiguous columns of the column-major basis.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#include "gcge_solver.h"

/* z (n x n, ld ldz, column-major) holds the symmetric matrix on entry (only the
 * LOWER triangle is read), the orthogonal reduction matrix on exit;
 * d = diagonal, e = sub-diagonal (e[0] = 0). */
static void tridiagonalise(int n, double *z, int ldz, double *d, double *e)
{
#define Z(i,j) z[(size_t)(j) * ldz + (i)]
	int i, j, k, l;
	double f, g, h, hh, scale;
	/* work on rows of the lower triangle, from the last row upwards */
	for (i = n - 1; i >= 1; --i) {
		l = i - 1; h = 0.0; scale = 0.0;
		if (l > 0) {
			for (k = 0; k <= l; ++k) scale += fabs(Z(i, k));
			if (scale == 0.0) {
				e[i] = Z(i, l);
			} else {
				for (k = 0; k <= l; ++k) { Z(i, k) /= scale; h += Z(i, k) * Z(i, k); }
				f = Z(i, l);
				g = (f >= 0.0) ? -sqrt(h) : sqrt(h);
				e[i] = scale * g;
				h -= f * g;
				Z(i, l) = f - g;
				f = 0.0;
				for (j = 0; j <= l; ++j) {
					Z(j, i) = Z(i, j) / h;          /* store u/H in column i */
					g = 0.0;
					for (k = 0; k <= j; ++k)     g += Z(j, k) * Z(i, k);
					for (k = j + 1; k <= l; ++k) g += Z(k, j) * Z(i, k);
					e[j] = g / h;
					f += e[j] * Z(i, j);
				}
				hh = f / (h + h);
				for (j = 0; j <= l; ++j) {
					f = Z(i, j);
					e[j] = g = e[j] - hh * f;
					for (k = 0; k <= j; ++k) Z(j, k) -= (f * e[k] + g * Z(i, k));
				}
			}
		} else {
			e[i] = Z(i, l);
		}
		d[i] = h;
	}
	d[0] = 0.0; e[0] = 0.0;
	/* accumulate the transformation */
	for (i = 0; i < n; ++i) {
		l = i - 1;
		if (d[i] != 0.0) {
			for (j = 0; j <= l; ++j) {
				g = 0.0;
				for (k = 0; k <= l; ++k) g += Z(i, k) * Z(k, j);
				for (k = 0; k <= l; ++k) Z(k, j) -= g * Z(k, i);
			}
		}
		d[i] = Z(i, i);
		Z(i, i) = 1.0;
		for (j = 0; j <= l; ++j) { Z(j, i) = 0.0; Z(i, j) = 0.0; }
	}
#undef Z
}

/* implicit QL on (d,e); rotations applied to the columns of z.  Returns 0, or
 * l+1 if eigenvalue l failed to converge in 60 sweeps. */
static int ql_implicit(int n, double *d, double *e, double *z, int ldz)
{
	int i, k, l, m, iter;
	double s, r, p, g, f, dd, c, b;
	for (i = 1; i < n; ++i) e[i - 1] = e[i];
	e[n - 1] = 0.0;
	for (l = 0; l < n; ++l) {
		iter = 0;
		do {
			for (m = l; m < n - 1; ++m) {
				dd = fabs(d[m]) + fabs(d[m + 1]);
				if (fabs(e[m]) <= DBL_EPSILON * dd) break;
			}
			if (m != l) {
				if (iter++ == 60) return l + 1;
				g = (d[l + 1] - d[l]) / (2.0 * e[l]);
				r = hypot(g, 1.0);
				g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
				s = c = 1.0; p = 0.0;
				for (i = m - 1; i >= l; --i) {
					f = s * e[i]; b = c * e[i];
					e[i + 1] = r = hypot(f, g);
					if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
					s = f / r; c = g / r;
					g = d[i + 1] - p;
					r = (d[i] - g) * s + 2.0 * c * b;
					d[i + 1] = g + (p = s * r);
					g = c * r - b;
					{   /* rotate columns i and i+1 of z (contiguous) */
						double *zi = z + (size_t)i * ldz, *zi1 = z + (size_t)(i + 1) * ldz;
						for (k = 0; k < n; ++k) {
							f = zi1[k];
							zi1[k] = s * zi[k] + c * f;
							zi[k]  = c * zi[k] - s * f;
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
 * Returns 0 on success. */
int GCGE_SymEig(char uplo, int n, const double *a, int lda, double *w,
		double *z, int ldz, double *work)
{
	int i, j, k, info;
	double *e = work, *tmp = work + n;
	if (n <= 0) return 0;
	/* copy the referenced triangle into the LOWER triangle of z */
	for (j = 0; j < n; ++j)
		for (i = j; i < n; ++i)
			z[(size_t)j * ldz + i] = (uplo == 'U' || uplo == 'u')
				? a[(size_t)i * lda + j] : a[(size_t)j * lda + i];
	if (n == 1) { w[0] = z[0]; z[0] = 1.0; return 0; }
	tridiagonalise(n, z, ldz, w, e);
	info = ql_implicit(n, w, e, z, ldz);
	if (info) return info;
	/* selection sort of the eigenpairs, ascending */
	for (i = 0; i < n - 1; ++i) {
		double p = w[k = i];
		for (j = i + 1; j < n; ++j) if (w[j] < p) p = w[k = j];
		if (k != i) {
			w[k] = w[i]; w[i] = p;
			memcpy(tmp, z + (size_t)i * ldz, n * sizeof(double));
			memcpy(z + (size_t)i * ldz, z + (size_t)k * ldz, n * sizeof(double));
			memcpy(z + (size_t)k * ldz, tmp, n * sizeof(double));
		}
	}
	return 0;
}
