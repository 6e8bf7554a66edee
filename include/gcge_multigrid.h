/* gcge_multigrid.h — host side of the multigrid hierarchy behind ops->MultiGridCreate.
 *
 * The reference's multigrid solver (BlockAlgebraicMultiGrid, src/ops_lin_sol.c:466-715) takes its hierarchy
 *     A_0 = A,  A_{l+1} ~ P_l^T A_l P_l,   P_l : level l+1 -> level l
 * from the back-end through the MultiGridCreate slot (src/ops.h:134-139; app/app_slepc.c:648-728 asks PETSc GAMG,
 * app/app_hypre.c BoomerAMG, app/app_lapack.c:863-929 builds a fixed 1-D toy).  This header is what OUR back-ends build
 * it from: plain aggregation on the host CSR arrays —
 *   - a lexicographic nx x ny x nz grid read off the rows' column offsets is coarsened 2 x 2 x 2 (cell-centred);
 *   - any other symmetric matrix by greedy aggregation over its strong couplings (Vanek / Mandel / Brezina);
 *   - P is the piecewise-constant prolongation of the aggregates (one 1.0 per row), A_{l+1} = scale * P^T A_l P.
 * scale = 1 is the Galerkin operator.  Piecewise-constant P over-estimates the energy of a smooth coarse function by 2
 * for second-order operators whatever the dimension (only the jumps across aggregate faces count), so the correction a
 * Galerkin coarse problem returns is half of what it should be; scale = 0.5 (the default of the back-ends) is the
 * classical over-correction (Braess 1995) folded into the coarse operator — BlockAMG itself adds the correction with
 * factor 1 (src/ops_lin_sol.c:626-640), so the hierarchy is the only place for it.
 * Everything here is host C; the back-ends wrap the levels into their own matrix handles.
 */
#ifndef GCGE_MULTIGRID_H
#define GCGE_MULTIGRID_H

#include "gcge_problems.h"

#ifdef __cplusplus
extern "C" {
#endif

/* nx, ny, nz of a lexicographic grid (index x + nx (y + ny z)) whose neighbour couplings the rows of A show: +-1 ... +-arm
 * along a line, multiples of nx between lines, of nx ny between planes (1-D and 2-D grids: ny and / or nz = 1).
 * Returns 1 and fills dims[3] (and *arm, may be NULL), 0 when the rows show no such grid.                          */
int gcge_mg_detect_grid (const GCGE_CSR *A, int dims[3], int *arm);     /* also: a slab of rows with GLOBAL columns (ncols = global size) */

/* 2 x 2 x 2 aggregates of an nx x ny x nz grid (the last aggregate of an odd direction holds one layer):
 * agg[r] = coarse index of row r, cdims = coarse grid.  Returns the number of aggregates.                          */
int gcge_mg_aggregate_grid (const int dims[3], int *agg, int cdims[3]);

/* greedy aggregation over the strong couplings |a_ij| >= theta * max_k |a_ik| (k != i): pass 1 forms an aggregate from
 * every node whose strong neighbours are all still free, pass 2 attaches the rest to the neighbouring aggregate they are
 * coupled to most strongly, pass 3 turns what is left (isolated rows) into aggregates of their own.
 * Returns the number of aggregates.                                                                                */
int gcge_mg_aggregate_graph (const GCGE_CSR *A, double theta, int *agg);

/* Ac = scale * P^T A P for the piecewise-constant P of `agg` (nc aggregates): nc x nc CSR, ascending columns.
 * Rows of the sum are accumulated in ascending fine-row order, entries of a row in storage order (deterministic).   */
int gcge_mg_galerkin (const GCGE_CSR *A, const int *agg, int nc, double scale, GCGE_CSR *Ac);

/* P (nf x nc, one entry 1.0 per row) and its transpose (nc x nf, ascending columns) as CSR                         */
int gcge_mg_prolongation (const int *agg, int nf, int nc, GCGE_CSR *P, GCGE_CSR *PT);

/* the whole hierarchy: level 0 is A itself (not copied: A[0] aliases the caller's arrays and is never freed here)   */
typedef struct GCGE_MG_ {
	int      num_levels;
	GCGE_CSR *A;          /* [num_levels]      A[0] = the caller's matrix                                  */
	GCGE_CSR *B;          /* [num_levels] or NULL (B == NULL): Galerkin P^T B P, never rescaled            */
	GCGE_CSR *P;          /* [num_levels - 1]  P[l]: rows(A[l]) x rows(A[l+1])                              */
	GCGE_CSR *PT;         /* [num_levels - 1]  the transposes                                               */
	int      (*dims)[3];  /* [num_levels] grid of the level, {0,0,0}: aggregated over the graph             */
} GCGE_MG;
/* Builds at most max_levels levels and stops early when a level has no more than min_rows rows or coarsening stalls
 * (fewer than 1.5 x fewer rows).  Returns 0 and fills mg (mg->num_levels >= 1), -3 out of memory.                  */
int  gcge_mg_build (const GCGE_CSR *A, const GCGE_CSR *B, int max_levels, int min_rows, double scale, GCGE_MG *mg);
void gcge_mg_free (GCGE_MG *mg);

/* The hierarchy of ONE row slab (one rank per GPU): A holds rows [part[rank], part[rank + 1]) with GLOBAL columns of a matrix on the
 * lexicographic grid `dims`; every slab is whole planes.  Every rank pairs ITS OWN planes from its first one (an odd count ends in a
 * thinner cell), columns in a neighbour's planes follow the neighbour's pairing through the shared partition: with cuts on even planes
 * the levels are the whole-matrix hierarchy's rows, with a cut on an odd plane the cells next to it are the rank's own — Galerkin
 * either way.  A level is coarsened while every slab holds a plane and some slab two — every rank evaluates that from the shared
 * partition, so all ranks build the same number of levels; level l + 1 has sum_r ceil(planes_r / 2) planes.  mg->A[l] (l >= 1): the coarse slab with GLOBAL coarse columns, row_begin =
 * part_levels[l * (world + 1) + rank]; mg->P[l] / PT[l]: local (owned fine rows x owned coarse rows: the cells of a slab lie inside
 * it).  part_levels (free with free()): the row partition of every level.  Returns 0; -2: the slab is not whole planes of `dims`. */
int gcge_mg_build_slab (const GCGE_CSR *A, const int dims[3], const long *part, int rank, int world, int max_levels, double scale,
		GCGE_MG *mg, long **part_levels_out);

/* process-wide defaults the back-ends' MultiGridCreate slots use (the slot's signature has no room for them,
 * src/ops.h:134): scale as above (default 0.5), min_rows (default 64), theta of the graph aggregation (default 0.25) */
void gcge_mg_set_defaults (double scale, int min_rows, double theta);
void gcge_mg_get_defaults (double *scale, int *min_rows, double *theta);

#ifdef __cplusplus
}
#endif
#endif
