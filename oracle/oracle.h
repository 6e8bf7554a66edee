/* TEST INFRASTRUCTURE — public interface of the CPU oracle (oracle/cpu_backend.c). */
#ifndef GCGE_ORACLE_H
#define GCGE_ORACLE_H
#include "gcge_ops.h"
#ifdef __cplusplus
extern "C" {
#endif
/* same layouts as the reference's LAPACKVEC (app/app_lapack.h:17-20) and CCSMAT (app/app_ccs.h:20-24) */
typedef struct ORACLE_VEC_ { double *data; int nrows; int ncols; int ldd; } ORACLE_VEC;
typedef struct ORACLE_CCS_ { double *data; int *i_row; int *j_col; int nrows; int ncols; } ORACLE_CCS;
/* halo plan of a row slab: local rows to ship (grouped by destination rank) and the exchange callback */
typedef struct ORACLE_HALO_ {
	int nsend; int *send_rows;
	void (*exchange) (double *sendbuf, double *recvbuf, int ncols, void *ctx);   /* HOST buffers, row-major x ncols */
	void *ctx;
} ORACLE_HALO;
void oracle_set_halo (const ORACLE_HALO *h);
void oracle_set_partition (long row_begin, long nglobal);   /* for the rank-count independent random start block */
void OPS_ORACLE_Set (struct OPS_ *ops);     /* counterpart of OPS_CCS_Set (app/app_ccs.c:213-249) */
void oracle_set_threads (int n);            /* OpenMP threads over block columns (app_ccs.c:117) */
int  oracle_get_threads (void);
#ifdef __cplusplus
}
#endif
#endif
