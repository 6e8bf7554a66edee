/* TestAppHIP — the driver a GCGE maintainer adds next to TestAppCCS (reference test/test_app_ccs.c:86-140, called from
 * test/main.c:40-49), compiled and exported (INTEGRATION.md shows the same file written against the reference's
 * own headers).  Same shape as the reference's: create the table, OPS_HIP_Set instead of OPS_CCS_Set, OPS_Setup, build the
 * matrices on the host, hand them to the back-end, TestEigenSolverGCG(A, B, flag, argc, argv, ops), tear down.
 *
 * Plain C against the C ABI; linked into libgcge_hip.so (it needs OPS_HIP_Set).  TestEigenSolverGCG / OPS_Create / OPS_Setup
 * are whatever the program links: libgcge_host.so's, or the reference's own objects (same names, identical table layout).
 *
 * Matrices: by default the reference's stock pair (test_app_ccs.c:142-184: 1-D linear FE, n = 807: A = tridiag(-1, 2, -1) / h,
 * B = h I).  Options (read with ops->GetOptionFromCommandLine like every other option of the harness):
 *   -hip_problem  fe1d | lap3d | fe3d | sio2     generator of include/gcge_problems.h     (default fe1d)
 *   -hip_size     n (fe1d) or grid points per direction                                    (default 807 / 32)
 *   -hip_petsc_A  file, -hip_petsc_B file        PETSc binary Mat files (test_app_slepc.c:416-445; B optional)
 *   -hip_mtx_A    file, -hip_mtx_B file          Matrix Market coordinate files (the form SuiteSparse ships SiO2 & co. in)
 *   -hip_flag     0 | 1    0: the solver stack's BlockPCG over the slots; 1: the back-end's fused block CG behind
 *                          ops->MultiLinearSolver (the hook test_app_ccs.c:109-120 uses for UMFPACK)       (default 0)
 *   -hip_device   d                                                                         (default 0)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gcge_hip.h"
#include "gcge_solver.h"

int TestEigenSolverGCG(void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops);

static int load_pair(struct OPS_ *ops, int argc, char **argv, GCGE_CSR *A, GCGE_CSR *B, int *haveB)
{
	char problem[64] = "fe1d", fileA[1024] = "", fileB[1024] = "", mtxA[1024] = "", mtxB[1024] = "";
	int size = -1, rc;
	ops->GetOptionFromCommandLine("-hip_problem", 's', problem, argc, argv, ops);
	ops->GetOptionFromCommandLine("-hip_size", 'i', &size, argc, argv, ops);
	ops->GetOptionFromCommandLine("-hip_petsc_A", 's', fileA, argc, argv, ops);
	ops->GetOptionFromCommandLine("-hip_petsc_B", 's', fileB, argc, argv, ops);
	ops->GetOptionFromCommandLine("-hip_mtx_A", 's', mtxA, argc, argv, ops);
	ops->GetOptionFromCommandLine("-hip_mtx_B", 's', mtxB, argc, argv, ops);
	*haveB = 0;
	memset(A, 0, sizeof *A); memset(B, 0, sizeof *B);
	if (fileA[0] != 0) {
		if ((rc = gcge_load_petsc_binary(fileA, 0, -1, A)) != 0) { ops->Printf("TestAppHIP: cannot read %s (%d)\n", fileA, rc); return rc; }
		if (fileB[0] != 0) {
			if ((rc = gcge_load_petsc_binary(fileB, 0, -1, B)) != 0) { ops->Printf("TestAppHIP: cannot read %s (%d)\n", fileB, rc); return rc; }
			*haveB = 1;
		}
		return 0;
	}
	if (mtxA[0] != 0) {
		if ((rc = gcge_load_matrix_market(mtxA, A)) != 0) { ops->Printf("TestAppHIP: cannot read %s (%d)\n", mtxA, rc); return rc; }
		if (mtxB[0] != 0) {
			if ((rc = gcge_load_matrix_market(mtxB, B)) != 0) { ops->Printf("TestAppHIP: cannot read %s (%d)\n", mtxB, rc); return rc; }
			*haveB = 1;
		}
		return 0;
	}
	if (0 == strcmp(problem, "fe1d")) { rc = gcge_problem_fe1d(size > 0 ? size : 800 + 7, A, B); *haveB = 1; }
	else if (0 == strcmp(problem, "lap3d")) rc = gcge_problem_lap3d(size > 0 ? size : 32, 0, -1, A);
	else if (0 == strcmp(problem, "fe3d")) { rc = gcge_problem_fe3d(size > 0 ? size : 32, 0, -1, A, B); *haveB = 1; }
	else if (0 == strcmp(problem, "sio2")) rc = gcge_problem_sio2_like(size > 0 ? size : 24, 8, 1.5, 3.0, 12345ULL, 0, -1, A);
	else { ops->Printf("TestAppHIP: unknown -hip_problem %s\n", problem); return -9; }
	return rc;
}

int TestAppHIP(int argc, char *argv[])
{
	OPS *ops = NULL;
	GCGE_CSR csrA, csrB;
	GCGE_HIP_MAT *A = NULL, *B = NULL;
	int flag = 0, device = 0, haveB = 0, rc;

	int k;
	for (k = 0; k + 1 < argc; ++k) if (argv[k] != NULL && 0 == strcmp(argv[k], "-hip_device")) device = atoi(argv[k + 1]);
	if (gcge_hip_init(device) != 0) return 1;     /* (before the table exists: OPS_HIP_Set needs the device) */
	OPS_Create(&ops);
	OPS_HIP_Set(ops);           /* instead of OPS_CCS_Set (app/app_ccs.c:213-249) */
	OPS_Setup(ops);

	if ((rc = load_pair(ops, argc, argv, &csrA, &csrB, &haveB)) != 0) { OPS_Destroy(&ops); return rc; }
	/* symmetric matrices: the CSR arrays are the CCS triple (j_col, i_row, data) of app/app_ccs.h:20-24 */
	A = gcge_hip_mat_create(csrA.nrows, csrA.nrows, 0, csrA.rowptr, csrA.colidx, csrA.val);
	if (haveB) B = gcge_hip_mat_create(csrB.nrows, csrB.nrows, 0, csrB.rowptr, csrB.colidx, csrB.val);
	if (A == NULL || (haveB && B == NULL)) { ops->Printf("TestAppHIP: matrix upload failed\n"); OPS_Destroy(&ops); return 2; }

	ops->GetOptionFromCommandLine("-hip_flag", 'i', &flag, argc, argv, ops);
	if (flag >= 1) gcge_hip_bpcg_setup(ops, 30, 1e-2, 1e-14, "abs");     /* its slot and workspace: test_app_ccs.c:112-120 */
	rc = TestEigenSolverGCG((void*)A, (void*)B, flag, argc, argv, ops);
	if (flag >= 1) gcge_hip_bpcg_release(ops);

	gcge_hip_mat_destroy(A);
	if (B != NULL) gcge_hip_mat_destroy(B);
	gcge_csr_free(&csrA);
	if (haveB) gcge_csr_free(&csrB);
	OPS_Destroy(&ops);
	return rc;
}
