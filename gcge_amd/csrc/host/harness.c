/* Eigensolver harness: allocates the workspace, sets the parameters, seeds the
 * generator, runs ops->EigenSolver and reports.  Same parameter flow and
 * defaults as the reference's test/test_eig_sol_gcg.c:28-169 so that runs are
 * comparable flag for flag (-nevConv -nevMax -blockSize -nevInit + -gcge_*).
 */
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gcge_solver.h"

/* evec_in != NULL: a block of nevMax columns owned by the caller whose first nevGiven columns are start vectors
 * (the `nevGiven` argument of ops->EigenSolver, reference src/ops_eig_sol_gcg.c:101-158) */
static int run_gcg(void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops,
		double *eval_out, void ***evec_out, GCGE_RunResult *res, void **evec_in, int nevGiven)
{
	int nevConv = 30, multiMax = 1, block_size, nevMax, nevInit, i;
	double gapMin = 1e-5, tol_gcg[2] = {1e-1, 1e-8}, *eval, *dbl_ws, t0;
	int max_iter_gcg = 500, *int_ws, sizeV, length_dbl_ws, length_int_ws;
	void **evec, **ws[4];

	ops->GetOptionFromCommandLine("-nevConv", 'i', &nevConv, argc, argv, ops);
	nevMax = 2 * nevConv;
	ops->GetOptionFromCommandLine("-nevMax", 'i', &nevMax, argc, argv, ops);
	block_size = nevConv < 30 ? (nevMax - nevConv) : nevConv / 5;
	ops->GetOptionFromCommandLine("-blockSize", 'i', &block_size, argc, argv, ops);
	nevInit = nevMax;
	ops->GetOptionFromCommandLine("-nevInit", 'i', &nevInit, argc, argv, ops);
	if (nevInit > nevMax) nevInit = nevMax;

	eval = (double*)calloc(nevMax, sizeof(double));
	if (evec_in != NULL) evec = evec_in;
	else {
		ops->MultiVecCreateByMat(&evec, nevMax, A, ops);
		ops->MultiVecSetRandomValue(evec, 0, nevMax, ops);
	}
	ops->MultiVecCreateByMat(&ws[0], nevMax + 2 * block_size, A, ops);
	ops->MultiVecSetRandomValue(ws[0], 0, nevMax + 2 * block_size, ops);
	for (i = 1; i < 4; ++i) {
		ops->MultiVecCreateByMat(&ws[i], block_size, A, ops);
		ops->MultiVecSetRandomValue(ws[i], 0, block_size, ops);
	}
	sizeV = nevInit + 2 * block_size;
	length_dbl_ws = 2 * sizeV * sizeV + 10 * sizeV + (nevMax + 2 * block_size) + nevMax * block_size;
	length_int_ws = 6 * sizeV + 2 * (block_size + 3);
	ops->Printf("length_dbl_ws = %d\n", length_dbl_ws);
	ops->Printf("length_int_ws = %d\n", length_int_ws);
	dbl_ws = (double*)calloc(length_dbl_ws, sizeof(double));
	int_ws = (int*)calloc(length_int_ws, sizeof(int));

	srand(0);   /* the initial block is the glibc rand() stream after srand(0) */
	t0 = ops->GetWtime();
	ops->Printf("===============================================\n");
	ops->Printf("GCG Eigen Solver\n");
	EigenSolverSetup_GCG(multiMax, gapMin, nevInit, nevMax, block_size, tol_gcg, max_iter_gcg,
			flag, ws, dbl_ws, int_ws, ops);
	EigenSolverSetParameters_GCG(50,
			"mgs", 80, 2, 2 * DBL_EPSILON,      /* initial X   */
			"mgs", -1, 2, 2 * DBL_EPSILON,      /* P (host)    */
			"mgs", 80, 2, 2 * DBL_EPSILON,      /* W           */
			30, 1e-2, 1e-14, "abs", 0,          /* block CG    */
			-1, gapMin, 2 * DBL_EPSILON, ops);
	EigenSolverSetParametersFromCommandLine_GCG(argc, argv, ops);
	ops->Printf("nevGiven = %d, nevConv = %d, nevMax = %d, block_size = %d, nevInit = %d\n",
			nevGiven, nevConv, nevMax, block_size, nevInit);
	ops->EigenSolver(A, B, eval, evec, nevGiven, &nevConv, ops);
	if (res != NULL) {
		res->seconds = ops->GetWtime() - t0;
		res->nevConv = nevConv; res->numIter = ((GCGSolver*)ops->eigen_solver_workspace)->numIter;
		res->nevMax = nevMax; res->block_size = block_size; res->nevInit = nevInit;
		res->timing = *GCGE_LastTiming();
	}
	ops->Printf("numIter = %d, nevConv = %d\n", ((GCGSolver*)ops->eigen_solver_workspace)->numIter, nevConv);
	ops->Printf("++++++++++++++++++++++++++++++++++++++++++++++\n");
	ops->Printf("Time is %.3f\n", ops->GetWtime() - t0);

	ops->MultiVecDestroy(&ws[0], nevMax + 2 * block_size, ops);
	for (i = 1; i < 4; ++i) ops->MultiVecDestroy(&ws[i], block_size, ops);
	free(dbl_ws); free(int_ws);
	if (eval_out != NULL) memcpy(eval_out, eval, nevMax * sizeof(double));
	ops->Printf("eigenvalues\n");
	for (i = 0; i < nevConv; ++i) ops->Printf("%d: %6.14e\n", i + 1, eval[i]);
	if (evec_out != NULL) *evec_out = evec;
	else if (evec_in == NULL) ops->MultiVecDestroy(&evec, nevMax, ops);
	free(eval);
	return 0;
}

int GCGE_RunGCG(void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops,
		double *eval_out, void ***evec_out, GCGE_RunResult *res)
{
	return run_gcg(A, B, flag, argc, argv, ops, eval_out, evec_out, res, NULL, 0);
}

int GCGE_RunGCGGiven(void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops,
		double *eval_out, void **evec, int nevGiven, GCGE_RunResult *res)
{
	if (evec == NULL || nevGiven < 0) return -1;
	return run_gcg(A, B, flag, argc, argv, ops, eval_out, NULL, res, evec, nevGiven);
}

int TestEigenSolverGCG(void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops)
{
	return GCGE_RunGCG(A, B, flag, argc, argv, ops, NULL, NULL, NULL);
}
