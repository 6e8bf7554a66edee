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

/* ---- BlockAMG as the solver of the W systems: what the reference's SiO2 driver does under OPS_USE_AMG
 * (test/test_eig_sol_SiO2_MAT.c:96-128,160-170): hierarchy from ops->MultiGridCreate, per-level work blocks, parameter arrays
 * {cycles, pre_0, post_0, pre_1, post_1, ...}, MultiLinearSolverSetup_BlockAMG, then GCG with a user-defined linear solver. */
GCGE_AMG *GCGE_AMGCreate(void *A, void *B, int max_levels, int block_size, int cycles, int smooth0, int smooth, double rate0,
		struct OPS_ *ops)
{
	GCGE_AMG *amg; int l, i, L = max_levels;
	if (ops->MultiGridCreate == NULL || max_levels < 2 || block_size < 1) return NULL;
	amg = (GCGE_AMG*)calloc(1, sizeof(GCGE_AMG));
	ops->MultiGridCreate(&amg->A_array, &amg->B_array, &amg->P_array, &L, A, B, ops);
	amg->num_levels = L; amg->block_size = block_size;
	amg->max_iter = (int*)calloc(2 * (size_t)L + 1, sizeof(int));
	amg->rate = (double*)calloc(L, sizeof(double)); amg->tol = (double*)calloc(L, sizeof(double));
	amg->max_iter[0] = cycles > 0 ? cycles : 1;
	for (l = 0; l < L; ++l) {
		amg->max_iter[2 * l + 1] = amg->max_iter[2 * l + 2] = l == 0 ? (smooth0 > 0 ? smooth0 : 5) : (smooth > 0 ? smooth : 4);
		amg->rate[l] = l == 0 ? (rate0 > 0.0 ? rate0 : 1e-2) : 1e-16;      /* test_eig_sol_SiO2_MAT.c:104-105 */
		amg->tol[l] = l == 0 ? 1e-14 : 1e-16;                                /* (level 0: the harness's compW_cg_tol, test_eig_sol_gcg.c:112; the SiO2 file has 1e-8) */
	}
	if (2 * (L - 1) + 2 > 2 * L) abort();
	amg->dbl_ws = (double*)calloc(6 * (size_t)block_size, sizeof(double));
	amg->int_ws = (int*)calloc(2 * (size_t)block_size, sizeof(int));
	/* blocks: right-hand side and solution of every coarse level, the residual block of every level; the CG's p and w blocks
	 * only where the smoother is the solver stack's own BlockPCG (a back-end's smoother brings its blocks) */
	amg->own_smoother = GCGE_HasBlockAMGSmoother(ops);
	for (i = 0; i < 5; ++i) amg->mv_ws[i] = (void***)calloc(L, sizeof(void**));
	for (l = 0; l < L; ++l) {
		for (i = 0; i < 5; ++i) {
			if (i < 2 && l == 0) continue;
			if (i > 2 && amg->own_smoother) { amg->mv_ws[i][l] = amg->mv_ws[2][l]; continue; }
			ops->MultiVecCreateByMat(&amg->mv_ws[i][l], block_size, amg->A_array[l], ops);
		}
	}
	return amg;
}
void GCGE_AMGInstall(GCGE_AMG *amg, struct OPS_ *ops)
{
	MultiLinearSolverSetup_BlockAMG(amg->max_iter, amg->rate, amg->tol, "abs", amg->A_array, amg->P_array, amg->num_levels,
			amg->mv_ws, amg->dbl_ws, amg->int_ws, NULL, ops);
}
void GCGE_AMGDestroy(GCGE_AMG **pamg, struct OPS_ *ops)
{
	GCGE_AMG *amg = *pamg; int l, i;
	if (amg == NULL) return;
	for (l = 0; l < amg->num_levels; ++l)
		for (i = 0; i < 5; ++i) {
			if (amg->mv_ws[i][l] == NULL || (i > 2 && amg->own_smoother)) continue;
			ops->MultiVecDestroy(&amg->mv_ws[i][l], amg->block_size, ops);
		}
	for (i = 0; i < 5; ++i) free(amg->mv_ws[i]);
	l = amg->num_levels;
	ops->MultiGridDestroy(&amg->A_array, &amg->B_array, &amg->P_array, &l, ops);
	free(amg->max_iter); free(amg->rate); free(amg->tol); free(amg->dbl_ws); free(amg->int_ws);
	free(amg); *pamg = NULL;
}

/* evec_in != NULL: a block of nevMax columns owned by the caller whose first nevGiven columns are start vectors
 * (the `nevGiven` argument of ops->EigenSolver, reference src/ops_eig_sol_gcg.c:101-158) */
static int run_gcg(void *A, void *B, int flag, int argc, char *argv[], struct OPS_ *ops,
		double *eval_out, void ***evec_out, GCGE_RunResult *res, void **evec_in, int nevGiven)
{
	int nevConv = 30, multiMax = 1, block_size, nevMax, nevInit, i;
	double gapMin = 1e-5, tol_gcg[2] = {1e-1, 1e-8}, *eval, *dbl_ws, t0;
	int max_iter_gcg = 500, *int_ws, sizeV, length_dbl_ws, length_int_ws;
	int amg_levels = 0, amg_cycles = 1, amg_smooth0 = 5, amg_smooth = 4; double amg_rate = 1e-2;
	GCGE_AMG *amg = NULL;
	void **evec, **ws[4];

	ops->GetOptionFromCommandLine("-nevConv", 'i', &nevConv, argc, argv, ops);
	nevMax = 2 * nevConv;
	ops->GetOptionFromCommandLine("-nevMax", 'i', &nevMax, argc, argv, ops);
	block_size = nevConv < 30 ? (nevMax - nevConv) : nevConv / 5;
	ops->GetOptionFromCommandLine("-blockSize", 'i', &block_size, argc, argv, ops);
	nevInit = nevMax;
	ops->GetOptionFromCommandLine("-nevInit", 'i', &nevInit, argc, argv, ops);
	if (nevInit > nevMax) nevInit = nevMax;

	/* -gcge_amg_levels L (>= 2): the W systems are solved by BlockAMG over the back-end's hierarchy instead of BlockPCG —
	 * OPS_USE_AMG of test/test_eig_sol_SiO2_MAT.c:27,96-128,160-170; a hierarchy the caller installed beforehand
	 * (GCGE_AMGCreate + GCGE_AMGInstall, flag 1) is used as it is */
	ops->GetOptionFromCommandLine("-gcge_amg_levels", 'i', &amg_levels, argc, argv, ops);
	ops->GetOptionFromCommandLine("-gcge_amg_cycles", 'i', &amg_cycles, argc, argv, ops);
	ops->GetOptionFromCommandLine("-gcge_amg_smooth0", 'i', &amg_smooth0, argc, argv, ops);
	ops->GetOptionFromCommandLine("-gcge_amg_smooth", 'i', &amg_smooth, argc, argv, ops);
	ops->GetOptionFromCommandLine("-gcge_amg_rate", 'f', &amg_rate, argc, argv, ops);
	if (amg_levels >= 2) {
		amg = GCGE_AMGCreate(A, B, amg_levels, block_size, amg_cycles, amg_smooth0, amg_smooth, amg_rate, ops);
		if (amg == NULL) { ops->Printf("-gcge_amg_levels: the back-end has no MultiGridCreate\n"); return -7; }
		GCGE_AMGInstall(amg, ops);
		flag = 1;
	}

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
	if (amg != NULL) GCGE_AMGDestroy(&amg, ops);
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
