/* test_app_hip_multi.c — a plain-C host driving several MI355X through libgcge_hip.so: the multi-GPU counterpart of
 * the reference's test/test_app_ccs.c (one process per GPU instead of the reference's MPI ranks; the collectives the
 * reference issues from src/ops_multi_vec.c:206-228 and src/ops_lin_sol.c:313-321,361-369 are RCCL calls inside the
 * back-end).  No MPI is needed: the only thing the ranks share up front is rank 0's 128-byte RCCL id, passed here
 * through a file (an MPI program would MPI_Bcast it).
 *
 *   build:  gcc -O2 -Iinclude tools/test_app_hip_multi.c -o /tmp/test_app_hip_multi \
 *               -Lgcge_amd/lib -lgcge_hip -lgcge_host -Wl,-rpath,$PWD/gcge_amd/lib -lm
 *   run (one process per GPU, e.g. from a shell loop or mpirun/srun):
 *           GCGE_RANK=r GCGE_WORLD=8 GCGE_ID_FILE=/tmp/gcge.id ./test_app_hip_multi N nev block nevMax
 *   problem: 7-point Laplacian on an N x N x (N * world) box cut into slabs along z (weak scaling), standard problem.
 *   Every rank prints its timing; rank 0 prints the Ritz values' error against the closed form. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "gcge_hip.h"
#include "gcge_solver.h"

static int env_int(const char *name, int dflt) { const char *s = getenv(name); return s ? atoi(s) : dflt; }

static void share_id(int rank, const char *path, unsigned char id[GCGE_HIP_COMM_ID_BYTES])
{
	char tmp[1024];
	if (rank == 0) {
		if (gcge_hip_comm_unique_id(id) != 0) { fprintf(stderr, "gcge_hip_comm_unique_id failed\n"); exit(2); }
		snprintf(tmp, sizeof tmp, "%s.tmp", path);
		FILE *f = fopen(tmp, "wb");
		if (!f || fwrite(id, 1, GCGE_HIP_COMM_ID_BYTES, f) != GCGE_HIP_COMM_ID_BYTES) { perror(tmp); exit(2); }
		fclose(f);
		rename(tmp, path);                              /* atomic: readers never see a partial file */
		return;
	}
	for (int tries = 0; tries < 6000; ++tries) {        /* up to 60 s */
		FILE *f = fopen(path, "rb");
		if (f) {
			size_t k = fread(id, 1, GCGE_HIP_COMM_ID_BYTES, f);
			fclose(f);
			if (k == GCGE_HIP_COMM_ID_BYTES) return;
		}
		usleep(10000);
	}
	fprintf(stderr, "rank %d: no id file %s\n", rank, path); exit(2);
}

int main(int argc, char **argv)
{
	const int rank = env_int("GCGE_RANK", 0), world = env_int("GCGE_WORLD", 1);
	const int device = env_int("GCGE_LOCAL_DEVICE", rank);
	const char *idfile = getenv("GCGE_ID_FILE") ? getenv("GCGE_ID_FILE") : "/tmp/gcge_rccl.id";
	const int N = argc > 1 ? atoi(argv[1]) : 32, nev = argc > 2 ? atoi(argv[2]) : 10;
	const int block = argc > 3 ? atoi(argv[3]) : 0, nevMax = argc > 4 ? atoi(argv[4]) : 0;

	if (gcge_hip_init(device) != 0) return 1;
	unsigned char id[GCGE_HIP_COMM_ID_BYTES];
	share_id(rank, idfile, id);
	if (gcge_hip_comm_init(rank, world, id) != 0) { fprintf(stderr, "gcge_hip_comm_init failed\n"); return 1; }

	/* row slabs of the Nx x Ny x Nz Laplacian, global column indices */
	const int Nz = N * world;
	const long n = (long)N * N * Nz;
	long *part = malloc((world + 1) * sizeof(long));
	for (int r = 0; r <= world; ++r) part[r] = (long)N * N * ((long)Nz * r / world);
	GCGE_CSR S;
	if (gcge_problem_lap3d_box(N, N, Nz, part[rank], part[rank + 1], &S) != 0) return 1;
	GCGE_HIP_MAT *A = gcge_hip_mat_create_slab(part, S.rowptr, S.colidx, S.val, block > 0 ? block : 64);
	if (A == NULL) return 1;

	OPS *ops = NULL;
	OPS_Create(&ops);
	OPS_HIP_Set(ops);
	OPS_Setup(ops);
	GCGE_SetQuiet(ops, 1);
	gcge_hip_set_random_mode(1, 20240601ULL);
	gcge_hip_bpcg_setup(ops, 30, 1e-2, 1e-14, "abs");               /* fused device CG behind flag 1 */

	char a_nev[16], a_blk[16], a_max[16];
	char *av[16]; int ac = 0;
	av[ac++] = "test_app_hip_multi";
	snprintf(a_nev, sizeof a_nev, "%d", nev); av[ac++] = "-nevConv"; av[ac++] = a_nev;
	if (block > 0) { snprintf(a_blk, sizeof a_blk, "%d", block); av[ac++] = "-blockSize"; av[ac++] = a_blk; }
	if (nevMax > 0) { snprintf(a_max, sizeof a_max, "%d", nevMax); av[ac++] = "-nevMax"; av[ac++] = a_max; }
	av[ac++] = "-gcge_initX_orth_method"; av[ac++] = "chol";
	av[ac++] = "-gcge_compW_orth_method"; av[ac++] = "chol";
	av[ac++] = "-gcge_print_usage"; av[ac++] = "0";
	const int nm = nevMax > 0 ? nevMax : 2 * nev;
	double *eval = calloc(nm, sizeof(double));
	GCGE_RunResult res;
	if (GCGE_RunGCG(A, NULL, 1, ac, av, ops, eval, NULL, &res) != 0) return 1;

	long n_ar = 0, n_ex = 0;
	gcge_hip_comm_stats(&n_ar, &n_ex);
	printf("rank %d/%d: n = %ld (%d rows here), converged %d in %d iterations, %.3f s, %ld all-reduces, %ld halo exchanges\n",
	       rank, world, n, S.nrows, res.nevConv, res.numIter, res.seconds, n_ar, n_ex);
	if (rank == 0) {
		/* closed form: 6 - 2cos(i pi/(N+1)) - 2cos(j pi/(N+1)) - 2cos(k pi/(Nz+1)), the smallest res.nevConv of them */
		const int c = 24;                         /* lowest modes per direction are enough for small nev */
		double *lam = malloc((size_t)c * c * c * sizeof(double)); int cnt = 0;
		for (int i = 1; i <= c && i <= N; ++i) for (int j = 1; j <= c && j <= N; ++j) for (int k = 1; k <= c && k <= Nz; ++k)
			lam[cnt++] = 6.0 - 2.0 * cos(i * M_PI / (N + 1)) - 2.0 * cos(j * M_PI / (N + 1)) - 2.0 * cos(k * M_PI / (Nz + 1));
		for (int a = 1; a < cnt; ++a) { double v = lam[a]; int b = a - 1; while (b >= 0 && lam[b] > v) { lam[b + 1] = lam[b]; --b; } lam[b + 1] = v; }
		double worst = 0.0;
		for (int k = 0; k < res.nevConv && k < cnt; ++k) { double e = fabs(eval[k] - lam[k]) / lam[k]; if (e > worst) worst = e; }
		printf("max relative error of the %d converged Ritz values vs the closed form: %.3e %s\n", res.nevConv, worst,
		       worst < 1e-10 && res.nevConv >= nev ? "OK" : "FAILED");
		free(lam);
		if (!(worst < 1e-10 && res.nevConv >= nev)) return 3;
	}
	gcge_hip_bpcg_release(ops);
	gcge_hip_mat_destroy(A);
	gcge_csr_free(&S);
	OPS_Destroy(&ops);
	gcge_hip_comm_finalize();
	free(part); free(eval);
	return 0;
}
