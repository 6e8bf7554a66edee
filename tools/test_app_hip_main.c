/* test_app_hip_main.c — the reference's test/main.c:40-49 with the one line a maintainer adds: TestAppHIP(argc, argv).
 * Plain C, no Python: links libgcge_hip.so (TestAppHIP, OPS_HIP_Set, the kernels) and libgcge_host.so (the solver stack and
 * TestEigenSolverGCG under the reference's names) — or, in a GCGE tree, the reference's own objects instead of the latter.
 *   gcc -O2 -Iinclude tools/test_app_hip_main.c -o /tmp/test_app_hip -Lgcge_amd/lib -lgcge_hip -lgcge_host -Wl,-rpath,$PWD/gcge_amd/lib -lm
 *   /tmp/test_app_hip                                   the reference's stock pair (1-D FE, n = 807): 38 iterations
 *   /tmp/test_app_hip -hip_problem lap3d -hip_size 50 -nevConv 20 -hip_flag 1          BASELINE config 1 on the GPU
 *   /tmp/test_app_hip -hip_mtx_A SiO2.mtx -nevConv 100 -nevMax 200 -blockSize 64 -hip_flag 1    a SuiteSparse file       */
#include "gcge_hip.h"

int main(int argc, char *argv[])
{
	return TestAppHIP(argc, argv);
}
