// The multigrid hierarchy of the HIP back-end: ops->MultiGridCreate / MultiGridDestroy (reference slots src/ops.h:134-139).
//
// The reference's BlockAMG (src/ops_lin_sol.c:466-715; ours: csrc/host/lin_sol.c) takes A_0 = A, A_{l+1}, P_l from the back-end:
// app/app_slepc.c:648-728 extracts them from PETSc GAMG, app/app_hypre.c from BoomerAMG, app/app_lapack.c:863-929 builds a 1-D toy.
// Here: the CSR arrays of A come back from the device (every upload keeps them), the aggregation hierarchy of
// include/gcge_multigrid.h (2 x 2 x 2 cells of a detected grid, greedy aggregates otherwise; A_{l+1} = scale P^T A_l P) is built on
// the host, and every level goes up again through gcge_hip_mat_create — so a coarse Laplacian gets the pattern kernels, a coarse
// real-space Hamiltonian its blocks, exactly like a matrix the caller uploads.  The prolongations are rectangular matrices
// (GCGE_HIP_MAT_::rect_ncols): CSR of P for MatDotMultiVec, CSR of P^T for MatTransDotMultiVec, both through the generic CSR
// kernel (spmm.hip) — one non-zero per fine row, every fine row of the block read or written exactly once.
// Row slabs (one rank per GPU): a slab of whole planes coarsens by itself (every rank pairs its own planes) — local prolongations, coarse slabs
// through the slab constructor (gcge_hip_mat_create_slab over RCCL, or a registered factory: the tests' torch.distributed transport).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "gcge_hip.h"
#include "gcge_solver.h"
#include "gcge_multigrid.h"
#include "gcge_hip_internal.h"

static double mg_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_rect(int nrows, int ncols, const int* rowptr, const int* colidx, const double* val,
                                                  const int* t_rowptr, const int* t_colidx, const double* t_val) {
  if (gcge_hip_init(-1) != 0) return nullptr;
  GCGE_HIP_MAT* P = (GCGE_HIP_MAT*)calloc(1, sizeof(GCGE_HIP_MAT));
  const size_t nnz = (size_t)rowptr[nrows];
  GCGE_REQUIRE(nrows > 0 && ncols > 0 && (size_t)t_rowptr[ncols] == nnz, "gcge_hip_mat_create_rect: the two triples hold the same entries");
  for (size_t k = 0; k < nnz; ++k) GCGE_REQUIRE(colidx[k] >= 0 && colidx[k] < ncols && t_colidx[k] >= 0 && t_colidx[k] < nrows, "gcge_hip_mat_create_rect: index in range");
  P->nrows = nrows; P->nglobal = nrows; P->nnz = (long)nnz; P->rect_ncols = ncols;
  P->rect_one_per_row = nnz == (size_t)nrows;
  for (int r = 0; r < nrows && P->rect_one_per_row; ++r) P->rect_one_per_row = rowptr[r + 1] - rowptr[r] == 1;
  auto up = [](const void* h, size_t bytes) { void* d = nullptr; GCGE_HIP_CHECK(hipMalloc(&d, bytes ? bytes : 8)); GCGE_HIP_CHECK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return d; };
  P->d_rowptr = (int*)up(rowptr, ((size_t)nrows + 1) * sizeof(int));
  P->d_colidx = (int*)up(colidx, nnz * sizeof(int));
  P->d_val = (double*)up(val, nnz * sizeof(double));
  P->d_t_rowptr = (int*)up(t_rowptr, ((size_t)ncols + 1) * sizeof(int));
  P->d_t_colidx = (int*)up(t_colidx, nnz * sizeof(int));
  P->d_t_val = (double*)up(t_val, nnz * sizeof(double));
  return P;
}

// the same from the CSR triple of P alone (the transpose by counting, ascending columns)
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_rect_csr(const GCGE_CSR* P) {
  const int nr = P->nrows, nc = P->ncols;
  const size_t nnz = (size_t)P->rowptr[nr];
  std::vector<int> tp((size_t)nc + 1, 0), tc(nnz ? nnz : 1); std::vector<double> tv(nnz ? nnz : 1);
  for (size_t k = 0; k < nnz; ++k) { GCGE_REQUIRE(P->colidx[k] >= 0 && P->colidx[k] < nc, "gcge_hip_mat_create_rect_csr: column in range"); ++tp[(size_t)P->colidx[k] + 1]; }
  for (int c = 0; c < nc; ++c) tp[c + 1] += tp[c];
  std::vector<int> fill(tp.begin(), tp.end() - 1);
  for (int r = 0; r < nr; ++r)
    for (int k = P->rowptr[r]; k < P->rowptr[r + 1]; ++k) { const int q = fill[P->colidx[k]]++; tc[q] = r; tv[q] = P->val[k]; }
  return gcge_hip_mat_create_rect(nr, nc, P->rowptr, P->colidx, P->val, tp.data(), tc.data(), tv.data());
}

// one hierarchy per MultiGridCreate call; found again through the A_array pointer MultiGridDestroy hands back
struct MgHold { void** A_array; std::vector<GCGE_HIP_MAT*> owned; };
static std::vector<MgHold> g_mg;
static double g_mg_seconds = 0.0;
extern "C" double gcge_hip_multigrid_seconds(void) { return g_mg_seconds; }   // host + upload time of the last MultiGridCreate

static void download_csr(const GCGE_HIP_MAT_* A, GCGE_CSR* out, std::vector<int>& rp, std::vector<int>& ci, std::vector<double>& va) {
  rp.resize((size_t)A->nrows + 1); ci.resize((size_t)(A->nnz ? A->nnz : 1)); va.resize((size_t)(A->nnz ? A->nnz : 1));
  GCGE_HIP_CHECK(hipStreamSynchronize((hipStream_t)gcge_hip_stream()));
  GCGE_HIP_CHECK(hipMemcpy(rp.data(), A->d_rowptr, ((size_t)A->nrows + 1) * sizeof(int), hipMemcpyDeviceToHost));
  GCGE_HIP_CHECK(hipMemcpy(ci.data(), A->d_colidx, (size_t)A->nnz * sizeof(int), hipMemcpyDeviceToHost));
  GCGE_HIP_CHECK(hipMemcpy(va.data(), A->d_val, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToHost));
  memset(out, 0, sizeof *out);
  out->nrows = A->nrows; out->ncols = A->nrows; out->row_begin = 0; out->nnz = A->nnz;
  out->rowptr = rp.data(); out->colidx = ci.data(); out->val = va.data();
}

static gcge_hip_slab_factory_fn g_slab_factory = nullptr; static void* g_slab_factory_ctx = nullptr;
extern "C" void gcge_hip_set_slab_factory(gcge_hip_slab_factory_fn fn, void* ctx) { g_slab_factory = fn; g_slab_factory_ctx = ctx; }

// a row slab (one rank per GPU): whole planes of a detected grid, cut on any plane boundary — every rank coarsens its own slab
// (gcge_mg_build_slab), the coarse slabs go through the slab constructor (ghost list, halo plan: collective)
static void multigrid_create_slab(void*** A_array, void*** B_array, void*** P_array, int* num_levels, const GCGE_HIP_MAT_* mA, void* A, void* B) {
  GCGE_REQUIRE(mA->h_part != nullptr && mA->part_world >= 1, "MultiGridCreate on a row slab: the partition of all ranks (gcge_hip_mat_create_slab / gcge_hip_mat_set_partition)");
  GCGE_REQUIRE(mA->nghost == 0 || mA->h_ghost_global != nullptr, "MultiGridCreate on a row slab: the global rows behind the halo columns (gcge_hip_mat_create_local_ghosts)");
  const int world = mA->part_world;
  int rank = -1;
  for (int r = 0; r < world; ++r) if (mA->h_part[r] == mA->row_begin && mA->h_part[r + 1] == (long)mA->row_begin + mA->nrows) rank = r;
  GCGE_REQUIRE(rank >= 0, "MultiGridCreate on a row slab: the slab is one of the partition's");
  GCGE_CSR cA; std::vector<int> rp, ci; std::vector<double> va;
  download_csr(mA, &cA, rp, ci, va);
  for (size_t k = 0; k < (size_t)mA->nnz; ++k) ci[k] = ci[k] < mA->nrows ? mA->row_begin + ci[k] : mA->h_ghost_global[ci[k] - mA->nrows];   // local -> GLOBAL columns
  cA.row_begin = mA->row_begin; cA.ncols = (int)mA->h_part[world];
  int dims[3] = {0, 0, 0};
  if (!gcge_mg_detect_grid(&cA, dims, nullptr) || (long)dims[0] * dims[1] * dims[2] != mA->h_part[world]) {
    fprintf(stderr, "MultiGridCreate (HIP back-end): a row slab is coarsened along a detected grid only; this matrix shows none\n");
    abort();
  }
  GCGE_MG mg; long* parts = nullptr;
  const int rc = gcge_mg_build_slab(&cA, dims, mA->h_part, rank, world, *num_levels, 0.0, &mg, &parts);
  if (rc != 0) { fprintf(stderr, "MultiGridCreate (HIP back-end): gcge_mg_build_slab failed (%d) — slabs must be whole planes of the %d x %d x %d grid\n", rc, dims[0], dims[1], dims[2]); abort(); }
  const int L = mg.num_levels;
  MgHold h;
  *A_array = (void**)calloc(L, sizeof(void*));
  *P_array = (void**)calloc(L > 1 ? L - 1 : 1, sizeof(void*));
  if (B_array != nullptr) *B_array = (void**)calloc(L, sizeof(void*));
  (*A_array)[0] = A;
  if (B_array != nullptr) (*B_array)[0] = B;
  const int buf_cols = mA->buf_cols > 0 ? mA->buf_cols : 128;
  for (int l = 1; l < L; ++l) {
    const long* pl = parts + (size_t)l * (world + 1);
    GCGE_HIP_MAT* a = g_slab_factory != nullptr
        ? g_slab_factory(pl, world, rank, mg.A[l].rowptr, mg.A[l].colidx, mg.A[l].val, buf_cols, g_slab_factory_ctx)
        : gcge_hip_mat_create_slab(pl, mg.A[l].rowptr, mg.A[l].colidx, mg.A[l].val, buf_cols);
    GCGE_REQUIRE(a != nullptr, "MultiGridCreate: construction of a coarse slab");
    if (a->h_part == nullptr) gcge_hip_mat_set_partition(a, pl, world);
    (*A_array)[l] = a; h.owned.push_back(a);
  }
  for (int l = 0; l + 1 < L; ++l) {
    GCGE_HIP_MAT* p = gcge_hip_mat_create_rect(mg.P[l].nrows, mg.P[l].ncols, mg.P[l].rowptr, mg.P[l].colidx, mg.P[l].val,
                                               mg.PT[l].rowptr, mg.PT[l].colidx, mg.PT[l].val);
    GCGE_REQUIRE(p != nullptr, "MultiGridCreate: upload of a prolongation");
    (*P_array)[l] = p; h.owned.push_back(p);
  }
  if (getenv("GCGE_MG_TRACE") != nullptr)
    for (int l = 0; l < L; ++l)
      fprintf(stderr, "MultiGridCreate: rank %d of %d, level %d: %d of %ld rows, %ld non-zeros, grid %d x %d x %d, K1 form %s\n", rank, world, l, mg.A[l].nrows,
              parts[(size_t)l * (world + 1) + world], (long)mg.A[l].nnz, mg.dims[l][0], mg.dims[l][1], mg.dims[l][2], gcge_hip_mat_spmm_form((const GCGE_HIP_MAT*)(*A_array)[l]));
  gcge_mg_free(&mg);
  free(parts);
  h.A_array = *A_array;
  g_mg.push_back(h);
  *num_levels = L;
}

extern "C" void gcge_hip_multigrid_create(void*** A_array, void*** B_array, void*** P_array, int* num_levels, void* A, void* B, struct OPS_* ops) {
  const GCGE_HIP_MAT_* mA = (const GCGE_HIP_MAT_*)A; const GCGE_HIP_MAT_* mB = (const GCGE_HIP_MAT_*)B;
  const double t0 = mg_now();
  GCGE_REQUIRE(mA != nullptr && mA->rect_ncols == 0 && num_levels != nullptr && *num_levels >= 1, "MultiGridCreate: a square matrix and a level count");
  if (mA->nghost > 0 || mA->part_world > 1) {
    multigrid_create_slab(A_array, B_array, P_array, num_levels, mA, A, B);
    g_mg_seconds = mg_now() - t0;
    return;
  }
  GCGE_CSR cA, cB; std::vector<int> rpA, ciA, rpB, ciB; std::vector<double> vaA, vaB;
  download_csr(mA, &cA, rpA, ciA, vaA);
  if (mB != nullptr) download_csr(mB, &cB, rpB, ciB, vaB);
  GCGE_MG mg;
  if (gcge_mg_build(&cA, mB != nullptr ? &cB : nullptr, *num_levels, 0, 0.0, &mg) != 0) { fprintf(stderr, "MultiGridCreate: out of host memory\n"); abort(); }
  const int L = mg.num_levels;
  MgHold h;
  *A_array = (void**)calloc(L, sizeof(void*));
  *P_array = (void**)calloc(L > 1 ? L - 1 : 1, sizeof(void*));
  if (B_array != nullptr) *B_array = (void**)calloc(L, sizeof(void*));
  (*A_array)[0] = A;
  if (B_array != nullptr) (*B_array)[0] = B;
  for (int l = 1; l < L; ++l) {
    GCGE_HIP_MAT* a = gcge_hip_mat_create(mg.A[l].nrows, mg.A[l].nrows, 0, mg.A[l].rowptr, mg.A[l].colidx, mg.A[l].val);
    GCGE_REQUIRE(a != nullptr, "MultiGridCreate: upload of a coarse matrix");
    (*A_array)[l] = a; h.owned.push_back(a);
    if (B_array != nullptr && mB != nullptr) {
      GCGE_HIP_MAT* b = gcge_hip_mat_create(mg.B[l].nrows, mg.B[l].nrows, 0, mg.B[l].rowptr, mg.B[l].colidx, mg.B[l].val);
      GCGE_REQUIRE(b != nullptr, "MultiGridCreate: upload of a coarse B");
      (*B_array)[l] = b; h.owned.push_back(b);
    }
  }
  for (int l = 0; l + 1 < L; ++l) {
    GCGE_HIP_MAT* p = gcge_hip_mat_create_rect(mg.P[l].nrows, mg.P[l].ncols, mg.P[l].rowptr, mg.P[l].colidx, mg.P[l].val,
                                               mg.PT[l].rowptr, mg.PT[l].colidx, mg.PT[l].val);
    GCGE_REQUIRE(p != nullptr, "MultiGridCreate: upload of a prolongation");
    (*P_array)[l] = p; h.owned.push_back(p);
  }
  if (getenv("GCGE_MG_TRACE") != nullptr) {
    for (int l = 0; l < L; ++l)
      fprintf(stderr, "MultiGridCreate: level %d: %d rows, %ld non-zeros, grid %d x %d x %d, K1 form %s\n", l, mg.A[l].nrows, (long)mg.A[l].nnz,
              mg.dims[l][0], mg.dims[l][1], mg.dims[l][2], gcge_hip_mat_spmm_form((const GCGE_HIP_MAT*)(*A_array)[l]));
  }
  gcge_mg_free(&mg);
  h.A_array = *A_array;
  g_mg.push_back(h);
  *num_levels = L;
  g_mg_seconds = mg_now() - t0;
}

extern "C" void gcge_hip_multigrid_destroy(void*** A_array, void*** B_array, void*** P_array, int* num_levels, struct OPS_* ops) {
  GCGE_HIP_CHECK(hipStreamSynchronize((hipStream_t)gcge_hip_stream()));
  for (size_t i = 0; i < g_mg.size(); ++i) {
    if (g_mg[i].A_array != *A_array) continue;
    for (GCGE_HIP_MAT* m : g_mg[i].owned) gcge_hip_mat_destroy(m);
    g_mg.erase(g_mg.begin() + (long)i);
    break;
  }
  free(*A_array); *A_array = nullptr;
  free(*P_array); *P_array = nullptr;
  if (B_array != nullptr && *B_array != nullptr) { free(*B_array); *B_array = nullptr; }
}
