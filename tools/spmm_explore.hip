// Exploration harness for K1: where do the bytes go?  (not part of the product)
//   A. copy calibration (several shapes)
//   B. stencil sweep  : 1/3/5/7-point matrices on the same grid (cost of each reuse distance)
//   C. column split   : m=64 done as 4x16 / 2x32 column passes
//   D. tile schedules : chunk->XCD maps that stream TI x TJ pencils along k per XCD
// Build: hipcc -O3 --offload-arch=gfx950 -I gcge_amd/csrc/hip tools/spmm_explore.hip gcge_amd/csrc/hip/spmm.hip -o /tmp/spmm_explore
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gcge_hip_internal.h"

extern "C" int gcge_hip_csr_spmm(int nrows, const int*, const int*, const double*, const double*,
                                 long, double*, long, int, void*);
extern "C" void gcge_hip_spmm_tune(int rows_per_wave, int xcd_group, int nt_store);
extern "C" void gcge_hip_spmm_set_chunk_map(const int* d_map, unsigned len);

__global__ void fill_kernel(double* x, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    z ^= z >> 31;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 29;
    x[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}
__global__ void copy16(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}
__global__ void copy16x4(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
  if (i + 3 * blockDim.x < n) {
    double2 v0 = a[i], v1 = a[i + blockDim.x], v2 = a[i + 2 * blockDim.x], v3 = a[i + 3 * blockDim.x];
    __builtin_nontemporal_store(v0.x, &b[i].x); __builtin_nontemporal_store(v0.y, &b[i].y);
    __builtin_nontemporal_store(v1.x, &b[i + blockDim.x].x); __builtin_nontemporal_store(v1.y, &b[i + blockDim.x].y);
    __builtin_nontemporal_store(v2.x, &b[i + 2 * blockDim.x].x); __builtin_nontemporal_store(v2.y, &b[i + 2 * blockDim.x].y);
    __builtin_nontemporal_store(v3.x, &b[i + 3 * blockDim.x].x); __builtin_nontemporal_store(v3.y, &b[i + 3 * blockDim.x].y);
  } else {
    for (int u = 0; u < 4; ++u) if (i + u * blockDim.x < n) b[i + u * blockDim.x] = a[i + u * blockDim.x];
  }
}
__global__ void copy8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}

struct Mat { std::vector<int> rp, ci; std::vector<double> va; int *d_rp, *d_ci; double* d_va; size_t nnz; };

static void build(int N, int npt, Mat& M) {  // npt in {1,3,5,7}
  size_t n = (size_t)N * N * N, p = 0;
  M.rp.resize(n + 1);
  for (int k = 0; k < N; ++k) for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) {
    size_t r = i + (size_t)N * (j + (size_t)N * k);
    M.rp[r] = (int)p;
    if (npt >= 7 && k > 0) M.ci.push_back((int)(r - (size_t)N * N)), M.va.push_back(-1.0), ++p;
    if (npt >= 5 && j > 0) M.ci.push_back((int)(r - N)), M.va.push_back(-1.0), ++p;
    if (npt >= 3 && i > 0) M.ci.push_back((int)(r - 1)), M.va.push_back(-1.0), ++p;
    M.ci.push_back((int)r), M.va.push_back(6.0), ++p;
    if (npt >= 3 && i < N - 1) M.ci.push_back((int)(r + 1)), M.va.push_back(-1.0), ++p;
    if (npt >= 5 && j < N - 1) M.ci.push_back((int)(r + N)), M.va.push_back(-1.0), ++p;
    if (npt >= 7 && k < N - 1) M.ci.push_back((int)(r + (size_t)N * N)), M.va.push_back(-1.0), ++p;
  }
  M.rp[n] = (int)p;
  M.nnz = p;
  GCGE_HIP_CHECK(hipMalloc(&M.d_rp, (n + 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&M.d_ci, p * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&M.d_va, p * sizeof(double)));
  GCGE_HIP_CHECK(hipMemcpy(M.d_rp, M.rp.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(M.d_ci, M.ci.data(), p * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(M.d_va, M.va.data(), p * sizeof(double), hipMemcpyHostToDevice));
}
static void release(Mat& M) { hipFree(M.d_rp); hipFree(M.d_ci); hipFree(M.d_va); M = Mat(); }

static hipEvent_t e0, e1;
template <class F> static float timeit(int reps, F f) {
  f();
  GCGE_HIP_CHECK(hipDeviceSynchronize());
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

// chunk = 16 consecutive rows (rows_per_wave = 4).  Tiles TI x TJ in (i,j), streamed along k,
// tiles dealt round-robin to the 8 XCDs; hardware block 8*q + x takes entry q of XCD x's list.
static std::vector<int> tile_map(int N, int TI, int TJ) {
  const int cpl = N / 16;  // chunks per i-line
  std::vector<std::vector<int>> lists(8);
  int t = 0;
  for (int j0 = 0; j0 < N; j0 += TJ)
    for (int i0 = 0; i0 < N; i0 += TI, ++t) {
      std::vector<int>& L = lists[t % 8];
      for (int k = 0; k < N; ++k)
        for (int j = j0; j < j0 + TJ; ++j)
          for (int ic = i0 / 16; ic < (i0 + TI) / 16; ++ic) L.push_back(ic + cpl * (j + N * k));
    }
  size_t mx = 0;
  for (auto& L : lists) mx = L.size() > mx ? L.size() : mx;
  std::vector<int> map(mx * 8, -1);
  for (int x = 0; x < 8; ++x)
    for (size_t q = 0; q < lists[x].size(); ++q) map[q * 8 + x] = lists[x][q];
  return map;
}

int main(int argc, char** argv) {
  int N = argc > 1 ? atoi(argv[1]) : 256;
  int reps = argc > 2 ? atoi(argv[2]) : 5;
  const int m = 64;
  size_t n = (size_t)N * N * N;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double *d_x, *d_y;
  GCGE_HIP_CHECK(hipMalloc(&d_x, n * m * sizeof(double)));
  GCGE_HIP_CHECK(hipMalloc(&d_y, n * m * sizeof(double)));
  fill_kernel<<<4096, 256>>>(d_x, n * m, 77u);
  GCGE_HIP_CHECK(hipDeviceSynchronize());

  // ---- A. copy calibration
  {
    size_t c2 = n * m / 2;
    double gb = 32.0 * c2 * 1e-9;
    float t;
    t = timeit(reps, [&] { copy16<<<2048, 256>>>((const double2*)d_x, (double2*)d_y, c2); });
    printf("A copy16 grid2048        %7.3f ms %7.1f GB/s\n", t, gb / t * 1e3);
    t = timeit(reps, [&] { copy16<<<(unsigned)((c2 + 255) / 256), 256>>>((const double2*)d_x, (double2*)d_y, c2); });
    printf("A copy16 grid=full       %7.3f ms %7.1f GB/s\n", t, gb / t * 1e3);
    t = timeit(reps, [&] { copy16x4<<<(unsigned)((c2 + 1023) / 1024), 256>>>((const double2*)d_x, (double2*)d_y, c2); });
    printf("A copy16x4 nt            %7.3f ms %7.1f GB/s\n", t, gb / t * 1e3);
    t = timeit(reps, [&] { copy8<<<(unsigned)((n * m + 255) / 256), 256>>>(d_x, d_y, n * m); });
    printf("A copy8  grid=full       %7.3f ms %7.1f GB/s\n", t, gb / t * 1e3);
  }

  // ---- B. stencil sweep (natural order, rpw=4 and 16, nt 0/1)
  for (int npt = 1; npt <= 7; npt += 2) {
    Mat M;
    build(N, npt, M);
    double alg = 12.0 * M.nnz + 4.0 * (n + 1) + 16.0 * (double)n * m;
    for (int rpw : {4, 16})
      for (int nt = 0; nt < 2; ++nt) {
        gcge_hip_spmm_tune(rpw, 1, nt);
        float t = timeit(reps, [&] { gcge_hip_csr_spmm((int)n, M.d_rp, M.d_ci, M.d_va, d_x, m, d_y, m, m, 0); });
        printf("B %dpt rpw=%2d nt=%d        %7.3f ms %7.1f GB/s alg (%.1f%%)\n", npt, rpw, nt, t, alg * 1e-6 / t, alg * 1e-6 / t / 80);
      }
    if (npt == 7) {
      // ---- C. column split
      gcge_hip_spmm_tune(4, 1, 0);
      for (int mp : {8, 16, 32}) {
        float t = timeit(reps, [&] {
          for (int c = 0; c < m; c += mp) gcge_hip_csr_spmm((int)n, M.d_rp, M.d_ci, M.d_va, d_x + c, m, d_y + c, m, mp, 0);
        });
        printf("C 7pt split %2d cols      %7.3f ms %7.1f GB/s alg (%.1f%%)\n", mp, t, alg * 1e-6 / t, alg * 1e-6 / t / 80);
      }
      // ---- D. tile schedules (chunk = 16 rows)
      for (int nt = 0; nt < 2; ++nt) {
        gcge_hip_spmm_tune(4, 1, nt);
        int tis[] = {256, 256, 256, 256, 64, 64, 32, 32, 128};
        int tjs[] = {2, 4, 8, 16, 8, 16, 16, 32, 8};
        for (int q = 0; q < 9; ++q) {
          if (tis[q] > N || tjs[q] > N) continue;
          std::vector<int> map = tile_map(N, tis[q], tjs[q]);
          int* d_map;
          GCGE_HIP_CHECK(hipMalloc(&d_map, map.size() * sizeof(int)));
          GCGE_HIP_CHECK(hipMemcpy(d_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
          gcge_hip_spmm_set_chunk_map(d_map, (unsigned)map.size());
          float t = timeit(reps, [&] { gcge_hip_csr_spmm((int)n, M.d_rp, M.d_ci, M.d_va, d_x, m, d_y, m, m, 0); });
          printf("D 7pt tile %3dx%-3d nt=%d   %7.3f ms %7.1f GB/s alg (%.1f%%)\n", tis[q], tjs[q], nt, t, alg * 1e-6 / t, alg * 1e-6 / t / 80);
          gcge_hip_spmm_set_chunk_map(nullptr, 0);
          hipFree(d_map);
        }
      }
    }
    release(M);
  }
  return 0;
}
