// Stand-alone micro-benchmark for K1 fast path (pad-8 format) on the 3-D 7-point Laplacian.
// Build:  hipcc -O3 --offload-arch=gfx950 -I gcge_amd/csrc/hip tools/spmm_bench.hip gcge_amd/csrc/hip/spmm.hip -o tools/spmm_bench
// Run  :  tools/spmm_bench N m ldx [reps]
// Prints one line per tuning variant: ms per launch, algorithmic GB/s (SURVEY.md §8d formula).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <array>
#include <algorithm>
#include "gcge_hip_internal.h"

extern "C" int gcge_hip_csr_spmm(int nrows, const int*, const int*, const double*, const double*,
                                 long, double*, long, int, void*);
extern "C" void gcge_hip_spmm_tune(int rows_per_wave, int xcd_group, int nt_store);
extern "C" void gcge_hip_spmm_variant(int variant, int batch);
extern "C" int gcge_hip_pad8_spmm(int nrows, const int*, const int*, const double*, const double*,
                                  long, double*, long, int, void*);
extern "C" void gcge_hip_spmm_pad8_tune(int rows_per_wave, int batch, int store_policy, int col_pass);
extern "C" void gcge_hip_spmm_pad8_gridcap(int cap);
extern "C" void gcge_hip_spmm_pad8_schedule(const int* d_sched, int len, int rows_per_wave, int grid);

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

__global__ void copy_kernel(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}

static void build_lap3d(int N, std::vector<int>& rp, std::vector<int>& ci, std::vector<double>& va) {
  const int npt = getenv("NPT") ? atoi(getenv("NPT")) : 7;
  size_t n = (size_t)N * N * N;
  rp.resize(n + 1);
  ci.reserve(7 * n);
  va.reserve(7 * n);
  size_t p = 0;
  for (int k = 0; k < N; ++k)
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) {
        size_t r = i + (size_t)N * (j + (size_t)N * k);
        rp[r] = (int)p;
        if (npt >= 7 && k > 0) ci.push_back((int)(r - (size_t)N * N)), va.push_back(-1.0), ++p;
        if (npt >= 5 && j > 0) ci.push_back((int)(r - N)), va.push_back(-1.0), ++p;
        if (npt >= 3 && i > 0) ci.push_back((int)(r - 1)), va.push_back(-1.0), ++p;
        ci.push_back((int)r), va.push_back(6.0), ++p;
        if (npt >= 3 && i < N - 1) ci.push_back((int)(r + 1)), va.push_back(-1.0), ++p;
        if (npt >= 5 && j < N - 1) ci.push_back((int)(r + N)), va.push_back(-1.0), ++p;
        if (npt >= 7 && k < N - 1) ci.push_back((int)(r + (size_t)N * N)), va.push_back(-1.0), ++p;
      }
  rp[n] = (int)p;
}


// ---- probe: upper bound of a pattern-compressed matrix stream (no per-entry (col,val) loads at all) ----
// lane -> (row slot g = l>>3, column pair i = l&7); 16-column passes; the 7-point stencil offsets/values are
// kernel arguments.  Boundary rows are treated like interior ones with clamped columns (WRONG numbers at the
// faces — this kernel only measures the memory pipeline).
typedef double v2dp __attribute__((ext_vector_type(2)));
struct StencilPat { long off[8]; double val[8]; };
__global__ __launch_bounds__(256) void pattern_probe(long nrows, const double* __restrict__ x, size_t ldx,
                                                     double* __restrict__ y, size_t ldy, StencilPat pat, int spw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 3, i = lane & 7;
  const double* xl = x + 2 * i;
  const long slice0 = ((long)blockIdx.x * 4 + wave) * spw;
  for (int sl = 0; sl < spw; ++sl) {
    const long row = (slice0 + sl) * 8 + g;
    if ((slice0 + sl) * 8 >= nrows) break;
    const long rc = min(row, nrows - 1);
    v2dp xv[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      long c = rc + pat.off[t]; c = c < 0 ? 0 : (c >= nrows ? nrows - 1 : c);
      xv[t] = *reinterpret_cast<const v2dp*>(xl + (size_t)c * ldx);
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int t = 0; t < 7; ++t) { a0 = fma(pat.val[t], xv[t].x, a0); a1 = fma(pat.val[t], xv[t].y, a1); }
    if (row < nrows) { v2dp o = {a0, a1}; __builtin_nontemporal_store(o, reinterpret_cast<v2dp*>(y + (size_t)row * ldy + 2 * i)); }
  }
}

int main(int argc, char** argv) {
  int N = argc > 1 ? atoi(argv[1]) : 128;
  int m = argc > 2 ? atoi(argv[2]) : 64;
  long ldx = argc > 3 ? atol(argv[3]) : m;
  int reps = argc > 4 ? atoi(argv[4]) : 10;
  size_t n = (size_t)N * N * N;
  std::vector<int> rp, ci;
  std::vector<double> va;
  build_lap3d(N, rp, ci, va);
  size_t nnz = ci.size();
  printf("Lap3D N=%d n=%zu nnz=%zu m=%d ldx=%ld\n", N, n, nnz, m, ldx);
  int *d_rp, *d_ci;
  double *d_va, *d_x, *d_y;
  GCGE_HIP_CHECK(hipMalloc(&d_rp, (n + 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&d_ci, nnz * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&d_va, nnz * sizeof(double)));
  GCGE_HIP_CHECK(hipMalloc(&d_x, n * ldx * sizeof(double)));
  GCGE_HIP_CHECK(hipMalloc(&d_y, n * (size_t)m * sizeof(double)));
  GCGE_HIP_CHECK(hipMemcpy(d_rp, rp.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(d_ci, ci.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(d_va, va.data(), nnz * sizeof(double), hipMemcpyHostToDevice));
  fill_kernel<<<4096, 256>>>(d_x, n * ldx, 1234u);
  GCGE_HIP_CHECK(hipDeviceSynchronize());
  const int x0 = (ldx >= 2 * m) ? (int)(ldx - m) : 0;  // use the LAST m columns of a wide block
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double alg_bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * (double)n * m;

  // calibration: float4-style copy of the same volume as X+Y
  {
    size_t cnt = n * (size_t)m / 2;
    copy_kernel<<<2048, 256>>>((const double2*)d_x, (double2*)d_y, cnt);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) copy_kernel<<<2048, 256>>>((const double2*)d_x, (double2*)d_y, cnt);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("copy   %8.3f ms  %8.1f GB/s (read+write %0.2f GB)\n", ms, 16.0 * cnt * 1e-6 / ms * 1e-3 * 1e3,
           16.0 * cnt * 1e-9);
  }

  // build the pad-8 format on the host
  std::vector<int> orp(n + 1);
  std::vector<int> pc;
  std::vector<double> pv;
  pc.reserve(8 * n);
  pv.reserve(8 * n);
  for (size_t r = 0; r < n; ++r) {
    orp[r] = (int)(pc.size() / 8);
    for (int k = rp[r]; k < rp[r + 1]; ++k) pc.push_back(ci[k]), pv.push_back(va[k]);
    while (pc.size() % 8) pc.push_back((int)r), pv.push_back(0.0);
  }
  orp[n] = (int)(pc.size() / 8);
  int *d_orp, *d_pc;
  double* d_pv;
  GCGE_HIP_CHECK(hipMalloc(&d_orp, (n + 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&d_pc, pc.size() * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&d_pv, pv.size() * sizeof(double)));
  GCGE_HIP_CHECK(hipMemcpy(d_orp, orp.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(d_pc, pc.data(), pc.size() * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(d_pv, pv.data(), pv.size() * sizeof(double), hipMemcpyHostToDevice));
  if (argc > 8) {
    int rpw = atoi(argv[5]), batch = atoi(argv[6]), store = atoi(argv[7]), pass = atoi(argv[8]);
    gcge_hip_spmm_pad8_tune(rpw, batch, store, pass);
    for (int r = 0; r < reps; ++r)
      gcge_hip_pad8_spmm((int)n, d_orp, d_pc, d_pv, d_x + x0, ldx, d_y, m, m, 0);
    GCGE_HIP_CHECK(hipDeviceSynchronize());
    printf("single config done\n");
    return 0;
  }
  if (getenv("BRICKS")) {
    // per-XCD brick schedules: brick = bx x by x bz grid points, chunk = 4*rpw consecutive rows along i
    struct Cfg { int bx, by, bz, rpw, grid, store; };
    std::vector<Cfg> cfgs;
    for (int store : {1, 2})
      for (int rpw : {4, 8})
        for (int grid : {1024, 2048, 4096})
          for (auto b : {std::array<int,3>{16,16,16}, {32,16,8}, {32,8,8}, {16,16,8}, {64,8,8}, {32,16,16}, {256,4,4}, {256,8,2}, {256,2,8}})
            cfgs.push_back({b[0], b[1], b[2], rpw, grid, store});
    int one[6] = {0,0,0,0,0,0}; bool only = false;
    if (getenv("BRICK_ONE")) { only = sscanf(getenv("BRICK_ONE"), "%d,%d,%d,%d,%d,%d", &one[0], &one[1], &one[2], &one[3], &one[4], &one[5]) == 6; }
    for (auto& c : cfgs) {
      if (only && !(c.bx == one[0] && c.by == one[1] && c.bz == one[2] && c.rpw == one[3] && c.grid == one[4] && c.store == one[5])) continue;
      const int rpb = 4 * c.rpw;
      if (c.bx % rpb || N % c.bx || N % c.by || N % c.bz) continue;
      std::vector<std::vector<int>> lists(8);
      int t = 0;
      for (int k0 = 0; k0 < N; k0 += c.bz) for (int j0 = 0; j0 < N; j0 += c.by) for (int i0 = 0; i0 < N; i0 += c.bx, ++t) {
        auto& L = lists[t % 8];
        for (int k = k0; k < k0 + c.bz; ++k) for (int j = j0; j < j0 + c.by; ++j) for (int ic = i0 / rpb; ic < (i0 + c.bx) / rpb; ++ic)
          L.push_back(ic + (N / rpb) * (j + N * k));
      }
      size_t len = 0; for (auto& L : lists) len = std::max(len, L.size());
      std::vector<int> sched(8 * len, -1);
      for (int x = 0; x < 8; ++x) for (size_t q = 0; q < lists[x].size(); ++q) sched[x * len + q] = lists[x][q];
      int* d_s; GCGE_HIP_CHECK(hipMalloc(&d_s, sched.size() * sizeof(int)));
      GCGE_HIP_CHECK(hipMemcpy(d_s, sched.data(), sched.size() * sizeof(int), hipMemcpyHostToDevice));
      gcge_hip_spmm_pad8_tune(c.rpw, 4, c.store, 0);
      gcge_hip_spmm_pad8_schedule(d_s, (int)len, c.rpw, c.grid);
      gcge_hip_pad8_spmm((int)n, d_orp, d_pc, d_pv, d_x + x0, ldx, d_y, m, m, 0);
      GCGE_HIP_CHECK(hipDeviceSynchronize());
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) gcge_hip_pad8_spmm((int)n, d_orp, d_pc, d_pv, d_x + x0, ldx, d_y, m, m, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("brick %3dx%2dx%2d rpw=%d grid=%4d store=%d  %8.3f ms  %8.1f GB/s alg (%.1f%%)\n", c.bx, c.by, c.bz, c.rpw, c.grid,
             c.store, ms, alg_bytes * 1e-6 / ms, alg_bytes * 1e-6 / ms / 80.0);
      gcge_hip_spmm_pad8_schedule(nullptr, 0, 0, 0);
      hipFree(d_s);
    }
    // grid-stride without schedule for comparison
    for (int cap : {0, 2048, 4096, 8192, 16384}) {
      if (only && cap != 0) continue;
      gcge_hip_spmm_pad8_tune(4, 4, 1, 0); gcge_hip_spmm_pad8_gridcap(cap);
      gcge_hip_pad8_spmm((int)n, d_orp, d_pc, d_pv, d_x + x0, ldx, d_y, m, m, 0);
      GCGE_HIP_CHECK(hipDeviceSynchronize());
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) gcge_hip_pad8_spmm((int)n, d_orp, d_pc, d_pv, d_x + x0, ldx, d_y, m, m, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("gridcap %5d  %8.3f ms (%.1f%%)\n", cap, ms, alg_bytes * 1e-6 / ms / 80.0);
    }
    gcge_hip_spmm_pad8_gridcap(0);
    if (only) return 0;
  }
  double best = 1e30;
  int rpw_list[] = {2, 4, 8};
  int batch_list[] = {4, 8, 16};
  int pass_list[] = {0, 32, 16};
  const bool quick = getenv("QUICK") != nullptr;
  for (int pi = 0; pi < 3; ++pi)
    for (int store = 0; store < 3; ++store)
      for (int ri = 0; ri < 3; ++ri)
        for (int bi = 0; bi < 3; ++bi) {
          if (pass_list[pi] >= m) continue;
          if (quick && (pi > 0 || store != 1 || batch_list[bi] == 16)) continue;
          gcge_hip_spmm_pad8_tune(rpw_list[ri], batch_list[bi], store, pass_list[pi]);
          int rc = gcge_hip_pad8_spmm((int)n, d_orp, d_pc, d_pv, d_x + x0, ldx, d_y, m, m, 0);
          if (rc) { printf("pad8 rc=%d\n", rc); return 2; }
          GCGE_HIP_CHECK(hipDeviceSynchronize());
          hipEventRecord(e0);
          for (int r = 0; r < reps; ++r)
            gcge_hip_pad8_spmm((int)n, d_orp, d_pc, d_pv, d_x + x0, ldx, d_y, m, m, 0);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          ms /= reps;
          if (ms < best) best = ms;
          printf("pad8 pass=%3d store=%d rpw=%2d batch=%2d  %8.3f ms  %8.1f GB/s alg  (%.1f%% of 8 TB/s)\n",
                 pass_list[pi], store, rpw_list[ri], batch_list[bi], ms, alg_bytes * 1e-6 / ms,
                 alg_bytes * 1e-6 / ms / 80.0);
        }
  printf("best %.3f ms  alg bytes %.3f GB\n", best, alg_bytes * 1e-9);

  if (getenv("PATTERN")) {
    StencilPat pat; long offs[7] = {-(long)N * N, -(long)N, -1, 0, 1, (long)N, (long)N * N};
    for (int t = 0; t < 7; ++t) { pat.off[t] = offs[t]; pat.val[t] = t == 3 ? 6.0 : -1.0; } pat.off[7] = 0; pat.val[7] = 0;
    for (int spw : {1, 2, 4}) {
      auto run = [&]() {
        for (int c0 = 0; c0 < m; c0 += 16) {
          long nb = ((long)n + 32L * spw - 1) / (32L * spw);
          pattern_probe<<<(unsigned)nb, 256>>>((long)n, d_x + x0 + c0, (size_t)ldx, d_y + c0, (size_t)m, pat, spw);
        }
      };
      run(); GCGE_HIP_CHECK(hipDeviceSynchronize());
      hipEventRecord(e0); for (int r = 0; r < reps; ++r) run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("pattern probe spw=%d  %8.3f ms  (%.1f%% of 8 TB/s on the CSR-equivalent bytes)\n", spw, ms, alg_bytes * 1e-6 / ms / 80.0);
    }
  }
  // verification on sampled rows against a host recomputation
  std::vector<double> hy(64 * (size_t)m), hx;
  double maxerr = 0.0;
  size_t sample_rows[] = {0, 1, (size_t)N, n / 2 + 17, n - 1, n / 3, (size_t)N * N + 5};
  for (size_t r : sample_rows) {
    std::vector<double> yr(m), acc(m, 0.0), xr(m);
    GCGE_HIP_CHECK(hipMemcpy(yr.data(), d_y + r * (size_t)m, m * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = rp[r]; k < rp[r + 1]; ++k) {
      GCGE_HIP_CHECK(hipMemcpy(xr.data(), d_x + (size_t)ci[k] * ldx + x0, m * sizeof(double),
                               hipMemcpyDeviceToHost));
      for (int j = 0; j < m; ++j) acc[j] = fma(va[k], xr[j], acc[j]);
    }
    for (int j = 0; j < m; ++j) maxerr = fmax(maxerr, fabs(acc[j] - yr[j]));
  }
  printf("verify: max abs err on sampled rows = %.3e\n", maxerr);
  return maxerr < 1e-12 ? 0 : 1;
}
