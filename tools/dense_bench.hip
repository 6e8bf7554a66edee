// Micro-benchmark of K2 (Gram) and K3 (panel update) at the C2 shapes.  Not part of the product.
// hipcc -O3 --offload-arch=gfx950 -I gcge_amd/csrc/hip -I include tools/dense_bench.hip gcge_amd/csrc/hip/gram_mfma.hip gcge_amd/csrc/hip/lincomb_mfma.hip gcge_amd/csrc/hip/vec_kernels.hip -o tools/_bin/dense_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gcge_hip_internal.h"
extern "C" int gcge_hip_gram(int, const double*, long, int, const double*, long, int, double*, void*);
extern "C" void gcge_hip_gram_tune(int);
extern "C" void gcge_hip_lincomb_tune(int);
extern "C" void gcge_hip_lincomb_wide_only(int);
extern "C" void gcge_hip_apply_pending(void) {}   // (the back-end's held-back column scaling: nothing pending in a stand-alone bench)
extern "C" int gcge_hip_lincomb(int, const double*, long, int, const double*, int, const double*, double*, long, void*);
__global__ void fillk(double* x, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) { unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    x[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
}
// practical FP64 MFMA ceiling: 16 independent accumulator tiles per wave pinned to a[0:127], operands in registers
#include "agpr_tiles.inc"
__global__ __launch_bounds__(256) void mfma_peak(double* out, int iters, double seed) {
#pragma unroll
  for (int t = 0; t < 16; ++t) agpr_tile_zero(t);
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + threadIdx.x * 2e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 16; ++t) agpr_tile_mfma(t, a, b);
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  double s = 0;
#pragma unroll
  for (int t = 0; t < 16; ++t) s += agpr_tile_read(t, 0);
  if (s == 1.2345) out[0] = s;
}
// streams `bytes` of HBM traffic (read + write) with 16-byte lanes, `reps` times
__global__ __launch_bounds__(256) void hbm_stream(const double2* __restrict__ src, double2* __restrict__ dst, size_t n2, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
      double2 v = src[i]; v.x += 1e-30; dst[i] = v;
    }
}
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 16777216; int reps = 3;
  if (getenv("LC_RF")) gcge_hip_lincomb_tune(atoi(getenv("LC_RF")));   // row fragments per wave of the panel update
  const long ldv = getenv("DB_LDV") ? atol(getenv("DB_LDV")) : 256, ldw = 128;
  double *V, *W, *G, *C;
  GCGE_HIP_CHECK(hipMalloc(&V, n * ldv * 8)); GCGE_HIP_CHECK(hipMalloc(&W, n * ldw * 8));
  GCGE_HIP_CHECK(hipMalloc(&G, 656 * 656 * 8)); GCGE_HIP_CHECK(hipMalloc(&C, 656 * 128 * 8));
  fillk<<<4096, 256>>>(V, n * ldv); fillk<<<4096, 256>>>(W, n * ldw); fillk<<<64, 256>>>(C, 656 * 128);
  GCGE_HIP_CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {4, 8}) {   // waves per CU: 1 or 2 per SIMD
    const int iters = 20000; const int blocks = 256 * wpc / 4 * 4;   // several blocks per CU
    mfma_peak<<<blocks, 256>>>(G, 100, 0.37); hipDeviceSynchronize();
    hipEventRecord(e0); mfma_peak<<<blocks, 256>>>(G, iters, 0.37); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mfma f64 16x16x4 register-only: %d blocks x 4 waves, %.2f ms -> %.1f TF\n", blocks, ms, 2048.0 * 16 * iters * blocks * 4 / ms * 1e-9);
  }
  {   // Does the FP64 MFMA rate survive concurrent HBM traffic?  Register-only loop (2 waves per SIMD) alone, and next to a
      // copy kernel on a second stream that moves ~4-5 TB/s.  (If it does not, the 46-50 TF of the Gram / panel-update
      // kernels are a chip-level limit under combined load, not a property of the kernels.)
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    const int iters = 20000, blocks = 2048;
    const size_t n2 = (size_t)n * 64 / 2;     // 8.6 GB read + 8.6 GB written per repetition
    hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    hbm_stream<<<1024, 256, 0, sb>>>((const double2*)V, (double2*)W, n2, 1); hipDeviceSynchronize();
    hipEventRecord(b0, sb); hbm_stream<<<1024, 256, 0, sb>>>((const double2*)V, (double2*)W, n2, 4); hipEventRecord(b1, sb);
    hipEventSynchronize(b1); float tb; hipEventElapsedTime(&tb, b0, b1);
    printf("copy alone: %.2f ms -> %.0f GB/s\n", tb, 4 * 2.0 * n2 * 16 / tb * 1e-6);
    hipEventRecord(b0, sb); hbm_stream<<<1024, 256, 0, sb>>>((const double2*)V, (double2*)W, n2, 24); hipEventRecord(b1, sb);
    hipEventRecord(a0, sa); mfma_peak<<<blocks, 256, 0, sa>>>(G, iters, 0.37); hipEventRecord(a1, sa);
    hipEventSynchronize(a1); hipEventSynchronize(b1);
    float ta; hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb, b0, b1);
    printf("mfma f64 16x16x4 register-only NEXT TO an HBM copy: %.2f ms -> %.1f TF  (copy: 24 reps in %.2f ms -> %.0f GB/s while both ran)\n",
           ta, 2048.0 * 16 * iters * blocks * 4 / ta * 1e-9, tb, 24 * 2.0 * n2 * 16 / tb * 1e-6);
  }
  if (getenv("DB_RAGGED")) {   // panels of 65 .. 127 columns: 5 / 6 / 7 column fragments (round 4) against the 128-column kernel
    for (int k : {208, 256}) for (int m : {66, 72, 80, 84, 90, 96, 102, 108, 112, 120, 128}) for (int wide : {1, 0}) {
      gcge_hip_lincomb_wide_only(wide);
      gcge_hip_lincomb((int)n, V, ldv, k, C, m, nullptr, W, ldw, 0); hipDeviceSynchronize();
      hipEventRecord(e0); for (int r = 0; r < reps; ++r) gcge_hip_lincomb((int)n, V, ldv, k, C, m, nullptr, W, ldw, 0);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("lincomb k=%3d m=%3d %s  %8.3f ms  %6.1f TF\n", k, m, wide ? "128-column kernel" : "fragments of m   ", ms, 2.0 * n * k * m / ms * 1e-9);
    }
    return 0;
  }
  int gk[] = {256, 192, 128, 64, 64, 256, 512, 512}, gm[] = {64, 64, 64, 64, 1, 128, 128, 64};
  for (int ms = 1; ms <= 4; ms *= 2)
  for (int i = 0; i < 8; ++i) {
    if (gk[i] > ldv) continue;
    int k = gk[i], m = gm[i];
    gcge_hip_gram_tune(ms); if (i == 0) printf("gram MS=%d\n", ms);
    gcge_hip_gram((int)n, V, ldv, k, W, ldw, m, G, 0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; ++r) gcge_hip_gram((int)n, V, ldv, k, W, ldw, m, G, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("gram    k=%3d m=%3d  %8.3f ms  %6.1f TF  %7.1f GB/s (min traffic)\n", k, m, ms, 2.0 * n * k * m / ms * 1e-9, 8.0 * n * (k + m) / ms * 1e-6);
  }
  int lk[] = {256, 256, 192, 64, 1, 512, 512}, lm[] = {128, 64, 64, 64, 63, 128, 64};
  for (int i = 0; i < 7; ++i) {
    if (lk[i] > ldv) continue;
    int k = lk[i], m = lm[i];
    gcge_hip_lincomb((int)n, V, ldv, k, C, m, nullptr, W, ldw, 0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; ++r) gcge_hip_lincomb((int)n, V, ldv, k, C, m, nullptr, W, ldw, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("lincomb k=%3d m=%3d  %8.3f ms  %6.1f TF  %7.1f GB/s (min traffic)\n", k, m, ms, 2.0 * n * k * m / ms * 1e-9, 8.0 * n * (k + m) / ms * 1e-6);
  }
  return 0;
}
