// Rate at which SEGMENTS of the rows of a row-major block of vectors come in: every pass reads (and optionally writes back to a
// second block) w consecutive doubles of every row of an n x 64 block, the 64 / w passes of a sweep run side by side
// (blockIdx.y) as the column passes of spmm_star.hip do.  Measurement aid: what bounds a kernel that touches 64 bytes of every
// 512-byte row at a time.      hipcc -O3 --offload-arch=gfx950 tools/seg_bench.hip -o tools/_bin/seg_bench && tools/_bin/seg_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

// LPR lanes per row (16 B each): w = 2 LPR doubles.  One block = 256 threads = 256 / LPR rows per iteration; rows blocked per workgroup
template <int LPR, bool WRITE>
__global__ __launch_bounds__(256) void seg_kernel(const double* __restrict__ x, double* __restrict__ y, long n, long rows_per_block, double* sink) {
  const int lane = threadIdx.x % LPR, rl = threadIdx.x / LPR;
  constexpr int RPI = 256 / LPR;
  const int col = blockIdx.y * 2 * LPR + 2 * lane;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  v2d acc = {0.0, 0.0};
  long r = r0 + rl;
  for (; r + 3 * RPI < r1; r += 4 * RPI) {
    const v2d a = *reinterpret_cast<const v2d*>(x + r * 64 + col);
    const v2d b = *reinterpret_cast<const v2d*>(x + (r + RPI) * 64 + col);
    const v2d c = *reinterpret_cast<const v2d*>(x + (r + 2 * RPI) * 64 + col);
    const v2d d = *reinterpret_cast<const v2d*>(x + (r + 3 * RPI) * 64 + col);
    if (WRITE) {
      __builtin_nontemporal_store(a, reinterpret_cast<v2d*>(y + r * 64 + col));
      __builtin_nontemporal_store(b, reinterpret_cast<v2d*>(y + (r + RPI) * 64 + col));
      __builtin_nontemporal_store(c, reinterpret_cast<v2d*>(y + (r + 2 * RPI) * 64 + col));
      __builtin_nontemporal_store(d, reinterpret_cast<v2d*>(y + (r + 3 * RPI) * 64 + col));
    } else { acc += a; acc += b; acc += c; acc += d; }
  }
  for (; r < r1; r += RPI) {
    const v2d a = *reinterpret_cast<const v2d*>(x + r * 64 + col);
    if (WRITE) __builtin_nontemporal_store(a, reinterpret_cast<v2d*>(y + r * 64 + col)); else acc += a;
  }
  if (!WRITE && acc.x + acc.y == 1.2345e-300) sink[0] = acc.x;
}

template <int LPR, bool WRITE>
static void run(const double* x, double* y, long n, double* sink, int nblk) {
  const long rpb = (n + nblk - 1) / nblk;
  dim3 grid(nblk, 32 / LPR);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((seg_kernel<LPR, WRITE>), grid, dim3(256), 0, 0, x, y, n, rpb, sink);
  CK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((seg_kernel<LPR, WRITE>), grid, dim3(256), 0, 0, x, y, n, rpb, sink);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)n * 512 * (WRITE ? 2 : 1);
  printf("segment %3d B, %s, %5d x %2d blocks: %.3f ms  %.2f TB/s\n", LPR * 16, WRITE ? "read + write" : "read only   ", nblk, 32 / LPR, ms, bytes / ms * 1e-9);
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 5000211;
  double *x, *y, *sink;
  CK(hipMalloc(&x, n * 512)); CK(hipMalloc(&y, n * 512)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(x, 0, n * 512)); CK(hipMemset(y, 0, n * 512));
  for (int nblk : {256, 1024, 4096}) {
    run<4, false>(x, y, n, sink, nblk); run<8, false>(x, y, n, sink, nblk); run<16, false>(x, y, n, sink, nblk); run<32, false>(x, y, n, sink, nblk);
    run<4, true>(x, y, n, sink, nblk); run<8, true>(x, y, n, sink, nblk); run<16, true>(x, y, n, sink, nblk); run<32, true>(x, y, n, sink, nblk);
  }
  return 0;
}
