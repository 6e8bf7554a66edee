// Ceiling study for the CG streaming kernels (3 reads + 2 writes, n x 64 doubles each).  Not part of the product.
// hipcc -O3 --offload-arch=gfx950 tools/stream_bench.hip -o /tmp/stream_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t _r = (e); if (_r != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(_r), __LINE__); exit(1);} } while (0)

template <int UNR, int NT>
__global__ __launch_bounds__(256) void xp_rows(long nrows, const double* __restrict__ r, long ldr, double* __restrict__ p, long ldp,
                                               double* __restrict__ x, long ldx, double a, double b) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = 2 * tx;
  const long step = (long)gridDim.x * 8;
  long row = (long)blockIdx.x * 8 + ty;
  for (; row + (UNR - 1) * step < nrows; row += step * UNR) {
    v2d pv[UNR], rv[UNR], xv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long rr = row + u * step;
      if (NT) {
        pv[u] = __builtin_nontemporal_load((const v2d*)(p + rr * ldp + j));
        rv[u] = __builtin_nontemporal_load((const v2d*)(r + rr * ldr + j));
        xv[u] = __builtin_nontemporal_load((const v2d*)(x + rr * ldx + j));
      } else {
        pv[u] = *(const v2d*)(p + rr * ldp + j); rv[u] = *(const v2d*)(r + rr * ldr + j); xv[u] = *(const v2d*)(x + rr * ldx + j);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long rr = row + u * step;
      v2d xo = {fma(a, pv[u].x, xv[u].x), fma(a, pv[u].y, xv[u].y)};
      v2d po = {fma(b, pv[u].x, rv[u].x), fma(b, pv[u].y, rv[u].y)};
      if (NT) { __builtin_nontemporal_store(xo, (v2d*)(x + rr * ldx + j)); __builtin_nontemporal_store(po, (v2d*)(p + rr * ldp + j)); }
      else { *(v2d*)(x + rr * ldx + j) = xo; *(v2d*)(p + rr * ldp + j) = po; }
    }
  }
}
// block-contiguous variant: a block owns a contiguous run of rows (no grid-stride interleaving between blocks)
template <int UNR, int NT>
__global__ __launch_bounds__(256) void xp_slab(long nrows, const double* __restrict__ r, long ldr, double* __restrict__ p, long ldp,
                                               double* __restrict__ x, long ldx, double a, double b, long rows_per_block) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = 2 * tx;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
  long row = r0 + ty;
  for (; row + (UNR - 1) * 8 < r1; row += 8 * UNR) {
    v2d pv[UNR], rv[UNR], xv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long rr = row + u * 8;
      if (NT) {
        pv[u] = __builtin_nontemporal_load((const v2d*)(p + rr * ldp + j));
        rv[u] = __builtin_nontemporal_load((const v2d*)(r + rr * ldr + j));
        xv[u] = __builtin_nontemporal_load((const v2d*)(x + rr * ldx + j));
      } else {
        pv[u] = *(const v2d*)(p + rr * ldp + j); rv[u] = *(const v2d*)(r + rr * ldr + j); xv[u] = *(const v2d*)(x + rr * ldx + j);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long rr = row + u * 8;
      v2d xo = {fma(a, pv[u].x, xv[u].x), fma(a, pv[u].y, xv[u].y)};
      v2d po = {fma(b, pv[u].x, rv[u].x), fma(b, pv[u].y, rv[u].y)};
      if (NT) { __builtin_nontemporal_store(xo, (v2d*)(x + rr * ldx + j)); __builtin_nontemporal_store(po, (v2d*)(p + rr * ldp + j)); }
      else { *(v2d*)(x + rr * ldx + j) = xo; *(v2d*)(p + rr * ldp + j) = po; }
    }
  }
}
template <int UNR>
__global__ __launch_bounds__(256) void r_rows(long nrows, const double* __restrict__ w, long ldw, double* __restrict__ r, long ldr,
                                              double a, double* __restrict__ partial) {
  __shared__ double red[256][2];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = 2 * tx;
  const long step = (long)gridDim.x * 8;
  double s0 = 0, s1 = 0;
  long row = (long)blockIdx.x * 8 + ty;
  for (; row + (UNR - 1) * step < nrows; row += step * UNR) {
    v2d wv[UNR], rv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long rr = row + u * step;
      wv[u] = __builtin_nontemporal_load((const v2d*)(w + rr * ldw + j));
      rv[u] = __builtin_nontemporal_load((const v2d*)(r + rr * ldr + j));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long rr = row + u * step;
      v2d ro = {fma(-a, wv[u].x, rv[u].x), fma(-a, wv[u].y, rv[u].y)};
      __builtin_nontemporal_store(ro, (v2d*)(r + rr * ldr + j));
      s0 = fma(ro.x, ro.x, s0); s1 = fma(ro.y, ro.y, s1);
    }
  }
  red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1;
  __syncthreads();
  if (ty == 0) { for (int q = 1; q < 8; ++q) { s0 += red[q * 32 + tx][0]; s1 += red[q * 32 + tx][1]; }
    partial[(long)blockIdx.x * 64 + j] = s0; partial[(long)blockIdx.x * 64 + j + 1] = s1; }
}
__global__ void copyk(const v2d* __restrict__ a, v2d* __restrict__ b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = a[i];
}
__global__ void readk(const v2d* __restrict__ a, double* out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  double s = 0; for (; i < n; i += st) { v2d v = a[i]; s += v.x + v.y; }
  if (s == 1.2345) out[0] = s;
}
__global__ void writek(v2d* __restrict__ b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x, st = (long)gridDim.x * blockDim.x;
  v2d z = {1.0, 2.0}; for (; i < n; i += st) b[i] = z;
}
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 16777216; long ldx = argc > 2 ? atol(argv[2]) : 264;
  double *r, *p, *x, *out; const long m = 64;
  CK(hipMalloc(&r, n * m * 8)); CK(hipMalloc(&p, n * m * 8)); CK(hipMalloc(&x, n * ldx * 8)); CK(hipMalloc(&out, 64));
  CK(hipMemset(r, 0, n * m * 8)); CK(hipMemset(p, 0, n * m * 8)); CK(hipMemset(x, 0, n * ldx * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb5 = 5.0 * n * m * 8 * 1e-9, gb1 = n * m * 8 * 1e-9;
  auto timeit = [&](const char* name, double gb, auto&& launch) {
    launch(); CK(hipDeviceSynchronize());
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3);
  };
  for (int g : {2048, 8192, 65536}) {
    char nm[96];
    snprintf(nm, 96, "read 1 stream grid %d", g); timeit(nm, gb1, [&] { readk<<<g, 256>>>((const v2d*)r, out, n * m / 2); });
    snprintf(nm, 96, "write 1 stream grid %d", g); timeit(nm, gb1, [&] { writek<<<g, 256>>>((v2d*)p, n * m / 2); });
    snprintf(nm, 96, "copy 1R+1W grid %d", g); timeit(nm, 2 * gb1, [&] { copyk<<<g, 256>>>((const v2d*)r, (v2d*)p, n * m / 2); });
  }
  { double* part; CK(hipMalloc(&part, 65536 * 64 * 8));
    for (long sh : {0L, 512L, 2048L, 8192L + 512, 65536L + 4096 + 256, 1048576L + 65536 + 4096}) {   // w shifted by sh doubles against r
      char nm[96]; double* w2; CK(hipMalloc(&w2, (n * m + sh) * 8)); CK(hipMemset(w2, 0, (n * m + sh) * 8));
      snprintf(nm, 96, "r-update unr4 grid 2048, w shifted %ld B", sh * 8); timeit(nm, 3 * gb1, [&] { r_rows<4><<<2048, 256>>>(n, w2 + sh, m, r, m, 0.5, part); });
      CK(hipFree(w2));
    }
    for (int g : {2048}) {
      char nm[96];
      snprintf(nm, 96, "r-update 2R+1W unr4 grid %d", g); timeit(nm, 3 * gb1, [&] { r_rows<4><<<g, 256>>>(n, p, m, r, m, 0.5, part); });
      snprintf(nm, 96, "r-update 2R+1W unr8 grid %d", g); timeit(nm, 3 * gb1, [&] { r_rows<8><<<g, 256>>>(n, p, m, r, m, 0.5, part); });
    }
  }
  if (getenv("SKEW")) {   // xp update with r placed at different offsets behind p inside ONE allocation, x as in the solver (ld 256)
    double* big; const long nb = n * m;
    CK(hipMalloc(&big, (2 * nb + (1L << 24)) * 8)); CK(hipMemset(big, 0, (2 * nb + (1L << 24)) * 8));
    for (long sk : {0L, 32L, 512L, 4096L + 32, 65536L, 65536L + 512, 1L << 20, (1L << 20) + 4096 + 64, (1L << 23) + 8192}) {
      char nm[96]; snprintf(nm, 96, "xp unr4 nt grid 4096 ldx 256, r = p + n*m + %ld doubles", sk);
      timeit(nm, gb5, [&] { xp_rows<4, 1><<<4096, 256>>>(n, big + nb + sk, m, big, m, x, 256, 0.5, 0.25); });
    }
    CK(hipFree(big));
  }
  if (getenv("XP")) for (long ld : {64L, ldx}) for (int g : {2048, 4096, 16384, 65536}) {
    char nm[96];
    snprintf(nm, 96, "xp rows unr4 nt  ldx %ld grid %d", ld, g); timeit(nm, gb5, [&] { xp_rows<4, 1><<<g, 256>>>(n, r, m, p, m, x, ld, 0.5, 0.25); });
    snprintf(nm, 96, "xp rows unr4 pl  ldx %ld grid %d", ld, g); timeit(nm, gb5, [&] { xp_rows<4, 0><<<g, 256>>>(n, r, m, p, m, x, ld, 0.5, 0.25); });
    snprintf(nm, 96, "xp rows unr8 nt  ldx %ld grid %d", ld, g); timeit(nm, gb5, [&] { xp_rows<8, 1><<<g, 256>>>(n, r, m, p, m, x, ld, 0.5, 0.25); });
    snprintf(nm, 96, "xp rows unr2 nt  ldx %ld grid %d", ld, g); timeit(nm, gb5, [&] { xp_rows<2, 1><<<g, 256>>>(n, r, m, p, m, x, ld, 0.5, 0.25); });
    long rpb = (n + g - 1) / g; rpb = (rpb + 31) / 32 * 32;
    snprintf(nm, 96, "xp slab unr4 nt  ldx %ld grid %d", ld, g); timeit(nm, gb5, [&] { xp_slab<4, 1><<<g, 256>>>(n, r, m, p, m, x, ld, 0.5, 0.25, rpb); });
    snprintf(nm, 96, "xp slab unr4 pl  ldx %ld grid %d", ld, g); timeit(nm, gb5, [&] { xp_slab<4, 0><<<g, 256>>>(n, r, m, p, m, x, ld, 0.5, 0.25, rpb); });
  }
  return 0;
}
