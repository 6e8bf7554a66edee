// K2 — tall-skinny Gram / projection  G(k x m) = Q[:,0:k)^T P[:,0:m)  on FP64 MFMA.
//
// Replaces the matA == NULL branch of DenseMatQtAP (reference app/app_lapack.c:64-183:
// dgemm('T','N') / dgemv / ddot over n rows) behind MultiVecLocalInnerProd
// (:299-313).  The reduction runs over the n rows of two ROW-major blocks, so an
// MFMA operand fragment is a 4-row x 16-column patch whose 16 columns are
// contiguous in memory: fragments are loaded straight from global memory
// (4 x 128-byte segments per wave instruction), no LDS staging needed.
//
//   v_mfma_f64_16x16x4_f64:  D(16x16) += A(16x4) B(4x16)
//     A: lane l holds A[i = l & 15][kk = l >> 4]      = Q[r + kk][i0 + i]
//     B: lane l holds B[kk = l >> 4][j = l & 15]      = P[r + kk][j0 + j]
//     D: lane l, reg t holds D[row = 4 t + (l >> 4)][col = l & 15]
//
// Work split: grid.x = row chunks (split-K), grid.y/z = 64 x 64 output tiles; the four
// waves of a block take interleaved 4-row steps of the chunk and combine in LDS; every
// block writes its partial tile to a slab that a second kernel sums in fixed order
// (bitwise reproducible, no float atomics on global memory).
//
// Roofline: 2 n k m flops on the FP64 MFMA pipe (78.6 TF peak) vs 8 n (k + m) bytes
// from HBM: k = m = 64 is 8 flop/B (HBM-bound at ~50 TF), k >= 256 is MFMA-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gcge_hip_internal.h"

extern "C" double* gcge_hip_partial_ws(size_t len);

namespace gcge {

typedef double v4d __attribute__((ext_vector_type(4)));

// One block: output tile rows [i0,i0+64) x cols [j0,j0+64), matrix rows [r0,r1).
__global__ __launch_bounds__(256) void gram_tile_kernel(long nrows, const double* __restrict__ q, long ldq,
    int k, const double* __restrict__ p, long ldp, int m, double* __restrict__ slab, long rows_per_chunk,
    int ntile_i, int ntile_j) {
  __shared__ double red[64 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.z * 64;
  const long r0 = (long)blockIdx.x * rows_per_chunk;
  const long r1 = min(nrows, r0 + rows_per_chunk);
  const int li = lane & 15, kk = lane >> 4;

  v4d acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  // column validity of my fragment lanes (partial tiles are zero-padded)
  bool qa[4], pb[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) { qa[a] = (i0 + 16 * a + li) < k; pb[a] = (j0 + 16 * a + li) < m; }
  // unconditional loads (a branch around a load serialises the wave on vmcnt(0)):
  // out-of-range columns read column 0, out-of-range rows read the last row, then a select zeroes them
  const double* qcol[4];
  const double* pcol[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    qcol[a] = qa[a] ? (q + i0 + 16 * a + li) : q;
    pcol[a] = pb[a] ? (p + j0 + 16 * a + li) : p;
  }

  // A wave owns 4*MS consecutive rows per macro-step: all fragment loads of the macro-step are issued
  // before its MFMAs, so several KB per wave are in flight while the previous results are consumed
  // (a 4-row step at a time leaves the loop latency-bound: measured 8 TF).
  constexpr int MS = 2;   // 4-row steps per macro-step (MS = 4 needs 288 registers: 1 wave/SIMD)
  for (long base = r0 + 4 * MS * wave; base < r1; base += 16 * MS) {
    double af[MS][4], bf[MS][4];
#pragma unroll
    for (int u = 0; u < MS; ++u) {
      const long rr = base + 4 * u + kk;
      const bool rv = rr < r1;
      const long rc = min(rr, nrows - 1);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double qv = qcol[a][rc * ldq], pv = pcol[a][rc * ldp];
        // keep the loads unconditional: without this hipcc sinks each load into its select and puts a
        // branch + s_waitcnt vmcnt(0) around every one of them (seen in the ISA; 10 TF instead of 30+)
        asm volatile("" : "+v"(qv), "+v"(pv));
        af[u][a] = (rv && qa[a]) ? qv : 0.0;
        bf[u][a] = (rv && pb[a]) ? pv : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < MS; ++u)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][a], bf[u][b], acc[a][b], 0, 0, 0);
  }

  // combine the four waves: wave 0 stores, the others add (LDS f64 atomics avoided: sequenced)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = 16 * a + 4 * t + kk, col = 16 * b + li;
            if (w == 0) red[row * 64 + col] = acc[a][b][t];
            else red[row * 64 + col] += acc[a][b][t];
          }
    }
    __syncthreads();
  }
  // slab layout: [chunk][tile_i][tile_j][64*64]
  double* out = slab + (((long)blockIdx.x * ntile_i + blockIdx.y) * ntile_j + blockIdx.z) * 4096;
  for (int e = threadIdx.x; e < 4096; e += 256) out[e] = red[e];
}

// g (row-major k x m) = sum over chunks of the slab tiles, fixed order
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ slab, int nchunks,
    int ntile_i, int ntile_j, int k, int m, double* __restrict__ g) {
  const int ti = blockIdx.y, tj = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;  // element inside the 64x64 tile
  if (e >= 4096) return;
  const int row = ti * 64 + e / 64, col = tj * 64 + e % 64;
  if (row >= k || col >= m) return;
  double s = 0.0;
  const long tile_stride = (long)ntile_i * ntile_j * 4096;
  const double* base = slab + ((long)ti * ntile_j + tj) * 4096 + e;
  for (int c = 0; c < nchunks; ++c) s += base[(long)c * tile_stride];
  g[(long)row * m + col] = s;
}

}  // namespace gcge

using namespace gcge;

extern "C" int gcge_hip_gram(int nrows, const double* d_q, long ldq, int k, const double* d_p, long ldp,
                             int m, double* d_g, void* stream) {
  if (k <= 0 || m <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (nrows <= 0) return (int)hipMemsetAsync(d_g, 0, (size_t)k * m * sizeof(double), st);
  const int ti = (k + 63) / 64, tj = (m + 63) / 64;
  // enough blocks to fill 256 CUs a few times over, chunks a multiple of 16 rows
  long nchunks = 2048 / ((long)ti * tj);
  if (nchunks < 64) nchunks = 64;
  long rpc = (((long)nrows + nchunks - 1) / nchunks + 63) / 64 * 64;
  if (rpc < 64) rpc = 64;
  nchunks = ((long)nrows + rpc - 1) / rpc;
  double* slab = gcge_hip_partial_ws((size_t)nchunks * ti * tj * 4096);
  hipLaunchKernelGGL(gram_tile_kernel, dim3((unsigned)nchunks, ti, tj), dim3(256), 0, st, (long)nrows, d_q,
                     ldq, k, d_p, ldp, m, slab, rpc, ti, tj);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(16, ti, tj), dim3(256), 0, st, slab, (int)nchunks, ti, tj, k, m,
                     d_g);
  return (int)hipGetLastError();
}
