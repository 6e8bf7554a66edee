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

// Fragment loads of one macro-step (4*MS rows): no predication at all, so the compiler is free to keep
// all 8*MS loads of a wave in flight (any select/branch next to a load makes hipcc wait with vmcnt(0)
// after every pair of loads: measured 15 TF).  `row` is wave-uniform.
template <int MS>
__device__ __forceinline__ void gram_load(double (&af)[MS][4], double (&bf)[MS][4], const double* const (&qp)[4],
                                          const double* const (&pp)[4], long row, long ldq, long ldp) {
#pragma unroll
  for (int u = 0; u < MS; ++u)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      af[u][a] = qp[a][(row + 4 * u) * ldq];
      bf[u][a] = pp[a][(row + 4 * u) * ldp];
    }
}

template <int MS>
__device__ __forceinline__ void gram_mfma(v4d (&acc)[4][4], const double (&af)[MS][4], const double (&bf)[MS][4]) {
#pragma unroll
  for (int u = 0; u < MS; ++u)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][a], bf[u][b], acc[a][b], 0, 0, 0);
}

// One block: output tile rows [i0,i0+64) x cols [j0,j0+64), matrix rows [r0,r1).
// Columns beyond k (or m) are not masked: those lanes read column 0 of the block instead, which only
// pollutes output entries (i >= k or j >= m) that the reduction kernel never reads.
template <int MS>
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

  // per-lane fragment origins: row kk of a 4-row step, my column of each 16-column fragment
  const double* qp[4];
  const double* pp[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int qc = i0 + 16 * a + li, pc = j0 + 16 * a + li;
    qp[a] = q + (long)kk * ldq + (qc < k ? qc : 0);
    pp[a] = p + (long)kk * ldp + (pc < m ? pc : 0);
  }

  // macro-steps of 4*MS rows are dealt round-robin to the four waves; two register sets so the
  // loads of the next macro-step are in flight while the MFMAs of the current one issue
  const long full = (r1 - r0) / (4 * MS);
  long s = wave;
  if (s < full) {
    // my macro-steps: s, s+4, ...; processed in pairs with two register sets.  No branch and no copy inside the
    // loop (hipcc sinks loads into a conditional consumer, and a register copy waits for the prefetch); the
    // scheduling barriers keep the machine scheduler from moving the prefetch below the MFMAs it overlaps.
    const long cnt = (full - s + 3) / 4, last = s + 4 * (cnt - 1);
    double a0[MS][4], b0[MS][4], a1[MS][4], b1[MS][4];
    gram_load<MS>(a0, b0, qp, pp, r0 + s * (4 * MS), ldq, ldp);
    for (long i = 0; i < cnt / 2; ++i, s += 8) {
      gram_load<MS>(a1, b1, qp, pp, r0 + (s + 4) * (4 * MS), ldq, ldp);
      __builtin_amdgcn_sched_barrier(0);
      gram_mfma<MS>(acc, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      gram_load<MS>(a0, b0, qp, pp, r0 + (s + 8 < last ? s + 8 : last) * (4 * MS), ldq, ldp);   // clamped
      __builtin_amdgcn_sched_barrier(0);
      gram_mfma<MS>(acc, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (cnt & 1) gram_mfma<MS>(acc, a0, b0);   // a0/b0 hold macro-step `last`
  }
  // the last (r1 - r0) mod 4*MS rows: predicated 4-row steps on wave 0 (at most MS of them per block)
  if (wave == 0) {
    for (long base = r0 + full * (4 * MS); base < r1; base += 4) {
      const bool rv = base + kk < r1;
      const long off = rv ? base : (r1 - 1 - kk);   // any valid row; the select below zeroes it
      double af[1][4], bf[1][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double qv = qp[a][off * ldq], pv = pp[a][off * ldp];
        asm volatile("" : "+v"(qv), "+v"(pv));
        af[0][a] = rv ? qv : 0.0;
        bf[0][a] = rv ? pv : 0.0;
      }
      gram_mfma<1>(acc, af, bf);
    }
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

static int g_gram_ms = 4;   // measured: MS 1/2/4 = 36.8 / 42.3 / 45.2 TF at k=256, m=64, n=2^24 (profiles/r01_dense)
extern "C" void gcge_hip_gram_tune(int ms) { if (ms == 1 || ms == 2 || ms == 4) g_gram_ms = ms; }

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
#define GCGE_GRAM(MS) hipLaunchKernelGGL(gram_tile_kernel<MS>, dim3((unsigned)nchunks, ti, tj), dim3(256), 0, st, \
                                         (long)nrows, d_q, ldq, k, d_p, ldp, m, slab, rpc, ti, tj)
  if (g_gram_ms == 1) GCGE_GRAM(1); else if (g_gram_ms == 4) GCGE_GRAM(4); else GCGE_GRAM(2);
#undef GCGE_GRAM
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(16, ti, tj), dim3(256), 0, st, slab, (int)nchunks, ti, tj, k, m,
                     d_g);
  return (int)hipGetLastError();
}
