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
// Roofline: 2 n k m flops on the FP64 MFMA pipe (78.6 TF peak; a register-only loop of the same instruction
// reaches 76.9 TF on this chip, tools/dense_bench.hip) vs 8 n (k + m) bytes from HBM: k = m = 64 is 8 flop/B
// (HBM-bound at ~50 TF), k >= 256 is MFMA-bound.  Measured 48-50 TF for k >= 128 whatever the prefetch depth,
// occupancy or tile-to-wave mapping.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gcge_hip_internal.h"

extern "C" double* gcge_hip_partial_ws(size_t len);

namespace gcge {
#include "agpr_tiles.inc"

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

// the 64 x 64 accumulator tile of a wave lives in a[0:127] by name (agpr_tiles.inc): fragment (a, b) = tile 4a + b
template <int MS>
__device__ __forceinline__ void gram_mfma(const double (&af)[MS][4], const double (&bf)[MS][4]) {
#pragma unroll
  for (int u = 0; u < MS; ++u)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) agpr_tile_mfma(4 * a + b, af[u][a], bf[u][b]);
}

// One block: TIB output tiles rows [i0,i0+64) x cols [j0,j0+64) side by side in i, matrix rows [r0,r1).
// The 4 waves are split TIB x RP (RP = 4 / TIB): wave w works on tile i = w % TIB and takes every RP-th macro-step
// of the chunk.  With TIB = 4 (k a multiple of 256: the Rayleigh-Ritz and W-projection Grams) every wave owns a
// whole 64 x 64 tile for the chunk, all four read the SAME P rows (L1 hits after the first) and nothing has to be
// combined; launching the four Q tiles as separate blocks instead re-reads P four times and makes the kernel
// bandwidth-bound (measured 48 TF whatever the prefetch depth).  Columns beyond k (or m) are not masked: those
// lanes read column 0 of the block, which only pollutes output entries the reduction kernel never reads.
template <int MS, int TIB>
__global__ __launch_bounds__(256) void gram_tile_kernel(long nrows, const double* __restrict__ q, long ldq,
    int k, const double* __restrict__ p, long ldp, int m, double* __restrict__ slab, long rows_per_chunk,
    int ntile_i, int ntile_j) {
  constexpr int RP = 4 / TIB;
  __shared__ double red[RP > 1 ? TIB * 64 * 64 : 1];
  const int lane = threadIdx.x & 63;
  // readfirstlane: tells the compiler the wave index is uniform, so row bases live in scalar registers
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tsel = wave % TIB, rpart = wave / TIB;
  const int ti = blockIdx.y * TIB + tsel;            // may be >= ntile_i for the last block in y: that wave idles
  const int i0 = ti * 64, j0 = blockIdx.z * 64;
  const long r0 = (long)blockIdx.x * rows_per_chunk;
  const long r1 = min(nrows, r0 + rows_per_chunk);
  const int li = lane & 15, kk = lane >> 4;
  const bool live = ti < ntile_i;

#pragma unroll
  for (int T = 0; T < 16; ++T) agpr_tile_zero(T);

  // per-lane fragment origins: row kk of a 4-row step, my column of each 16-column fragment
  const double* qp[4];
  const double* pp[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int qc = i0 + 16 * a + li, pc = j0 + 16 * a + li;
    qp[a] = q + (long)kk * ldq + (qc < k ? qc : 0);
    pp[a] = p + (long)kk * ldp + (pc < m ? pc : 0);
  }

  // macro-steps of 4*MS rows are dealt round-robin to the RP row parts; two register sets so the
  // loads of the next macro-step are in flight while the MFMAs of the current one issue
  const long full = (r1 - r0) / (4 * MS);
  long s = rpart;
  if (live && s < full) {
    // my macro-steps: s, s+RP, ...; processed in pairs with two register sets.  No branch and no copy inside the
    // loop (hipcc sinks loads into a conditional consumer, and a register copy waits for the prefetch); the
    // scheduling barriers keep the machine scheduler from moving the prefetch below the MFMAs it overlaps.
    const long cnt = (full - s + RP - 1) / RP, last = s + RP * (cnt - 1);
    double a0[MS][4], b0[MS][4], a1[MS][4], b1[MS][4];
    gram_load<MS>(a0, b0, qp, pp, r0 + s * (4 * MS), ldq, ldp);
    for (long i = 0; i < cnt / 2; ++i, s += 2 * RP) {
      gram_load<MS>(a1, b1, qp, pp, r0 + (s + RP) * (4 * MS), ldq, ldp);
      __builtin_amdgcn_sched_barrier(0);
      gram_mfma<MS>(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      gram_load<MS>(a0, b0, qp, pp, r0 + (s + 2 * RP < last ? s + 2 * RP : last) * (4 * MS), ldq, ldp);   // clamped
      __builtin_amdgcn_sched_barrier(0);
      gram_mfma<MS>(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (cnt & 1) gram_mfma<MS>(a0, b0);   // a0/b0 hold macro-step `last`
  }
  // the last (r1 - r0) mod 4*MS rows: predicated 4-row steps on the first row part (at most MS of them per block)
  if (live && rpart == 0) {
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
      gram_mfma<1>(af, bf);
    }
  }

  // the MFMAs above are inline asm, invisible to the hazard recogniser: let the last ones retire before the
  // accumulators are read (16 passes x 4 cycles)
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  // slab layout: [chunk][tile_i][tile_j][64*64]
  if (RP == 1) {   // every wave owns its tile: straight to the slab
    if (live) {
      double* out = slab + (((long)blockIdx.x * ntile_i + ti) * ntile_j + blockIdx.z) * 4096;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int t = 0; t < 4; ++t) out[(16 * a + 4 * t + kk) * 64 + 16 * b + li] = agpr_tile_read(4 * a + b, t);
    }
    return;
  }
  // combine the RP row parts of a tile: part 0 stores, the others add (LDS f64 atomics avoided: sequenced)
  double* mine = red + tsel * 4096;
  for (int w = 0; w < RP; ++w) {
    if (rpart == w) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = 16 * a + 4 * t + kk, col = 16 * b + li;
            const double v = agpr_tile_read(4 * a + b, t);
            if (w == 0) mine[row * 64 + col] = v;
            else mine[row * 64 + col] += v;
          }
    }
    __syncthreads();
  }
  for (int tt = 0; tt < TIB; ++tt) {
    const int tio = blockIdx.y * TIB + tt;
    if (tio >= ntile_i) break;
    double* out = slab + (((long)blockIdx.x * ntile_i + tio) * ntile_j + blockIdx.z) * 4096;
    for (int e = threadIdx.x; e < 4096; e += 256) out[e] = red[tt * 4096 + e];
  }
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

static int g_gram_ms = 2;   // 4-row steps per macro-step; with the accumulators pinned to AGPRs MS = 1, 2, 4 all give 48-50 TF at
                            // k=256, m=64, n=2^24 (profiles/r01_dense/07); MS = 2 keeps two waves per SIMD
extern "C" void gcge_hip_gram_tune(int ms) { if (ms == 1 || ms == 2 || ms == 4) g_gram_ms = ms; }

extern "C" int gcge_hip_gram(int nrows, const double* d_q, long ldq, int k, const double* d_p, long ldp,
                             int m, double* d_g, void* stream) {
  gcge_hip_apply_pending();
  if (k <= 0 || m <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (nrows <= 0) return (int)hipMemsetAsync(d_g, 0, (size_t)k * m * sizeof(double), st);
  const int ti = (k + 63) / 64, tj = (m + 63) / 64;
  // waves of a block side by side on Q tiles when the tile count divides (P rows shared through L1); an idle wave
  // (3 tiles on 4 waves) costs more than it saves: 10.6 vs 8.5 ms at k = 192 — and so do blocks of three waves, one tile
  // each (round 3, tools/gram_probe.py: 9.9 vs 8.4 ms)
  const int tib = ti % 4 == 0 ? 4 : (ti % 2 == 0 ? 2 : 1);
  const int gy = (ti + tib - 1) / tib;
  // enough blocks to fill 256 CUs a few times over, chunks a multiple of 16 rows
  static const long chunk_target = 512;   // blocks per launch aimed at: two resident blocks per CU, one round (2048: 10.83 ms at k = 256, m = 64, 512: 10.08 ms; k = 64: 3.68 -> 2.90 ms — the fixed-order reduction of the slab walks fewer chunks)
  long nchunks = chunk_target / ((long)gy * tj);
  if (nchunks < 64) nchunks = 64;
  long rpc = (((long)nrows + nchunks - 1) / nchunks + 63) / 64 * 64;
  if (rpc < 64) rpc = 64;
  nchunks = ((long)nrows + rpc - 1) / rpc;
  double* slab = gcge_hip_partial_ws((size_t)nchunks * ti * tj * 4096);
#define GCGE_GRAM(MS, TIBV) hipLaunchKernelGGL((gram_tile_kernel<MS, TIBV>), dim3((unsigned)nchunks, gy, tj), dim3(256), 0, st, \
                                               (long)nrows, d_q, ldq, k, d_p, ldp, m, slab, rpc, ti, tj)
#define GCGE_GRAM_MS(TIBV) do { if (g_gram_ms == 1) GCGE_GRAM(1, TIBV); else if (g_gram_ms == 4) GCGE_GRAM(4, TIBV); else GCGE_GRAM(2, TIBV); } while (0)
  if (tib == 4) GCGE_GRAM_MS(4); else if (tib == 2) GCGE_GRAM_MS(2); else GCGE_GRAM_MS(1);
#undef GCGE_GRAM_MS
#undef GCGE_GRAM
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(16, ti, tj), dim3(256), 0, st, slab, (int)nchunks, ti, tj, k, m,
                     d_g);
  return (int)hipGetLastError();
}
