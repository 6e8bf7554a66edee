// K1 (fast path) — CSR SpMM on the back-end's internal "pad-8" matrix format.
//
// Same contract as spmm.hip (Y[:, y0:y0+m) = A X[:, x0:x0+m), reference
// app/app_ccs.c:50-139), but tuned to what the counters showed on MI355X
// (profiles/spmm_r01_notes.md): the texture-address path charges an 8-byte/lane
// load like a 16-byte one, and every L2 miss — Infinity-Cache hit or HBM —
// crosses the same ~6.2 TB/s fabric.  Therefore:
//   * every lane moves 16 B (two adjacent columns): LPR = m/2 lanes serve one
//     non-zero, so one wave instruction fetches G = 64/LPR X-row segments;
//   * the matrix is held with every row padded to a multiple of 8 non-zeros
//     (pad entries: the row's own column with value 0, so every load is
//     unconditional — a branch around a load makes hipcc drain vmcnt(0) after
//     each one), so a step of G <= 8 non-zeros never straddles a row and
//     64-entry index/value chunks stay aligned;
//   * BATCH steps are in flight per wave regardless of row boundaries;
//   * Y is written once with a cache policy chosen by the caller
//     (plain / nt / sc1 = write-through without keeping the line in L2).
//
// Row-major multivector layout: element (r,c) at data[r*ld + c]; requires m, ldx,
// ldy even and 16-byte aligned x/y column origins (callers fall back to spmm.hip
// otherwise).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gcge_hip_internal.h"

namespace gcge {

template <int ST>
__device__ __forceinline__ void store_row16(double* q, double a, double b) {
  if (ST == 2) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d t = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(t) : "memory");
  } else if (ST == 1) {
    __builtin_nontemporal_store(a, q);
    __builtin_nontemporal_store(b, q + 1);
  } else {
    *reinterpret_cast<double2*>(q) = make_double2(a, b);
  }
}

__device__ __forceinline__ double shfl_f64(double v, int src) {
  int lo = __shfl(__double2loint(v), src, 64);
  int hi = __shfl(__double2hiint(v), src, 64);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __shfl_xor(__double2loint(v), mask, 64);
  int hi = __shfl_xor(__double2hiint(v), mask, 64);
  return __hiloint2double(hi, lo);
}

// orp[r] = first OCTET (group of 8 padded non-zeros) of row r; orp[nrows] = total octets.
// DOT = 1 additionally accumulates dot_j = sum_r X[r,j] * Y[r,j] (the p.w of a CG step, with X = p
// the SpMM input): per-block partials go to dot_partial[blockIdx.x * m + j].  Blocks walk the row
// chunks grid-stride (chunk = blockIdx.x, += gridDim.x): consecutive blocks still work on
// consecutive chunks at the same time, and the number of partials stays small.
// ACC = 1 (a list of rows ADDED to what another kernel wrote; rpw <= 4): the listed rows' Y values are requested when the wave
// starts, beside the index loads — fetched at the end of each row they cost a memory round trip per row (short rows: the whole time).
template <int LPR, int BATCH, int ST, int DOT, int ACC = 0>
__global__ __launch_bounds__(256) void spmm_pad8_kernel(
    int nrows, const int* __restrict__ orp, const int* __restrict__ pcol,
    const double* __restrict__ pval, const double* __restrict__ x, size_t ldx,
    double* __restrict__ y, size_t ldy, int m, int rpw, long nchunks, double* __restrict__ dot_partial,
    const int* __restrict__ sched, int sched_len, const int* __restrict__ rowmap, int accumulate, long own0) {
  constexpr int G = 64 / LPR;  // non-zeros per wave instruction
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane / LPR;
  const int c0 = 2 * (lane % LPR);
  const bool act = c0 < m;
  const double* __restrict__ xl = x + (act ? c0 : 0);  // inactive lanes re-read column 0 (never stored)
  double d0 = 0.0, d1 = 0.0;
  // Chunk order.  Default: chunk = blockIdx.x, += gridDim.x (the chip sweeps the matrix front to back).
  // With a schedule (8 lists of sched_len chunk ids, one per XCD; -1 = empty slot) the blocks that share
  // an XCD (blockIdx.x % 8, observed round-robin placement: speed only, never correctness) walk THEIR list
  // together, so a list that enumerates the matrix brick by brick keeps a brick's X rows in that XCD's L2.
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, Jx = gridDim.x >> 3;
 for (long it = 0;; ++it) {
  long chunk;
  if (sched != nullptr) {
    const long pos = it * Jx + jx;
    if (pos >= sched_len) break;
    chunk = sched[(long)xcd * sched_len + pos];
    if (chunk < 0) continue;
  } else {
    chunk = blockIdx.x + it * (long)gridDim.x;
    if (chunk >= nchunks) break;
  }
  const long row0 = (chunk * 4 + wave) * (long)rpw;
  if (row0 >= nrows) continue;
  const int nr = min(rpw, (int)(nrows - row0));
  double* __restrict__ yl = y + (size_t)row0 * ldy + c0;
  // DOT: the wave's own rows of X (rpw <= 4), loaded up front.  own0: the row of X that belongs to row 0 of this launch — a launch
  // over a row strip [r0, r1) of a slab gathers through the WHOLE block x (columns are local row numbers) while its own rows start at r0
  double xo0[4] = {0.0, 0.0, 0.0, 0.0}, xo1[4] = {0.0, 0.0, 0.0, 0.0};
  if (DOT) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 v = *reinterpret_cast<const double2*>(xl + (size_t)(own0 + min(row0 + q, (long)nrows - 1)) * ldx);
      xo0[q] = v.x; xo1[q] = v.y;
    }
  }

  double yo0[4] = {0.0, 0.0, 0.0, 0.0}, yo1[4] = {0.0, 0.0, 0.0, 0.0};
  if (ACC) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = rowmap[min(row0 + q, (long)nrows - 1)];
      const double2 v = *reinterpret_cast<const double2*>(y + (size_t)rr * ldy + (act ? c0 : 0));
      yo0[q] = v.x; yo1[q] = v.y;
    }
  }
  const long myend = 8L * orp[row0 + 1 + min(lane, nr - 1)];
  // positions are counted relative to the wave's first padded non-zero
  const long s = 8L * orp[row0];
  const int rel_end_mine = (int)(myend - s);
  const int e = __builtin_amdgcn_readlane(rel_end_mine, nr - 1);
  int r = 0;
  int pos = 0;
  int next_end = __builtin_amdgcn_readfirstlane(rel_end_mine);
  double acc0 = 0.0, acc1 = 0.0;
  const int* __restrict__ pc = pcol + s;
  const double* __restrict__ pv = pval + s;

#define GCGE_FLUSH_ROWS()                                                              \
  while (r < nr && pos == next_end) {                                                  \
    if (G >= 2) { acc0 += shfl_xor_f64(acc0, 32); acc1 += shfl_xor_f64(acc1, 32); }    \
    if (G >= 4) { acc0 += shfl_xor_f64(acc0, 16); acc1 += shfl_xor_f64(acc1, 16); }    \
    if (G >= 8) { acc0 += shfl_xor_f64(acc0, 8); acc1 += shfl_xor_f64(acc1, 8); }      \
    if (act && g == 0) {                                                               \
      double* yq = rowmap != nullptr ? y + (size_t)rowmap[row0 + r] * ldy + c0 : yl + (size_t)r * ldy; \
      if (ACC) {   /* Y += (a listed remainder on top of what another kernel wrote) */ \
        acc0 += r == 0 ? yo0[0] : (r == 1 ? yo0[1] : (r == 2 ? yo0[2] : yo0[3]));         \
        acc1 += r == 0 ? yo1[0] : (r == 1 ? yo1[1] : (r == 2 ? yo1[2] : yo1[3]));         \
      } else if (accumulate) { const double2 yo = *reinterpret_cast<const double2*>(yq); acc0 += yo.x; acc1 += yo.y; } \
      store_row16<ST>(yq, acc0, acc1);                                                 \
    }                                                                                  \
    if (DOT) {                                                                         \
      const double o0 = r == 0 ? xo0[0] : (r == 1 ? xo0[1] : (r == 2 ? xo0[2] : xo0[3])); \
      const double o1 = r == 0 ? xo1[0] : (r == 1 ? xo1[1] : (r == 2 ? xo1[2] : xo1[3])); \
      d0 = fma(acc0, o0, d0); d1 = fma(acc1, o1, d1);                                  \
    }                                                                                  \
    acc0 = 0.0; acc1 = 0.0;                                                            \
    ++r;                                                                               \
    next_end = __builtin_amdgcn_readlane(rel_end_mine, min(r, nr - 1));                \
  }

  GCGE_FLUSH_ROWS();  // leading empty rows
  int ncol = 0;
  double nval = 0.0;
  if (lane < e) {
    ncol = __builtin_nontemporal_load(pc + lane);
    nval = __builtin_nontemporal_load(pv + lane);
  }
  for (int base = 0; base < e; base += 64) {
    const int mycol = ncol;
    const double myval = nval;
    const int cnt = min(64, e - base);  // multiple of 8
    ncol = 0; nval = 0.0;
    if (base + 64 + lane < e) {
      ncol = __builtin_nontemporal_load(pc + base + 64 + lane);
      nval = __builtin_nontemporal_load(pv + base + 64 + lane);
    }
    const int nsteps = cnt / G;
    for (int t0 = 0; t0 < nsteps; t0 += BATCH) {
      double xv0[BATCH], xv1[BATCH], av[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        // branch-free: steps past the end of the chunk re-read its last step with weight 0
        const int t = min(t0 + u, nsteps - 1);
        const int idx = t * G + g;
        const int c = __shfl(mycol, idx, 64);
        const double a = shfl_f64(myval, idx);
        av[u] = (t0 + u < nsteps) ? a : 0.0;
        const double2 v = *reinterpret_cast<const double2*>(xl + (size_t)c * ldx);
        xv0[u] = v.x; xv1[u] = v.y;
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int t = t0 + u;
        if (t < nsteps) {
          acc0 = fma(av[u], xv0[u], acc0);
          acc1 = fma(av[u], xv1[u], acc1);
          pos += G;
          GCGE_FLUSH_ROWS();
        }
      }
    }
  }
#undef GCGE_FLUSH_ROWS
 }  // chunk loop
  if (DOT) {   // block partial: every group holds the row totals, group 0 of each wave contributes
    __shared__ double sred[4][128];
    for (int e = threadIdx.x; e < 512; e += 256) (&sred[0][0])[e] = 0.0;
    __syncthreads();
    if (g == 0) { sred[wave][2 * lane] = act ? d0 : 0.0; sred[wave][2 * lane + 1] = act ? d1 : 0.0; }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < m && t < 2 * LPR) dot_partial[(long)blockIdx.x * m + t] = (sred[0][t] + sred[1][t]) + (sred[2][t] + sred[3][t]);
  }
}

}  // namespace gcge

using namespace gcge;

static int g_p8_rpw = 8, g_p8_batch = 8, g_p8_store = 1, g_p8_pass = 0, g_p8_user = 0;
// A matrix given as a LIST of rows (orp over the list, d_map[i] = row of Y the i-th listed row belongs to; the rows of Y that are
// not listed are left alone): the remainder of a split whose other rows a different kernel writes (spmm_dense.hip under
// spmm_star.hip).  Holds for the gcge_hip_pad8_spmm calls until it is reset with NULL (one stream, one caller).
static const int* g_p8_rowmap = nullptr;
static int g_p8_accumulate = 0;      // 1: Y[listed rows] += instead of =
extern "C" void gcge_hip_spmm_pad8_row_map(const int* d_map) { g_p8_rowmap = d_map; g_p8_accumulate = 0; }
extern "C" void gcge_hip_spmm_pad8_row_map_add(const int* d_map) { g_p8_rowmap = d_map; g_p8_accumulate = d_map != nullptr; }
// rows per wave from the average row length (in octets of 8 padded non-zeros), unless a caller has tuned by hand.
// Measured on the SiO2-like matrix (36 nnz/row, n = 5e6, 64 columns; profiles/r01_spmm_explore/26, 27):
// rows per wave 8 / 4 / 2 / 1 = 9.87 / 8.33 / 6.95 / 6.81 ms — long rows keep a wave busy on their own, and fewer
// rows per wave means fewer distinct X rows competing for the L1; on the 7-point stencil (1 octet per row) 8 is best.
extern "C" void gcge_hip_spmm_pad8_auto(double avg_octets_per_row) {
  if (g_p8_user) return;
  g_p8_rpw = avg_octets_per_row >= 4.0 ? 1 : (avg_octets_per_row >= 2.5 ? 2 : (avg_octets_per_row >= 1.5 ? 4 : 8));
}
extern "C" void gcge_hip_spmm_pad8_tune(int rows_per_wave, int batch, int store_policy, int col_pass) {
  g_p8_user = 1;
  if (rows_per_wave >= 1 && rows_per_wave <= 64) g_p8_rpw = rows_per_wave;
  if (batch == 4 || batch == 8 || batch == 16) g_p8_batch = batch;
  if (store_policy >= 0 && store_policy <= 2) g_p8_store = store_policy;
  if (col_pass >= 0) g_p8_pass = col_pass;  // 0: widest pass that fits (<=128 columns)
}

static int g_p8_acc_early = 1;   // adding lists request their Y rows up front (0: at the end of each row, the round-4 form)
extern "C" void gcge_hip_spmm_pad8_acc_early(int on) { g_p8_acc_early = on != 0; }
static int g_p8_gridcap = 0;   // 0: one chunk per block
extern "C" void gcge_hip_spmm_pad8_gridcap(int cap) { g_p8_gridcap = cap; }
// optional per-XCD chunk schedule (device array of 8*len ints) valid for ONE rows-per-wave value
static const int* g_p8_sched = nullptr; static int g_p8_sched_len = 0, g_p8_sched_rpw = 0, g_p8_sched_grid = 0;
extern "C" void gcge_hip_spmm_pad8_schedule(const int* d_sched, int len, int rows_per_wave, int grid) {
  g_p8_sched = d_sched; g_p8_sched_len = len; g_p8_sched_rpw = rows_per_wave; g_p8_sched_grid = grid;
}

template <int LPR, int BATCH, int ST>
static void p8_launch(int nrows, const int* orp, const int* pcol, const double* pval,
                      const double* x, size_t ldx, double* y, size_t ldy, int m, hipStream_t st) {
  if (g_p8_accumulate && g_p8_rowmap != nullptr && g_p8_acc_early) {
    const int rpw = g_p8_rpw < 4 ? g_p8_rpw : 4;
    const long nch = ((long)nrows + 4 * rpw - 1) / (4 * rpw);
    hipLaunchKernelGGL((spmm_pad8_kernel<LPR, BATCH, ST, 0, 1>), dim3((unsigned)nch), dim3(256), 0, st, nrows, orp, pcol, pval, x, ldx, y, ldy, m,
                       rpw, nch, (double*)nullptr, (const int*)nullptr, 0, g_p8_rowmap, 1, 0L);
    return;
  }
  const unsigned rows_per_block = 4u * (unsigned)g_p8_rpw;
  const long nchunks = ((long)nrows + rows_per_block - 1) / rows_per_block;
  long grid = nchunks;
  if (g_p8_gridcap > 0 && grid > g_p8_gridcap) grid = g_p8_gridcap;
  const int* sched = nullptr; int slen = 0;
  if (g_p8_sched != nullptr && g_p8_sched_rpw == g_p8_rpw) { sched = g_p8_sched; slen = g_p8_sched_len; grid = g_p8_sched_grid; }
  hipLaunchKernelGGL((spmm_pad8_kernel<LPR, BATCH, ST, 0>), dim3((unsigned)grid), dim3(256), 0, st, nrows, orp,
                     pcol, pval, x, ldx, y, ldy, m, g_p8_rpw, nchunks, (double*)nullptr, sched, slen, g_p8_rowmap, g_p8_accumulate, 0L);
}
// fused SpMM + column dots: rows_per_wave fixed at 4, at most `grid` partial rows
template <int LPR>
static void p8_launch_dot(int nrows, const int* orp, const int* pcol, const double* pval, const double* x,
                          size_t ldx, double* y, size_t ldy, int m, double* partial, long grid, hipStream_t st, long own0) {
  const int rpw = 4;   // the kernel keeps the wave's own X rows in 4 register pairs; fewer rows per wave do not pay here
  const long nchunks = ((long)nrows + 4 * rpw - 1) / (4 * rpw);
  hipLaunchKernelGGL((spmm_pad8_kernel<LPR, 4, 1, 1>), dim3((unsigned)grid), dim3(256), 0, st, nrows, orp, pcol,
                     pval, x, ldx, y, ldy, m, rpw, nchunks, partial, (const int*)nullptr, 0, (const int*)nullptr, 0, own0);
}
template <int LPR, int BATCH>
static void p8_store(int nrows, const int* orp, const int* pcol, const double* pval,
                     const double* x, size_t ldx, double* y, size_t ldy, int m, hipStream_t st) {
  switch (g_p8_store) {
    case 0: p8_launch<LPR, BATCH, 0>(nrows, orp, pcol, pval, x, ldx, y, ldy, m, st); break;
    case 2: p8_launch<LPR, BATCH, 2>(nrows, orp, pcol, pval, x, ldx, y, ldy, m, st); break;
    default: p8_launch<LPR, BATCH, 1>(nrows, orp, pcol, pval, x, ldx, y, ldy, m, st); break;
  }
}
template <int LPR>
static void p8_batch(int nrows, const int* orp, const int* pcol, const double* pval,
                     const double* x, size_t ldx, double* y, size_t ldy, int m, hipStream_t st) {
  switch (g_p8_batch) {
    case 4: p8_store<LPR, 4>(nrows, orp, pcol, pval, x, ldx, y, ldy, m, st); break;
    case 16: p8_store<LPR, 16>(nrows, orp, pcol, pval, x, ldx, y, ldy, m, st); break;
    default: p8_store<LPR, 8>(nrows, orp, pcol, pval, x, ldx, y, ldy, m, st); break;
  }
}

extern "C" double* gcge_hip_partial_ws(size_t len);
extern "C" void gcge_hip_reduce_partials(const double* d_partial, int nblocks, int len, double* d_out, void* stream);

// Y = A X and d_dots[j] = sum_r X[x_own_row0 + r, j] Y[r,j] in one pass (ncols <= 128).  -1: alignment contract not met.
// d_orp / d_y start at the launch's first row; d_x is the block the column indices point into, x_own_row0 the row of d_x that
// belongs to the launch's first row (0 for a whole matrix, r0 for the strip [r0, r1) of a split product on a row slab).
extern "C" int gcge_hip_pad8_spmm_dot(int nrows, const int* d_orp, const int* d_pcol, const double* d_pval,
                                      const double* d_x, long ldx, double* d_y, long ldy, int ncols,
                                      double* d_dots, void* stream, long x_own_row0) {
  if (nrows <= 0 || ncols <= 0) return 0;
  if (ncols > 128 || (ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int rpw = 4;
  const long nchunks = ((long)nrows + 4 * rpw - 1) / (4 * rpw);
  const long grid = nchunks < 8192 ? nchunks : 8192;
  double* part = gcge_hip_partial_ws((size_t)grid * ncols);
  if (ncols > 64) p8_launch_dot<64>(nrows, d_orp, d_pcol, d_pval, d_x, ldx, d_y, ldy, ncols, part, grid, st, x_own_row0);
  else if (ncols > 32) p8_launch_dot<32>(nrows, d_orp, d_pcol, d_pval, d_x, ldx, d_y, ldy, ncols, part, grid, st, x_own_row0);
  else if (ncols > 16) p8_launch_dot<16>(nrows, d_orp, d_pcol, d_pval, d_x, ldx, d_y, ldy, ncols, part, grid, st, x_own_row0);
  else p8_launch_dot<8>(nrows, d_orp, d_pcol, d_pval, d_x, ldx, d_y, ldy, ncols, part, grid, st, x_own_row0);
  gcge_hip_reduce_partials(part, (int)grid, ncols, d_dots, st);
  return (int)hipGetLastError();
}

// C-ABI (include/gcge_hip.h).  Returns 0 on success, -1 if the operands do not meet the
// alignment contract (caller then uses gcge_hip_csr_spmm), else a hipError_t.
extern "C" int gcge_hip_pad8_spmm(int nrows, const int* d_orp, const int* d_pcol,
                                  const double* d_pval, const double* d_x, long ldx, double* d_y,
                                  long ldy, int ncols, void* stream) {
  gcge_hip_apply_pending();
  if (nrows <= 0 || ncols <= 0) return 0;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15))
    return -1;
  hipStream_t st = (hipStream_t)stream;
  int done = 0;
  while (done < ncols) {
    int m = ncols - done;
    int cap = g_p8_pass > 0 ? g_p8_pass : 128;
    if (m > cap) m = cap;
    const double* x = d_x + done;
    double* y = d_y + done;
    if (m > 64) p8_batch<64>(nrows, d_orp, d_pcol, d_pval, x, ldx, y, ldy, m, st);
    else if (m > 32) p8_batch<32>(nrows, d_orp, d_pcol, d_pval, x, ldx, y, ldy, m, st);
    else if (m > 16) p8_batch<16>(nrows, d_orp, d_pcol, d_pval, x, ldx, y, ldy, m, st);
    else p8_batch<8>(nrows, d_orp, d_pcol, d_pval, x, ldx, y, ldy, m, st);
    done += m;
  }
  return (int)hipGetLastError();
}
