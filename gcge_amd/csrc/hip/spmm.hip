// K1 — CSR SpMM on a block of vectors:  Y[:, y0:y0+m) = A * X[:, x0:x0+m)
//
// Replaces the reference's CCS scatter product  app/app_ccs.c:50-139
// (MatDotMultiVec: zero Y, then for every column j of A scatter
//  y[i_row] += a * x[j], OpenMP over the m block columns).  For the symmetric
// matrices GCGE handles CCS(A) == CSR(A), so the same three arrays are read
// here row-wise: no scatter, no atomics, Y written exactly once.
//
// Device layout of a block of vectors ("multivector"): ROW-major,
//   element (row r, column c)  ->  data[r * ld + c],
// so the m values of one matrix row needed by one non-zero are contiguous
// (m = 64 doubles = 512 B = one 8-byte load per lane of a 64-wide wavefront).
//
// Roofline: HBM.  Algorithmic bytes per launch:
//   12*nnz + 4*(n+1) + 16*n*m     (values+indices once, X read once, Y written once)
//
// Two kernels:
//   spmm_wave_row<CPL>   one wavefront per matrix row, CPL (1|2) columns per lane,
//                        covers m in (32, 64*CPL]; non-zeros of the row are fetched
//                        by one coalesced load and broadcast with v_readlane.
//   spmm_subwave<LPR>    LPR lanes per row (LPR = 1..32) for narrow column runs
//                        (BlockPCG hands over runs of 1..b unconverged columns).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gcge_hip_internal.h"

namespace gcge {

__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// XCD-aware logical block id.  The dispatcher deals blocks round-robin over the
// 8 XCDs (b and b+8 share an L2).  With group G > 1, runs of G consecutive
// logical blocks are kept on one XCD so that the +-N stencil neighbours of a
// row chunk are served by the same L2, while the chip as a whole still walks
// the matrix front to back (keeps the +-N^2 planes inside the Infinity Cache).
// Bijective on [0, 8*G*ceil(nb/(8G))); callers skip ids >= nb.  Speed only.
__device__ __forceinline__ unsigned logical_block(unsigned b, unsigned G) {
  if (G <= 1) return b;
  unsigned k = b >> 3, x = b & 7u;
  return ((k / G) * 8u + x) * G + (k % G);
}

template <int CPL, int ROWS_PER_WAVE, int NT>
__global__ __launch_bounds__(256) void spmm_wave_row(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ colidx,
    const double* __restrict__ val, const double* __restrict__ x, size_t ldx,
    double* __restrict__ y, size_t ldy, int m, unsigned nblocks, unsigned xcd_group,
    const int* __restrict__ chunk_map) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned lb;
  if (chunk_map != nullptr) {
    const int cm = chunk_map[blockIdx.x];
    if (cm < 0) return;
    lb = (unsigned)cm;
  } else {
    lb = logical_block(blockIdx.x, xcd_group);
  }
  if (lb >= nblocks) return;
  int row = (int)((lb * 4u + (unsigned)wave) * (unsigned)ROWS_PER_WAVE);
  const int row_end = min(row + ROWS_PER_WAVE, nrows);
  if (row >= row_end) return;
  const int c0 = lane * CPL;
  const bool act = c0 < m;
  const double* __restrict__ xl = x + c0;

  // software-prefetched row descriptor + first 64 non-zeros of the next row
  int s = rowptr[row], e = rowptr[row + 1];
  int mycol = 0;
  double myval = 0.0;
  if (s + lane < e) {
    mycol = __builtin_nontemporal_load(colidx + s + lane);
    myval = __builtin_nontemporal_load(val + s + lane);
  }
  for (; row < row_end; ++row) {
    const int cs = s, ce = e;
    const int ccol = mycol;
    const double cval = myval;
    if (row + 1 < row_end) {
      s = e;
      e = rowptr[row + 2];
      mycol = 0;
      myval = 0.0;
      if (s + lane < e) {
        mycol = __builtin_nontemporal_load(colidx + s + lane);
        myval = __builtin_nontemporal_load(val + s + lane);
      }
    }
    double acc0 = 0.0, acc1 = 0.0;
    int cnt = min(64, ce - cs);
    int k = 0;
    // batches of 8 independent X-row loads in flight per lane
    for (; k + 8 <= cnt; k += 8) {
      double xv[8][CPL];
      double av[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = __builtin_amdgcn_readlane(ccol, k + u);
        av[u] = readlane_f64(cval, k + u);
        const double* p = xl + (size_t)c * ldx;
        if (CPL == 2) {
          double2 t = act ? *reinterpret_cast<const double2*>(p) : make_double2(0.0, 0.0);
          xv[u][0] = t.x;
          xv[u][CPL - 1] = t.y;
        } else {
          xv[u][0] = act ? *p : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc0 = fma(av[u], xv[u][0], acc0);
        if (CPL == 2) acc1 = fma(av[u], xv[u][CPL - 1], acc1);
      }
    }
    if (k < cnt) {
      double xv[8][CPL];
      double av[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool on = (k + u) < cnt;
        const int c = on ? __builtin_amdgcn_readlane(ccol, (k + u) & 63) : 0;
        av[u] = on ? readlane_f64(cval, (k + u) & 63) : 0.0;
        const double* p = xl + (size_t)c * ldx;
        if (CPL == 2) {
          double2 t = (act && on) ? *reinterpret_cast<const double2*>(p) : make_double2(0.0, 0.0);
          xv[u][0] = t.x;
          xv[u][CPL - 1] = t.y;
        } else {
          xv[u][0] = (act && on) ? *p : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc0 = fma(av[u], xv[u][0], acc0);
        if (CPL == 2) acc1 = fma(av[u], xv[u][CPL - 1], acc1);
      }
    }
    // rows longer than 64 non-zeros: remaining chunks (irregular matrices)
    for (int base = cs + 64; base < ce; base += 64) {
      int lc = 0;
      double lv = 0.0;
      if (base + lane < ce) {
        lc = colidx[base + lane];
        lv = val[base + lane];
      }
      const int n2 = min(64, ce - base);
      for (int q = 0; q < n2; ++q) {
        const int c = __builtin_amdgcn_readlane(lc, q);
        const double a = readlane_f64(lv, q);
        const double* p = xl + (size_t)c * ldx;
        if (act) {
          if (CPL == 2) {
            double2 t = *reinterpret_cast<const double2*>(p);
            acc0 = fma(a, t.x, acc0);
            acc1 = fma(a, t.y, acc1);
          } else {
            acc0 = fma(a, *p, acc0);
          }
        }
      }
    }
    if (act) {
      double* q = y + (size_t)row * ldy + c0;
      if (CPL == 2) {
        double2 t = make_double2(acc0, acc1);
        if (NT) __builtin_nontemporal_store(t.x, q), __builtin_nontemporal_store(t.y, q + 1);
        else *reinterpret_cast<double2*>(q) = t;
      } else {
        if (NT) __builtin_nontemporal_store(acc0, q);
        else *q = acc0;
      }
    }
  }
}

// spmm_stream — the wide-block kernel (m in (32, 64*CPL]).
// A wavefront owns `rpw` (<= 64) consecutive rows and walks their non-zeros as ONE
// flat stream: 64 (col,val) pairs per coalesced load, broadcast with v_readlane, and
// BATCH independent X-row loads in flight at any time regardless of where the row
// boundaries fall (a 7-point row alone keeps only 1 HBM miss in flight per wave and
// leaves the kernel latency-bound: measured 35 % of HBM peak; see profiles/).
// Row ends are held one per lane and compared on the scalar unit; a row's result is
// stored as soon as its last non-zero has been consumed.
template <int CPL, int BATCH, int NT>
__global__ __launch_bounds__(256) void spmm_stream(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ colidx,
    const double* __restrict__ val, const double* __restrict__ x, size_t ldx,
    double* __restrict__ y, size_t ldy, int m, int rpw, unsigned nchunks, unsigned xcd_group,
    const int* __restrict__ chunk_map) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned lb;
  if (chunk_map != nullptr) {
    const int cm = chunk_map[blockIdx.x];
    if (cm < 0) return;
    lb = (unsigned)cm;
  } else {
    lb = logical_block(blockIdx.x, xcd_group);
  }
  if (lb >= nchunks) return;
  const long row0 = ((long)lb * 4 + wave) * (long)rpw;
  if (row0 >= nrows) return;
  const int nr = min(rpw, (int)(nrows - row0));
  const int c0 = lane * CPL;
  const bool act = c0 < m;
  const double* __restrict__ xl = x + c0;
  double* __restrict__ yl = y + (size_t)row0 * ldy + c0;

  const int myend = rowptr[row0 + 1 + min(lane, nr - 1)];
  const int s = rowptr[row0];
  const int e = __builtin_amdgcn_readlane(myend, nr - 1);
  int r = 0;
  int pos = s;
  int next_end = __builtin_amdgcn_readfirstlane(myend);
  double acc0 = 0.0, acc1 = 0.0;

#define GCGE_FLUSH_ROWS()                                                        \
  while (r < nr && pos == next_end) {                                            \
    if (act) {                                                                   \
      double* q = yl + (size_t)r * ldy;                                          \
      if (CPL == 2) {                                                            \
        if (NT) { __builtin_nontemporal_store(acc0, q); __builtin_nontemporal_store(acc1, q + 1); } \
        else *reinterpret_cast<double2*>(q) = make_double2(acc0, acc1);          \
      } else {                                                                   \
        if (NT) __builtin_nontemporal_store(acc0, q);                            \
        else *q = acc0;                                                          \
      }                                                                          \
    }                                                                            \
    acc0 = 0.0; acc1 = 0.0;                                                      \
    ++r;                                                                         \
    next_end = __builtin_amdgcn_readlane(myend, min(r, nr - 1));                 \
  }

  GCGE_FLUSH_ROWS();  // leading empty rows
  int ncol = 0;
  double nval = 0.0;
  if (s + lane < e) {
    ncol = __builtin_nontemporal_load(colidx + s + lane);
    nval = __builtin_nontemporal_load(val + s + lane);
  }
  for (int base = s; base < e; base += 64) {
    const int mycol = ncol;
    const double myval = nval;
    const int cnt = min(64, e - base);
    ncol = 0; nval = 0.0;
    if (base + 64 + lane < e) {  // prefetch the next 64 non-zeros
      ncol = __builtin_nontemporal_load(colidx + base + 64 + lane);
      nval = __builtin_nontemporal_load(val + base + 64 + lane);
    }
    for (int k0 = 0; k0 < cnt; k0 += BATCH) {
      double xv0[BATCH], xv1[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int k = k0 + u;
        const bool on = k < cnt;
        const int c = __builtin_amdgcn_readlane(mycol, k & 63);
        const double* p = xl + (size_t)c * ldx;
        if (CPL == 2) {
          double2 t = (on && act) ? *reinterpret_cast<const double2*>(p) : make_double2(0.0, 0.0);
          xv0[u] = t.x; xv1[u] = t.y;
        } else {
          xv0[u] = (on && act) ? *p : 0.0; xv1[u] = 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int k = k0 + u;
        if (k < cnt) {
          const double a = readlane_f64(myval, k & 63);
          acc0 = fma(a, xv0[u], acc0);
          if (CPL == 2) acc1 = fma(a, xv1[u], acc1);
          ++pos;
          GCGE_FLUSH_ROWS();
        }
      }
    }
  }
#undef GCGE_FLUSH_ROWS
}

// LPR lanes cooperate on one row; lane j of the slot owns column j (j < m <= LPR).
template <int LPR>
__global__ __launch_bounds__(256) void spmm_subwave(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ colidx,
    const double* __restrict__ val, const double* __restrict__ x, size_t ldx,
    double* __restrict__ y, size_t ldy, int m) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = t / LPR;
  const int j = (int)(t % LPR);
  if (row >= (size_t)nrows || j >= m) return;
  const int s = rowptr[row], e = rowptr[row + 1];
  double acc = 0.0;
  const double* __restrict__ xj = x + j;
  int k = s;
  for (; k + 4 <= e; k += 4) {
    const int c0 = colidx[k], c1 = colidx[k + 1], c2 = colidx[k + 2], c3 = colidx[k + 3];
    const double a0 = val[k], a1 = val[k + 1], a2 = val[k + 2], a3 = val[k + 3];
    const double x0 = xj[(size_t)c0 * ldx], x1 = xj[(size_t)c1 * ldx];
    const double x2 = xj[(size_t)c2 * ldx], x3 = xj[(size_t)c3 * ldx];
    acc = fma(a0, x0, acc);
    acc = fma(a1, x1, acc);
    acc = fma(a2, x2, acc);
    acc = fma(a3, x3, acc);
  }
  for (; k < e; ++k) acc = fma(val[k], xj[(size_t)colidx[k] * ldx], acc);
  y[row * ldy + j] = acc;
}

}  // namespace gcge

using namespace gcge;

// Tunables (process-wide; set through gcge_hip_spmm_tune for experiments).
static int g_rows_per_wave = 16;
static int g_xcd_group = 1;
static int g_nt_store = 0;
static const int* g_chunk_map = nullptr;  // device array, one logical chunk id per hardware block
static unsigned g_chunk_map_len = 0;

extern "C" void gcge_hip_spmm_set_chunk_map(const int* d_map, unsigned len) {
  g_chunk_map = d_map;
  g_chunk_map_len = len;
}

extern "C" void gcge_hip_spmm_tune(int rows_per_wave, int xcd_group, int nt_store) {
  if (rows_per_wave == 4 || rows_per_wave == 8 || rows_per_wave == 16 || rows_per_wave == 32 || rows_per_wave == 64)
    g_rows_per_wave = rows_per_wave;
  if (xcd_group >= 1) g_xcd_group = xcd_group;
  g_nt_store = nt_store ? 1 : 0;
}

template <int CPL, int RPW, int NT>
static void launch_wave_row(int nrows, const int* rowptr, const int* colidx, const double* val,
                            const double* x, size_t ldx, double* y, size_t ldy, int m,
                            hipStream_t st) {
  const unsigned rows_per_block = 4u * RPW;
  const unsigned nb = (unsigned)(((size_t)nrows + rows_per_block - 1) / rows_per_block);
  const unsigned G = (unsigned)g_xcd_group;
  unsigned grid = nb;
  if (G > 1) grid = (nb + 8u * G - 1) / (8u * G) * (8u * G);
  const int* cmap = nullptr;
  if (g_chunk_map != nullptr && g_chunk_map_len >= nb) {
    cmap = g_chunk_map;
    grid = g_chunk_map_len;
  }
  hipLaunchKernelGGL((spmm_wave_row<CPL, RPW, NT>), dim3(grid), dim3(256), 0, st, nrows, rowptr,
                     colidx, val, x, ldx, y, ldy, m, nb, G, cmap);
}

template <int CPL>
static void dispatch_wave_row(int nrows, const int* rowptr, const int* colidx, const double* val,
                              const double* x, size_t ldx, double* y, size_t ldy, int m,
                              hipStream_t st) {
#define GCGE_CASE(R)                                                                            \
  case R:                                                                                       \
    if (g_nt_store) launch_wave_row<CPL, R, 1>(nrows, rowptr, colidx, val, x, ldx, y, ldy, m, st); \
    else launch_wave_row<CPL, R, 0>(nrows, rowptr, colidx, val, x, ldx, y, ldy, m, st);           \
    break;
  switch (g_rows_per_wave) {
    GCGE_CASE(4)
    GCGE_CASE(8)
    GCGE_CASE(32)
    default:
      GCGE_CASE(16)
  }
#undef GCGE_CASE
}

static int g_variant = 1;  // 0 = spmm_wave_row, 1 = spmm_stream
static int g_batch = 16;
extern "C" void gcge_hip_spmm_variant(int variant, int batch) {
  g_variant = variant;
  if (batch == 8 || batch == 16 || batch == 32) g_batch = batch;
}

template <int CPL, int BATCH, int NT>
static void launch_stream(int nrows, const int* rowptr, const int* colidx, const double* val,
                          const double* x, size_t ldx, double* y, size_t ldy, int m,
                          hipStream_t st) {
  const int rpw = g_rows_per_wave;
  const unsigned rows_per_block = 4u * (unsigned)rpw;
  const unsigned nb = (unsigned)(((size_t)nrows + rows_per_block - 1) / rows_per_block);
  const unsigned G = (unsigned)g_xcd_group;
  unsigned grid = nb;
  if (G > 1) grid = (nb + 8u * G - 1) / (8u * G) * (8u * G);
  const int* cmap = nullptr;
  if (g_chunk_map != nullptr && g_chunk_map_len >= nb) {
    cmap = g_chunk_map;
    grid = g_chunk_map_len;
  }
  hipLaunchKernelGGL((spmm_stream<CPL, BATCH, NT>), dim3(grid), dim3(256), 0, st, nrows, rowptr,
                     colidx, val, x, ldx, y, ldy, m, rpw, nb, G, cmap);
}

template <int CPL>
static void dispatch_stream(int nrows, const int* rowptr, const int* colidx, const double* val,
                            const double* x, size_t ldx, double* y, size_t ldy, int m,
                            hipStream_t st) {
#define GCGE_CASE(B)                                                                              \
  case B:                                                                                         \
    if (g_nt_store) launch_stream<CPL, B, 1>(nrows, rowptr, colidx, val, x, ldx, y, ldy, m, st);   \
    else launch_stream<CPL, B, 0>(nrows, rowptr, colidx, val, x, ldx, y, ldy, m, st);             \
    break;
  switch (g_batch) {
    GCGE_CASE(8)
    GCGE_CASE(32)
    default:
      GCGE_CASE(16)
  }
#undef GCGE_CASE
}

template <int LPR>
static void launch_subwave(int nrows, const int* rowptr, const int* colidx, const double* val,
                           const double* x, size_t ldx, double* y, size_t ldy, int m,
                           hipStream_t st) {
  const size_t threads = (size_t)nrows * LPR;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  hipLaunchKernelGGL((spmm_subwave<LPR>), dim3(grid), dim3(256), 0, st, nrows, rowptr, colidx, val,
                     x, ldx, y, ldy, m);
}

// C-ABI: see include/gcge_hip.h.  x/y point at element (0, first column).
extern "C" int gcge_hip_csr_spmm(int nrows, const int* d_rowptr, const int* d_colidx,
                                 const double* d_val, const double* d_x, long ldx, double* d_y,
                                 long ldy, int ncols, void* stream) {
  gcge_hip_apply_pending();
  if (nrows <= 0 || ncols <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int done = 0;
  while (done < ncols) {
    const double* x = d_x + done;
    double* y = d_y + done;
    int m = ncols - done;
    const bool vec2 = ((ldx | ldy) % 2 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    if (m > 32) {
      if (vec2 && m > 64) {
        m = min(m, 128);
        m &= ~1;
        if (g_variant == 1) dispatch_stream<2>(nrows, d_rowptr, d_colidx, d_val, x, (size_t)ldx, y, (size_t)ldy, m, st);
        else dispatch_wave_row<2>(nrows, d_rowptr, d_colidx, d_val, x, (size_t)ldx, y, (size_t)ldy, m, st);
      } else {
        m = min(m, 64);
        if (g_variant == 1) dispatch_stream<1>(nrows, d_rowptr, d_colidx, d_val, x, (size_t)ldx, y, (size_t)ldy, m, st);
        else dispatch_wave_row<1>(nrows, d_rowptr, d_colidx, d_val, x, (size_t)ldx, y, (size_t)ldy, m, st);
      }
    } else if (m > 16) launch_subwave<32>(nrows, d_rowptr, d_colidx, d_val, x, ldx, y, ldy, m, st);
    else if (m > 8) launch_subwave<16>(nrows, d_rowptr, d_colidx, d_val, x, ldx, y, ldy, m, st);
    else if (m > 4) launch_subwave<8>(nrows, d_rowptr, d_colidx, d_val, x, ldx, y, ldy, m, st);
    else if (m > 2) launch_subwave<4>(nrows, d_rowptr, d_colidx, d_val, x, ldx, y, ldy, m, st);
    else if (m > 1) launch_subwave<2>(nrows, d_rowptr, d_colidx, d_val, x, ldx, y, ldy, m, st);
    else launch_subwave<1>(nrows, d_rowptr, d_colidx, d_val, x, ldx, y, ldy, m, st);
    done += m;
  }
  return (int)hipGetLastError();
}
