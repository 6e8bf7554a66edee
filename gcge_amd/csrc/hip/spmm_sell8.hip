// K1 (narrow-pass path) — CSR SpMM in passes of 16 columns on the pad-8 matrix copy.
//
// Why passes: on a 3-D stencil matrix every X row is needed again one grid plane (N^2 rows) later.
// With 64 columns in flight a plane of X is 33.5 MB — more than the whole 32 MB of L2 — so each
// row is fetched three times over the fabric (measured: 2.27e8 128-B read requests per launch
// against a minimum of 0.8e8, profiles/r01_spmm_explore).  With 16 columns a plane is 8.4 MB
// (1.05 MB per XCD) and stays in L2: 1.44e8 requests for the four passes, A re-read included.
//
// Why this kernel: four passes only pay if a pass is cheap in instructions.  Lane mapping:
//   lane l -> row slot g = l >> 3 (8 rows of a "slice" per wave instruction)
//             column pair i = l & 7 (columns 2i, 2i+1 of the 16-column pass, one 16-byte load)
// so one global_load_dwordx4 fetches the 128-byte X segments of 8 different rows, every lane
// accumulates its own two outputs (no cross-lane reduction, no row-boundary bookkeeping), and
// the (col,val) octets of the 8 rows arrive with one coalesced load and are broadcast inside the
// 8-lane groups.  Rows longer than 8 non-zeros simply take more octet rounds (slice-wise maximum);
// pad entries carry value 0 and the row's own column.
//
// Same contract as spmm.hip / spmm_pad8.hip (reference app/app_ccs.c:50-139); needs 16-byte aligned
// column origins and even leading dimensions (callers fall back otherwise).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gcge_hip_internal.h"

extern "C" double* gcge_hip_partial_ws(size_t len);
extern "C" void gcge_hip_reduce_partials(const double* d_partial, int nblocks, int len, double* d_out, void* stream);

namespace gcge {

__device__ __forceinline__ double shfl_f64_s8(double v, int src) {
  int lo = __shfl(__double2loint(v), src, 64);
  int hi = __shfl(__double2hiint(v), src, 64);
  return __hiloint2double(hi, lo);
}

// SPW slices (of 8 rows) per wave; DOT: also accumulate sum_r X[r,j] Y[r,j] into dot_partial[block][16]
template <int SPW, int DOT>
__global__ __launch_bounds__(256) void spmm_sell8_kernel(
    int nrows, const int* __restrict__ orp, const int* __restrict__ pcol, const double* __restrict__ pval,
    const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int m,
    double* __restrict__ dot_partial) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 3, i = lane & 7;
  const bool act = 2 * i < m;
  const double* __restrict__ xl = x + (act ? 2 * i : 0);
  double d0 = 0.0, d1 = 0.0;
  const long slice0 = ((long)blockIdx.x * 4 + wave) * SPW;
  // (col,val) of the first octet of the NEXT slice are requested before the current slice is processed:
  // a slice then costs one memory round trip (its 8 X-row loads) instead of two dependent ones
  auto row_of = [&](int sl) { return (slice0 + sl) * 8 + g; };
  long nrow = row_of(0);
  long nrc = nrow < nrows ? nrow : (long)nrows - 1;
  int n_o0 = orp[nrc], n_o1 = (nrow < nrows) ? orp[nrc + 1] : n_o0;
  int ncolv = pcol[(long)n_o0 * 8 + i];
  double nvalv = pval[(long)n_o0 * 8 + i];
#pragma unroll 1
  for (int sl = 0; sl < SPW; ++sl) {
    if ((slice0 + sl) * 8 >= nrows) break;                 // wave-uniform
    const long row = nrow, rc = nrc;
    const bool rowok = row < nrows;
    const int o0 = n_o0, noct = n_o1 - n_o0;
    int mycol = ncolv; double myval = nvalv;
    if (sl + 1 < SPW) {                                     // prefetch the next slice's first octet
      nrow = row_of(sl + 1);
      nrc = nrow < nrows ? nrow : (long)nrows - 1;
      n_o0 = orp[nrc]; n_o1 = (nrow < nrows) ? orp[nrc + 1] : n_o0;
      ncolv = pcol[(long)n_o0 * 8 + i];
      nvalv = pval[(long)n_o0 * 8 + i];
      asm volatile("" : "+v"(ncolv), "+v"(nvalv));
    }
    // slice-wise maximum number of octets (wave-uniform): max over the 8 row slots
    int mx = noct;
    mx = max(mx, __shfl_xor(mx, 8, 64)); mx = max(mx, __shfl_xor(mx, 16, 64)); mx = max(mx, __shfl_xor(mx, 32, 64));
    mx = __builtin_amdgcn_readfirstlane(mx);
    double acc0 = 0.0, acc1 = 0.0;
    for (int o = 0; o < mx; ++o) {
      if (o > 0) {   // further octets of long rows: fetched on demand
        const bool have = o < noct;
        const long e = ((long)o0 + (have ? o : 0)) * 8 + i;
        mycol = pcol[e]; myval = pval[e];
        asm volatile("" : "+v"(mycol), "+v"(myval));
        if (!have) { mycol = (int)rc; myval = 0.0; }
      } else if (noct <= 0) { mycol = (int)rc; myval = 0.0; }
      double xv0[8], xv1[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int c = __shfl(mycol, (lane & ~7) | t, 64);
        const double2 v = *reinterpret_cast<const double2*>(xl + (size_t)c * ldx);
        xv0[t] = v.x; xv1[t] = v.y;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const double a = shfl_f64_s8(myval, (lane & ~7) | t);
        acc0 = fma(a, xv0[t], acc0); acc1 = fma(a, xv1[t], acc1);
      }
    }
    if (rowok && act) {
      double* q = y + (size_t)row * ldy + 2 * i;
      __builtin_nontemporal_store(acc0, q);
      __builtin_nontemporal_store(acc1, q + 1);
    }
    if (DOT) {
      const double2 o = *reinterpret_cast<const double2*>(xl + (size_t)rc * ldx);
      if (rowok && act) { d0 = fma(acc0, o.x, d0); d1 = fma(acc1, o.y, d1); }
    }
  }
  if (DOT) {
    // sum the 8 row slots of the wave, then the 4 waves of the block
    d0 += shfl_f64_s8(d0, lane ^ 8);  d1 += shfl_f64_s8(d1, lane ^ 8);
    d0 += shfl_f64_s8(d0, lane ^ 16); d1 += shfl_f64_s8(d1, lane ^ 16);
    d0 += shfl_f64_s8(d0, lane ^ 32); d1 += shfl_f64_s8(d1, lane ^ 32);
    __shared__ double sred[4][16];
    if (lane < 8) { sred[wave][2 * lane] = d0; sred[wave][2 * lane + 1] = d1; }
    __syncthreads();
    if (threadIdx.x < 16 && (int)threadIdx.x < m)
      dot_partial[(long)blockIdx.x * m + threadIdx.x] =
          (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
  }
}

}  // namespace gcge

using namespace gcge;

static int g_s8_spw = 2;
extern "C" void gcge_hip_spmm_sell8_tune(int slices_per_wave) {
  if (slices_per_wave == 1 || slices_per_wave == 2 || slices_per_wave == 4 || slices_per_wave == 8) g_s8_spw = slices_per_wave;
}

template <int DOT>
static void s8_launch(int nrows, const int* orp, const int* pcol, const double* pval, const double* x, size_t ldx,
                      double* y, size_t ldy, int m, double* partial, long* nblocks_out, hipStream_t st) {
  const long rows_per_block = 32L * g_s8_spw;
  const long nb = ((long)nrows + rows_per_block - 1) / rows_per_block;
  if (nblocks_out) *nblocks_out = nb;
  switch (g_s8_spw) {
#define GCGE_S8(S) case S: hipLaunchKernelGGL((spmm_sell8_kernel<S, DOT>), dim3((unsigned)nb), dim3(256), 0, st, nrows, orp, \
                                                pcol, pval, x, ldx, y, ldy, m, partial); break;
    GCGE_S8(1) GCGE_S8(4) GCGE_S8(8)
    default: GCGE_S8(2)
#undef GCGE_S8
  }
}

// Y[:,0:ncols) = A X[:,0:ncols) in passes of 16 columns.  -1: alignment contract not met.
extern "C" int gcge_hip_sell8_spmm(int nrows, const int* d_orp, const int* d_pcol, const double* d_pval,
                                   const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream) {
  if (nrows <= 0 || ncols <= 0) return 0;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15)) return -1;
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0; c0 < ncols; c0 += 16) {
    const int m = (ncols - c0 < 16) ? ncols - c0 : 16;
    s8_launch<0>(nrows, d_orp, d_pcol, d_pval, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, nullptr, nullptr, st);
  }
  return (int)hipGetLastError();
}

// same, plus d_dots[j] = sum_r X[r,j] Y[r,j]
extern "C" int gcge_hip_sell8_spmm_dot(int nrows, const int* d_orp, const int* d_pcol, const double* d_pval,
                                       const double* d_x, long ldx, double* d_y, long ldy, int ncols,
                                       double* d_dots, void* stream) {
  if (nrows <= 0 || ncols <= 0) return 0;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const long rows_per_block = 32L * g_s8_spw;
  const long nb = ((long)nrows + rows_per_block - 1) / rows_per_block;
  for (int c0 = 0; c0 < ncols; c0 += 16) {
    const int m = (ncols - c0 < 16) ? ncols - c0 : 16;
    double* part = gcge_hip_partial_ws((size_t)nb * 16);
    s8_launch<1>(nrows, d_orp, d_pcol, d_pval, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, part, nullptr, st);
    gcge_hip_reduce_partials(part, (int)nb, m, d_dots + c0, st);
  }
  return (int)hipGetLastError();
}
