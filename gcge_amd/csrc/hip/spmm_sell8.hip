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

typedef double v2d __attribute__((ext_vector_type(2)));

// A wave works on "wave chunks" of SPW slices (8 SPW consecutive rows, SPW <= 7), grid-stride.
// Memory pipeline of a chunk: one load brings the 8 SPW + 1 row pointers (lane l holds orp[row0 + l]);
// the (col,val) octets of slice s+1 are requested before the 8 X-row loads of slice s, so a slice
// costs one memory round trip.  Nothing next to a load is predicated (hipcc turns `cond ? load : c`
// into a branch and then waits with vmcnt(0) after every load): indices are clamped and the loaded
// (col,val) are blended with integer masks instead.
// DOT: also accumulate sum_r X[r,j] Y[r,j] into dot_partial[block][16] (16 values per block).
template <int SPW, int DOT>
__global__ __launch_bounds__(256) void spmm_sell8_kernel(
    int nrows, const int* __restrict__ orp, const int* __restrict__ pcol, const double* __restrict__ pval,
    const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int m, long nwchunks,
    double* __restrict__ dot_partial) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 3, i = lane & 7;
  const int grp = lane & ~7;
  const bool act = 2 * i < m;
  const double* __restrict__ xl = x + (act ? 2 * i : 0);
  const int last_oct = max(orp[nrows] - 1, 0);
  double d0 = 0.0, d1 = 0.0;
  for (long wc = (long)blockIdx.x * 4 + wave; wc < nwchunks; wc += (long)gridDim.x * 4) {
    const long row0 = wc * (8 * SPW);
    const int orpv = orp[min(row0 + lane, (long)nrows)];
    // operands of slice 0
    int o0 = __shfl(orpv, g, 64), o1 = __shfl(orpv, g + 1, 64);
    long e0 = (long)min(o0, last_oct) * 8 + i;
    int ncolv = pcol[e0];
    double nvalv = pval[e0];
#pragma unroll
    for (int sl = 0; sl < SPW; ++sl) {
      const long row = row0 + 8 * sl + g;
      const long rc = min(row, (long)nrows - 1);
      const int noct = o1 - o0, ob = o0;
      int mycol = ncolv;
      double myval = nvalv;
      if (sl + 1 < SPW) {   // compile-time: request the next slice's first octet now
        o0 = __shfl(orpv, 8 * (sl + 1) + g, 64); o1 = __shfl(orpv, 8 * (sl + 1) + g + 1, 64);
        e0 = (long)min(o0, last_oct) * 8 + i;
        ncolv = pcol[e0];
        nvalv = pval[e0];
      }
      // slice-wise maximum number of octets (wave-uniform): max over the 8 row slots
      int mx = noct;
      mx = max(mx, __shfl_xor(mx, 8, 64)); mx = max(mx, __shfl_xor(mx, 16, 64)); mx = max(mx, __shfl_xor(mx, 32, 64));
      mx = __builtin_amdgcn_readfirstlane(mx);
      double acc0 = 0.0, acc1 = 0.0;
      for (int o = 0; o < mx; ++o) {
        const bool have = o < noct;
        if (o > 0) {   // further octets of long rows: fetched on demand (wave-uniform branch)
          const long e = ((long)ob + (have ? o : 0)) * 8 + i;
          mycol = pcol[min(e, (long)last_oct * 8 + 7)];
          myval = pval[min(e, (long)last_oct * 8 + 7)];
        }
        // rows without this octet (or beyond nrows): own column, value 0 — blended with masks, not selects
        const int msk = have ? -1 : 0;
        const int c_use = (int)rc + ((mycol - (int)rc) & msk);
        const long long vb = __double_as_longlong(myval) & (long long)msk;
        const double v_use = __longlong_as_double(vb);
        double xv0[8], xv1[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int c = __shfl(c_use, grp | t, 64);
          const v2d v = *reinterpret_cast<const v2d*>(xl + (size_t)c * ldx);
          xv0[t] = v.x; xv1[t] = v.y;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const double a = shfl_f64_s8(v_use, grp | t);
          acc0 = fma(a, xv0[t], acc0); acc1 = fma(a, xv1[t], acc1);
        }
      }
      if (row < nrows && act) {
        v2d out = {acc0, acc1};
        __builtin_nontemporal_store(out, reinterpret_cast<v2d*>(y + (size_t)row * ldy + 2 * i));
      }
      if (DOT) {
        const v2d ow = *reinterpret_cast<const v2d*>(xl + (size_t)rc * ldx);
        const double w = (row < nrows && act) ? 1.0 : 0.0;
        d0 = fma(acc0 * w, ow.x, d0); d1 = fma(acc1 * w, ow.y, d1);
      }
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

static int g_s8_spw = 4, g_s8_dotgrid = 4096;
extern "C" void gcge_hip_spmm_sell8_tune(int slices_per_wave) {
  if (slices_per_wave == 1 || slices_per_wave == 2 || slices_per_wave == 4 || slices_per_wave == 7) g_s8_spw = slices_per_wave;
}

// grid = 0: one wave chunk per wave
template <int DOT>
static long s8_launch(int nrows, const int* orp, const int* pcol, const double* pval, const double* x, size_t ldx,
                      double* y, size_t ldy, int m, double* partial, long grid, hipStream_t st) {
  const long nwc = ((long)nrows + 8L * g_s8_spw - 1) / (8L * g_s8_spw);
  long nb = (nwc + 3) / 4;
  if (grid > 0 && nb > grid) nb = grid;
  switch (g_s8_spw) {
#define GCGE_S8(S) case S: hipLaunchKernelGGL((spmm_sell8_kernel<S, DOT>), dim3((unsigned)nb), dim3(256), 0, st, nrows, orp, \
                                                pcol, pval, x, ldx, y, ldy, m, nwc, partial); break;
    GCGE_S8(1) GCGE_S8(2) GCGE_S8(7)
    default: GCGE_S8(4)
#undef GCGE_S8
  }
  return nb;
}

// Y[:,0:ncols) = A X[:,0:ncols) in passes of 16 columns.  -1: alignment contract not met.
extern "C" int gcge_hip_sell8_spmm(int nrows, const int* d_orp, const int* d_pcol, const double* d_pval,
                                   const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream) {
  if (nrows <= 0 || ncols <= 0) return 0;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15)) return -1;
  hipStream_t st = (hipStream_t)stream;
  for (int c0 = 0; c0 < ncols; c0 += 16) {
    const int m = (ncols - c0 < 16) ? ncols - c0 : 16;
    s8_launch<0>(nrows, d_orp, d_pcol, d_pval, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, nullptr, 0, st);
  }
  return (int)hipGetLastError();
}

// same, plus d_dots[j] = sum_r X[r,j] Y[r,j]  (a bounded grid walks the wave chunks, so few partials)
extern "C" int gcge_hip_sell8_spmm_dot(int nrows, const int* d_orp, const int* d_pcol, const double* d_pval,
                                       const double* d_x, long ldx, double* d_y, long ldy, int ncols,
                                       double* d_dots, void* stream) {
  if (nrows <= 0 || ncols <= 0) return 0;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int npass = (ncols + 15) / 16;
  double* part = gcge_hip_partial_ws((size_t)g_s8_dotgrid * 16 * npass);
  for (int c0 = 0, ps = 0; c0 < ncols; c0 += 16, ++ps) {
    const int m = (ncols - c0 < 16) ? ncols - c0 : 16;
    double* pp = part + (size_t)ps * g_s8_dotgrid * 16;
    const long nb = s8_launch<1>(nrows, d_orp, d_pcol, d_pval, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, pp,
                                 g_s8_dotgrid, st);
    gcge_hip_reduce_partials(pp, (int)nb, m, d_dots + c0, st);
  }
  return (int)hipGetLastError();
}
