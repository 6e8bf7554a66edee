// K3 — panel update  Y[:,0:m) = X[:,0:k) C + Y diag(beta)  on FP64 MFMA.
//
// Replaces MultiVecLinearComb's dgemm('N','N') (reference app/app_lapack.c:463-534):
// Ritz vectors (ops_eig_sol_gcg.c:183, the largest GEMM of the loop), the P update
// (:433-437) and the Gram–Schmidt updates x1 -= x0 coef (ops_orth.c:253,347).
//
// Rows are the long dimension: a block owns 64 rows (16 per wave) and ALL m <= 128
// output columns, so X is read once and Y written once.  The reduction index is the
// column of X, which is the contiguous direction of a row-major block: the 64 x KT
// X tile is staged through LDS with full-row coalesced loads (an A fragment read
// straight from global memory would touch 16 rows x 32 B per instruction), C (k x m,
// row-major) is staged KT rows at a time.
//
//   v_mfma_f64_16x16x4_f64:  D(16x16) += A(16x4) B(4x16)
//     A: lane l holds X[r0 + 16 w + (l & 15)][k0 + (l >> 4)]   (LDS, row stride KT+2: conflict-free)
//     B: lane l holds C[k0 + (l >> 4)][16 t + (l & 15)]        (LDS, row stride = 16 mod 32)
//     D: lane l, reg u holds Y[r0 + 16 w + 4 u + (l >> 4)][16 t + (l & 15)]
//
// In-place use (x == y with disjoint column ranges, ops_orth.c:70,90,253) is safe:
// a block reads and writes only its own 64 rows and never the same columns.
// Roofline: 2 n k m flops (FP64 MFMA) vs 8 n (k + m [+ m]) bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gcge_hip_internal.h"

namespace gcge {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int LC_KT = 32;          // k-tile
constexpr int LC_XS = LC_KT + 2;   // LDS row stride of the X tile (doubles)

template <int NT>  // NT 16-column output fragments per wave: m <= 16 NT
__global__ __launch_bounds__(256) void lincomb_kernel(long nrows, const double* x, long ldx, int k,
    const double* __restrict__ c, int m, const double* __restrict__ beta, double* y, long ldy, int cs) {
  extern __shared__ __align__(16) double lds[];
  double* xs = lds;                 // [64][LC_XS]
  double* cst = lds + 64 * LC_XS;   // [LC_KT][cs]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const long r0 = (long)blockIdx.x * 64;

  v4d acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};

  // register double-buffering: the next k-tile is fetched from global memory while the MFMAs of the
  // current one run; it is written to LDS after the barrier that ends the current tile
  constexpr int XE = 64 * LC_KT / 256;            // X elements per thread and tile (8)
  constexpr int CE = LC_KT * 16 * NT / 256;       // C elements per thread and tile (2 NT)
  double xr[XE];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < XE; ++q) {
      const int e = threadIdx.x + 256 * q, row = e / LC_KT, col = e % LC_KT;
      const long gr = min(r0 + row, nrows - 1);
      const int gc = min(k0 + col, k - 1);
      double v = x[gr * ldx + gc];
      asm volatile("" : "+v"(v));   // unconditional load (see gram_mfma.hip)
      xr[q] = (r0 + row < nrows && k0 + col < k) ? v : 0.0;
    }
  };
  // the coefficient tile is small and L2-resident: it goes straight to LDS (no register stage)
  auto stash = [&](int k0) {
#pragma unroll
    for (int q = 0; q < XE; ++q) { const int e = threadIdx.x + 256 * q; xs[(e / LC_KT) * LC_XS + e % LC_KT] = xr[q]; }
#pragma unroll
    for (int q = 0; q < CE; ++q) {
      const int e = threadIdx.x + 256 * q, row = e / (16 * NT), col = e % (16 * NT);
      const int gr = min(k0 + row, k - 1), gc = min(col, m - 1);
      double v = c[(long)gr * m + gc];
      asm volatile("" : "+v"(v));
      cst[row * cs + col] = (k0 + row < k && col < m) ? v : 0.0;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < k; k0 += LC_KT) {
    stash(k0);
    __syncthreads();
    if (k0 + LC_KT < k) fetch(k0 + LC_KT);
#pragma unroll
    for (int s = 0; s < LC_KT; s += 4) {
      const double a = xs[(16 * wave + li) * LC_XS + s + kk];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double b = cst[(s + kk) * cs + 16 * t + li];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // epilogue: Y = acc + beta_j * Y
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = 16 * t + li;
    if (col >= m) continue;
    const double bj = (beta != nullptr) ? beta[col] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long row = r0 + 16 * wave + 4 * u + kk;
      if (row < nrows) {
        double* py = y + row * ldy + col;
        *py = (beta != nullptr) ? fma(bj, *py, acc[t][u]) : acc[t][u];
      }
    }
  }
}

}  // namespace gcge

using namespace gcge;

template <int NT>
static void lc_launch(int nrows, const double* x, long ldx, int k, const double* c, int m,
                      const double* beta, double* y, long ldy, hipStream_t st) {
  const int cs = (16 * NT + 31) / 32 * 32 + 16;  // row stride of the C tile: 16 mod 32 doubles
  const size_t shmem = (size_t)(64 * LC_XS + LC_KT * cs) * sizeof(double);
  const unsigned grid = (unsigned)(((long)nrows + 63) / 64);
  hipLaunchKernelGGL((lincomb_kernel<NT>), dim3(grid), dim3(256), shmem, st, (long)nrows, x, ldx, k, c, m, beta,
                     y, ldy, cs);
}

// d_c: row-major k x m coefficient block on the device; d_beta: m scale factors or NULL
extern "C" int gcge_hip_lincomb(int nrows, const double* d_x, long ldx, int k, const double* d_c, int m,
                                const double* d_beta, double* d_y, long ldy, void* stream) {
  if (nrows <= 0 || m <= 0 || k <= 0) return 0;
  if (m > 128) return -2;  // callers split wider panels
  hipStream_t st = (hipStream_t)stream;
  if (m <= 16) lc_launch<1>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  else if (m <= 32) lc_launch<2>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  else if (m <= 64) lc_launch<4>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  else lc_launch<8>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  return (int)hipGetLastError();
}
