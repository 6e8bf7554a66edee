// K4 / K6 and helpers — streaming kernels on row-major blocks of vectors.
//
//   gcge_hip_axpby        Y[:,0:m) = alpha X[:,0:m) + beta Y      reference: app/app_lapack.c:334-395
//   gcge_hip_colscale     Y[:,j)  *= s[j]                          (x == NULL branch of :463-534)
//   gcge_hip_coldots      d[j] = sum_r X[r,j] Y[r,j]               ('D' branch of DenseMatQtAP :70-118)
//   gcge_hip_fill_uniform counter-based U[0,1) fill (mode 1 of MultiVecSetRandomValue)
//   transposing copies    row-major device block <-> column-major staging
//
// Roofline: HBM.  Bytes per element: axpby 8*(2 or 3), dots 16, scale 16.
// One lane per COLUMN of a row segment, rows looped: a wave touches contiguous
// 8*m-byte row segments; 16-byte lanes are used when the column origin and the
// leading dimensions are even.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gcge_hip_internal.h"

namespace gcge {

// generic: thread = (row, column) with the column index fastest
__global__ __launch_bounds__(256) void axpby_kernel(long nrows, double alpha,
    const double* __restrict__ x, long ldx, double beta, double* y, long ldy, int m, int mode) {
  // mode: 0 y = a x + b y ; 1 y = a x (no read of y) ; 2 y = b y ; 3 y = 0
  const long total = nrows * (long)m;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; idx < total; idx += stride) {
    const long r = idx / m;
    const int j = (int)(idx - r * m);
    double* py = y + r * ldy + j;
    if (mode == 0) *py = alpha * x[r * ldx + j] + beta * (*py);
    else if (mode == 1) *py = alpha * x[r * ldx + j];
    else if (mode == 2) *py = beta * (*py);
    else *py = 0.0;
  }
}

// 16-byte lanes: m even, x/y origins 16-byte aligned, ldx/ldy even
__global__ __launch_bounds__(256) void axpby2_kernel(long nrows, double alpha,
    const double* __restrict__ x, long ldx, double beta, double* y, long ldy, int m2, int mode) {
  const long total = nrows * (long)m2;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; idx < total; idx += stride) {
    const long r = idx / m2;
    const int j = 2 * (int)(idx - r * m2);
    double2* py = reinterpret_cast<double2*>(y + r * ldy + j);
    double2 v;
    if (mode == 0) {
      const double2 a = *reinterpret_cast<const double2*>(x + r * ldx + j);
      const double2 b = *py;
      v.x = alpha * a.x + beta * b.x; v.y = alpha * a.y + beta * b.y;
    } else if (mode == 1) {
      const double2 a = *reinterpret_cast<const double2*>(x + r * ldx + j);
      v.x = alpha * a.x; v.y = alpha * a.y;
    } else if (mode == 2) {
      const double2 b = *py;
      v.x = beta * b.x; v.y = beta * b.y;
    } else { v.x = 0.0; v.y = 0.0; }
    *py = v;
  }
}


// The same in the form the streaming kernels of block_pcg.hip use: `tpr` threads walk a row (two columns each), a block owns
// a contiguous slab of rows, four rows are in flight per thread before the first is consumed, and the mode is a template
// parameter — no index division and no mode branch per element (the grid-stride form above moves a 64-column copy at
// 4.5 TB/s).  m2 <= 256 column pairs.
typedef double v2d_ax __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void axpby2_rows_kernel(long nrows, double alpha, const double* __restrict__ x, long ldx,
    double beta, double* y, long ldy, int m2, int tpr) {
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, rpb = 256 / tpr;
  if (tx >= m2) return;
  const int j = 2 * tx;
  const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + rpb - 1) / rpb * rpb;
  const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
  long row = (long)blockIdx.x * slab + ty;
  auto one = [&](long r, v2d_ax a, v2d_ax b) {
    v2d_ax v;
    if (MODE == 0) { v.x = alpha * a.x + beta * b.x; v.y = alpha * a.y + beta * b.y; }
    else if (MODE == 1) { v.x = alpha * a.x; v.y = alpha * a.y; }
    else { v.x = beta * b.x; v.y = beta * b.y; }
    __builtin_nontemporal_store(v, reinterpret_cast<v2d_ax*>(y + r * ldy + j));
  };
  for (; row + 3L * rpb < rend; row += 4L * rpb) {
    v2d_ax a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long r = row + (long)u * rpb;
      if (MODE != 2) a[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d_ax*>(x + r * ldx + j)); else a[u] = v2d_ax{0.0, 0.0};
      if (MODE != 1) b[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d_ax*>(y + r * ldy + j)); else b[u] = v2d_ax{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(row + (long)u * rpb, a[u], b[u]);
  }
  for (; row < rend; row += rpb) {
    v2d_ax a = v2d_ax{0.0, 0.0}, b = v2d_ax{0.0, 0.0};
    if (MODE != 2) a = *reinterpret_cast<const v2d_ax*>(x + row * ldx + j);
    if (MODE != 1) b = *reinterpret_cast<const v2d_ax*>(y + row * ldy + j);
    one(row, a, b);
  }
}

__global__ __launch_bounds__(256) void colscale_kernel(long nrows, double* y, long ldy, int m,
    const double* __restrict__ s) {
  const long total = nrows * (long)m;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; idx < total; idx += stride) {
    const long r = idx / m;
    const int j = (int)(idx - r * m);
    y[r * ldy + j] *= s[j];
  }
}

// partial[b*m + j] = sum over the block's rows of x[r,j]*y[r,j];  256 threads = 4 row lanes x 64 cols
__global__ __launch_bounds__(256) void coldots_partial(long nrows, const double* __restrict__ x, long ldx,
    const double* __restrict__ y, long ldy, int m, double* __restrict__ partial, long rows_per_block) {
  __shared__ double red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(nrows, r0 + rows_per_block);
  for (int c0 = 0; c0 < m; c0 += 64) {
    const int j = c0 + tx;
    double s0 = 0.0, s1 = 0.0;
    if (j < m) {
      long r = r0 + ty;
      for (; r + 4 < r1; r += 8) {
        s0 = fma(x[r * ldx + j], y[r * ldy + j], s0);
        s1 = fma(x[(r + 4) * ldx + j], y[(r + 4) * ldy + j], s1);
      }
      for (; r < r1; r += 4) s0 = fma(x[r * ldx + j], y[r * ldy + j], s0);
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && j < m)
      partial[(long)blockIdx.x * m + j] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    __syncthreads();
  }
}
// the same sweep with two results: partial[b*2m + j] = sum x[r,j]*y[r,j], partial[b*2m + m + j] = sum y[r,j]^2 — the pair the
// fused CG asks for after a product it could not fuse the sums into (p.w and w.w: one read of p and w instead of p, w, w)
__global__ __launch_bounds__(256) void coldots2_partial(long nrows, const double* __restrict__ x, long ldx,
    const double* __restrict__ y, long ldy, int m, double* __restrict__ partial, long rows_per_block) {
  __shared__ double red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(nrows, r0 + rows_per_block);
  for (int c0 = 0; c0 < m; c0 += 64) {
    const int j = c0 + tx;
    double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
    if (j < m) {
      long r = r0 + ty;
      for (; r + 4 < r1; r += 8) {
        const double ya = y[r * ldy + j], yb = y[(r + 4) * ldy + j];
        s0 = fma(x[r * ldx + j], ya, s0); q0 = fma(ya, ya, q0);
        s1 = fma(x[(r + 4) * ldx + j], yb, s1); q1 = fma(yb, yb, q1);
      }
      for (; r < r1; r += 4) { const double ya = y[r * ldy + j]; s0 = fma(x[r * ldx + j], ya, s0); q0 = fma(ya, ya, q0); }
    }
    red[0][ty][tx] = s0 + s1; red[1][ty][tx] = q0 + q1;
    __syncthreads();
    if (ty == 0 && j < m) {
      partial[(long)blockIdx.x * 2 * m + j] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
      partial[(long)blockIdx.x * 2 * m + m + j] = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
    }
    __syncthreads();
  }
}
// partial[b*m + j] = sum over the rows of block b of (w[r,j] - lambda_j x[r,j])^2 — the squared residual norms of Ritz pairs after a
// product w = A x that could not carry them (ops_eig_sol_gcg.c:195-315 forms lambda B x, subtracts and takes norms: 9 block
// streams for B = NULL; here 2)
__global__ __launch_bounds__(256) void resid_sq_partial(long nrows, const double* __restrict__ w, long ldw,
    const double* __restrict__ x, long ldx, int m, const double* __restrict__ lambda, double* __restrict__ partial, long rows_per_block) {
  __shared__ double red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(nrows, r0 + rows_per_block);
  for (int c0 = 0; c0 < m; c0 += 64) {
    const int j = c0 + tx;
    double s0 = 0.0, s1 = 0.0;
    if (j < m) {
      const double lam = lambda[j];
      long r = r0 + ty;
      for (; r + 4 < r1; r += 8) {
        const double da = fma(-lam, x[r * ldx + j], w[r * ldw + j]), db = fma(-lam, x[(r + 4) * ldx + j], w[(r + 4) * ldw + j]);
        s0 = fma(da, da, s0); s1 = fma(db, db, s1);
      }
      for (; r < r1; r += 4) { const double da = fma(-lam, x[r * ldx + j], w[r * ldw + j]); s0 = fma(da, da, s0); }
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && j < m) partial[(long)blockIdx.x * m + j] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    __syncthreads();
  }
}
// out[j] = sum_b partial[b*len + j]; 16 row groups x 64 columns per block, fixed summation
// tree => bitwise reproducible, and no thread walks more than nblocks/16 entries
__global__ __launch_bounds__(1024) void reduce_partials(const double* __restrict__ partial, int nblocks, int len,
    double* __restrict__ out) {
  __shared__ double red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  double s0 = 0.0, s1 = 0.0;
  if (j < len) {
    int b = ty;
    for (; b + 16 < nblocks; b += 32) { s0 += partial[(long)b * len + j]; s1 += partial[(long)(b + 16) * len + j]; }
    for (; b < nblocks; b += 16) s0 += partial[(long)b * len + j];
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  for (int h = 8; h > 0; h >>= 1) {
    if (ty < h) red[ty][tx] += red[ty + h][tx];
    __syncthreads();
  }
  if (ty == 0 && j < len) out[j] = red[0][tx];
}

__device__ __forceinline__ double u01(unsigned long long seed, unsigned long long index) {
  unsigned long long z = seed + (index + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
// element (global row g, column c) <- u01(seed, c*nglobal + g): independent of the row partition
__global__ __launch_bounds__(256) void fill_uniform_kernel(long nrows, long row_begin, long nglobal,
    double* y, long ldy, int c0, int m, unsigned long long seed) {
  const long total = nrows * (long)m;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; idx < total; idx += stride) {
    const long r = idx / m;
    const int j = (int)(idx - r * m);
    y[r * ldy + c0 + j] = u01(seed, (unsigned long long)(c0 + j) * (unsigned long long)nglobal +
                                        (unsigned long long)(row_begin + r));
  }
}

// dst (row-major, ld ldd) [r, j] = src (column-major, ld lds) [r + lds*j], tile-transposed through LDS
__global__ __launch_bounds__(256) void colmajor_to_rowmajor(long nrows, int m, const double* __restrict__ src,
    long lds, double* __restrict__ dst, long ldd) {
  __shared__ double tile[32][33];
  const long r0 = (long)blockIdx.x * 32;
  const int j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int jj = ty; jj < 32; jj += 8)
    if (r0 + tx < nrows && j0 + jj < m) tile[jj][tx] = src[(long)(j0 + jj) * lds + r0 + tx];
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8)
    if (r0 + rr < nrows && j0 + tx < m) dst[(r0 + rr) * ldd + j0 + tx] = tile[tx][rr];
}
__global__ __launch_bounds__(256) void rowmajor_to_colmajor(long nrows, int m, const double* __restrict__ src,
    long lds, double* __restrict__ dst, long ldd) {
  __shared__ double tile[32][33];
  const long r0 = (long)blockIdx.x * 32;
  const int j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8)
    if (r0 + rr < nrows && j0 + tx < m) tile[rr][tx] = src[(r0 + rr) * lds + j0 + tx];
  __syncthreads();
  for (int jj = ty; jj < 32; jj += 8)
    if (r0 + tx < nrows && j0 + jj < m) dst[(long)(j0 + jj) * ldd + r0 + tx] = tile[tx][jj];
}

}  // namespace gcge

using namespace gcge;

static inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  const long cap = 256L * 32;  // 32 blocks per CU, grid-stride the rest
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" int gcge_hip_axpby(int nrows, double alpha, const double* d_x, long ldx, double beta,
                              double* d_y, long ldy, int m, void* stream) {
  gcge_hip_apply_pending();
  if (nrows <= 0 || m <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int mode;
  if (d_x == nullptr || alpha == 0.0) mode = (beta == 0.0) ? 3 : 2;
  else mode = (beta == 0.0) ? 1 : 0;
  if (mode == 2 && beta == 1.0) return 0;
  if (mode == 3 && ldy == m) {  // contiguous block: plain memset
    return (int)hipMemsetAsync(d_y, 0, (size_t)nrows * m * sizeof(double), st);
  }
  const bool vec2 = (m % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)d_y & 15) == 0) &&
                    (mode >= 2 || ((ldx % 2 == 0) && (((uintptr_t)d_x & 15) == 0)));
  // A wide range that starts on an odd column of both blocks (the solver's X / P / W ranges move with the number of locked pairs) or
  // has an odd width: the odd columns at its ends go through the element-wise kernel, the even-aligned middle through the 16-byte
  // lanes (the element-wise kernel moves a 128-column copy at 3.6 TB/s).  Same operation per element.
  if (!vec2 && mode <= 2 && m >= 8 && ldy % 2 == 0 && (mode == 2 || ldx % 2 == 0) && (long)nrows >= 1024 &&
      (mode == 2 || (((uintptr_t)d_x & 15) == ((uintptr_t)d_y & 15)))) {
    const int head = (((uintptr_t)d_y & 15) == 8) ? 1 : 0;         // one column up to the next 16-byte boundary
    const int tail = (m - head) % 2;
    const int mid = m - head - tail;
    if (mid >= 2 && (head || tail)) {
      const double* xm = d_x ? d_x + head : nullptr;
      if (head) hipLaunchKernelGGL(axpby_kernel, dim3(grid_for((long)nrows)), dim3(256), 0, st, (long)nrows, alpha, d_x, ldx, beta, d_y, ldy, 1, mode);
      if (tail) hipLaunchKernelGGL(axpby_kernel, dim3(grid_for((long)nrows)), dim3(256), 0, st, (long)nrows, alpha, d_x ? d_x + head + mid : nullptr, ldx, beta,
                                   d_y + head + mid, ldy, 1, mode);
      return gcge_hip_axpby(nrows, alpha, xm, ldx, beta, d_y + head, ldy, mid, stream);
    }
  }
  // x == y with different column ranges (column copies inside one block) is fine for the row form as well: a thread reads
  // and writes only its own (row, column pair)
  if (vec2 && mode <= 2 && m / 2 <= 256 && (long)nrows >= 1024) {
    const int m2 = m / 2;
    int tpr = 1; while (tpr < m2) tpr *= 2;
    const int rpb = 256 / tpr;
    long g = ((long)nrows + (long)rpb * 8 - 1) / ((long)rpb * 8); if (g > 8192) g = 8192; if (g < 1) g = 1;
    if (mode == 0) hipLaunchKernelGGL(axpby2_rows_kernel<0>, dim3((unsigned)g), dim3(256), 0, st, (long)nrows, alpha, d_x, ldx, beta, d_y, ldy, m2, tpr);
    else if (mode == 1) hipLaunchKernelGGL(axpby2_rows_kernel<1>, dim3((unsigned)g), dim3(256), 0, st, (long)nrows, alpha, d_x, ldx, beta, d_y, ldy, m2, tpr);
    else hipLaunchKernelGGL(axpby2_rows_kernel<2>, dim3((unsigned)g), dim3(256), 0, st, (long)nrows, alpha, d_x, ldx, beta, d_y, ldy, m2, tpr);
  } else if (vec2)
    hipLaunchKernelGGL(axpby2_kernel, dim3(grid_for((long)nrows * (m / 2))), dim3(256), 0, st, (long)nrows,
                       alpha, d_x, ldx, beta, d_y, ldy, m / 2, mode);
  else
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for((long)nrows * m)), dim3(256), 0, st, (long)nrows, alpha,
                       d_x, ldx, beta, d_y, ldy, m, mode);
  return (int)hipGetLastError();
}

extern "C" int gcge_hip_colscale(int nrows, double* d_y, long ldy, int m, const double* d_s, void* stream) {
  if (nrows <= 0 || m <= 0) return 0;
  hipLaunchKernelGGL(colscale_kernel, dim3(grid_for((long)nrows * m)), dim3(256), 0, (hipStream_t)stream,
                     (long)nrows, d_y, ldy, m, d_s);
  return (int)hipGetLastError();
}

// workspace for partial sums, grown on demand (owned by this translation unit)
static double* g_partial = nullptr;
static size_t g_partial_len = 0;
extern "C" double* gcge_hip_partial_ws(size_t len) {
  if (len > g_partial_len) {
    if (g_partial) GCGE_HIP_CHECK(hipFree(g_partial));
    g_partial_len = len + len / 4 + 4096;
    GCGE_HIP_CHECK(hipMalloc(&g_partial, g_partial_len * sizeof(double)));
  }
  return g_partial;
}

// ---- the three single-column operations of a column-wise Gram-Schmidt (reference OrthSelf, src/ops_orth.c:45-118) ----------
// On a row-major block a "column" is a strided walk, but the k columns it is combined with sit in the SAME rows right next
// to it: every kernel here walks row segments of k contiguous doubles (one lane per column, `tpc` lanes per row, several
// rows per wave instruction) and takes the single column's element from the same row.
// (a) partial[b*k + i] = sum over the block's rows of X[r, i] * y[r]      (panel . column: the k x 1 Gram of MGS step k)
template <int UNR>
__global__ __launch_bounds__(256) void panel_dot1_partial(long nrows, const double* __restrict__ x, long ldx,
    const double* __restrict__ y, long ldy, int k, double* __restrict__ partial, long rows_per_block, int tpc) {
  __shared__ double red[256];
  const int tx = threadIdx.x % tpc, ty = threadIdx.x / tpc, rpi = 256 / tpc;   // rpi rows per block iteration
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
  for (int c0 = 0; c0 < k; c0 += tpc) {
    const int j = c0 + tx;
    const int jj = j < k ? j : 0;
    double s[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) s[u] = 0.0;
    long r = r0 + ty;
    for (; r + (long)(UNR - 1) * rpi < r1; r += (long)UNR * rpi) {
      double xv[UNR], yv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) { xv[u] = x[(r + (long)u * rpi) * ldx + jj]; yv[u] = y[(r + (long)u * rpi) * ldy]; }
#pragma unroll
      for (int u = 0; u < UNR; ++u) s[u] = fma(xv[u], yv[u], s[u]);
    }
    for (; r < r1; r += rpi) s[0] = fma(x[r * ldx + jj], y[r * ldy], s[0]);
    double t = 0.0;
#pragma unroll
    for (int u = 0; u < UNR; ++u) t += s[u];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int h = rpi / 2; h > 0; h >>= 1) {
      if (ty < h) red[threadIdx.x] += red[threadIdx.x + h * tpc];
      __syncthreads();
    }
    if (ty == 0 && j < k) partial[(long)blockIdx.x * k + j] = red[tx];
    __syncthreads();
  }
}
// (b) Y[r, j] = x[r] * c[j] + beta[j] * Y[r, j]  for j < m  (rank-1 update; beta == NULL: 1)
template <int UNR>
__global__ __launch_bounds__(256) void rank1_update_kernel(long nrows, const double* __restrict__ x, long ldx,
    const double* __restrict__ c, const double* __restrict__ beta, double* __restrict__ y, long ldy, int m, int tpc) {
  const int tx = threadIdx.x % tpc, ty = threadIdx.x / tpc, rpi = 256 / tpc;
  const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + rpi - 1) / rpi * rpi;
  const long r0 = (long)blockIdx.x * slab, r1 = min(nrows, r0 + slab);
  for (int c0 = 0; c0 < m; c0 += tpc) {
    const int j = c0 + tx;
    if (j >= m) continue;
    const double cj = c[j], bj = beta ? beta[j] : 1.0;
    long r = r0 + ty;
    for (; r + (long)(UNR - 1) * rpi < r1; r += (long)UNR * rpi) {
      double xv[UNR], yv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) { xv[u] = x[(r + (long)u * rpi) * ldx]; yv[u] = y[(r + (long)u * rpi) * ldy + j]; }
#pragma unroll
      for (int u = 0; u < UNR; ++u) y[(r + (long)u * rpi) * ldy + j] = fma(xv[u], cj, bj == 0.0 ? 0.0 : bj * yv[u]);   // beta 0: Y may hold anything
    }
    for (; r < r1; r += rpi) y[r * ldy + j] = fma(x[r * ldx], cj, bj == 0.0 ? 0.0 : bj * y[r * ldy + j]);
  }
}
// (c) y[r] *= s   (one column)
__global__ __launch_bounds__(256) void colscale1_kernel(long nrows, double* __restrict__ y, long ldy, double s) {
  const long stride = (long)gridDim.x * 256;
  long r = (long)blockIdx.x * 256 + threadIdx.x;
  for (; r + 3 * stride < nrows; r += 4 * stride) {
    const double a = y[r * ldy], b = y[(r + stride) * ldy], c = y[(r + 2 * stride) * ldy], d = y[(r + 3 * stride) * ldy];
    y[r * ldy] = a * s; y[(r + stride) * ldy] = b * s; y[(r + 2 * stride) * ldy] = c * s; y[(r + 3 * stride) * ldy] = d * s;
  }
  for (; r < nrows; r += stride) y[r * ldy] *= s;
}

// (d) one step of a column-wise Gram-Schmidt in ONE sweep (what OrthSelf of the reference does in three slot calls + the
// first of the next step, src/ops_orth.c:58-93):  x_k *= s (written back);  y_j += x_k c_j for the w columns behind it;
// partial[b*w + i] = sum over the block's rows of y_i(new) * y_0(new) — the k x 1 Gram the NEXT step asks for.
// base points at column k of the block (element (r, k) at base[r*ld]); w <= 64 columns, tpc = power of two >= w.
template <int UNR>
__global__ __launch_bounds__(256) void mgs_step_kernel(long nrows, double* __restrict__ base, long ld, double s,
    const double* __restrict__ c, int w, double* __restrict__ partial, long rows_per_block, int tpc) {
  // same decomposition, same UNR accumulators and the same reduction tree as panel_dot1_partial: the Gram column this kernel
  // leaves is BIT-IDENTICAL to the one the separate kernel would compute from the updated panel
  __shared__ double red[256];
  const int tx = threadIdx.x % tpc, ty = threadIdx.x / tpc, rpi = 256 / tpc;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
  const bool mine = tx < w;
  const int j = mine ? tx : 0;
  const double cj = c[j], c0 = c[0];
  double acc[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) acc[u] = 0.0;
  long r = r0 + ty;
  for (; r + (long)(UNR - 1) * rpi < r1; r += (long)UNR * rpi) {
    double xk[UNR], y0[UNR], yj[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const double* row = base + (r + (long)u * rpi) * ld;
      xk[u] = row[0]; y0[u] = row[1]; yj[u] = row[1 + j];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      double* row = base + (r + (long)u * rpi) * ld;
      const double q = xk[u] * s;
      const double y0n = fma(q, c0, y0[u]), yjn = fma(q, cj, yj[u]);
      if (tx == 0) row[0] = q;
      if (mine) row[1 + j] = yjn;
      acc[u] = fma(yjn, y0n, acc[u]);
    }
  }
  for (; r < r1; r += rpi) {
    double* row = base + r * ld;
    const double q = row[0] * s, y0n = fma(q, c0, row[1]), yjn = fma(q, cj, row[1 + j]);
    if (tx == 0) row[0] = q;
    if (mine) row[1 + j] = yjn;
    acc[0] = fma(yjn, y0n, acc[0]);
  }
  double t = 0.0;
#pragma unroll
  for (int u = 0; u < UNR; ++u) t += acc[u];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int h = rpi / 2; h > 0; h >>= 1) {
    if (ty < h) red[threadIdx.x] += red[threadIdx.x + h * tpc];
    __syncthreads();
  }
  if (ty == 0 && mine) partial[(long)blockIdx.x * w + tx] = red[tx];
}

static int pow2_at_least(int v, int cap) { int t = 1; while (t < v && t < cap) t *= 2; return t; }
// row slabs of the panel kernels: blocks of `rpb` rows (a multiple of the rpi rows a block walks per step), at most 2048 blocks
static void panel_geometry(int nrows, int rpi, long* nb_out, long* rpb_out) {
  long nb = ((long)nrows + 4L * rpi * 4 - 1) / (4L * rpi * 4);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  const long rpb = (((long)nrows + nb - 1) / nb + rpi - 1) / rpi * rpi;
  *nb_out = ((long)nrows + rpb - 1) / rpb; *rpb_out = rpb;
}
// d_out[i] = sum_r X[r, i] y[r], i < k: X = d_x (leading dimension ldx), y = d_y with stride ldy
extern "C" int gcge_hip_panel_dot1(int nrows, const double* d_x, long ldx, int k, const double* d_y, long ldy, double* d_out, void* stream) {
  if (k <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (nrows <= 0) return (int)hipMemsetAsync(d_out, 0, k * sizeof(double), st);
  const int tpc = pow2_at_least(k, 64), rpi = 256 / tpc;
  long nb, rpb;
  panel_geometry(nrows, rpi, &nb, &rpb);
  double* part = gcge_hip_partial_ws((size_t)nb * k);
  hipLaunchKernelGGL(panel_dot1_partial<4>, dim3((unsigned)nb), dim3(256), 0, st, (long)nrows, d_x, ldx, d_y, ldy, k, part, rpb, tpc);
  hipLaunchKernelGGL(reduce_partials, dim3((k + 63) / 64), dim3(1024), 0, st, part, (int)nb, k, d_out);
  return (int)hipGetLastError();
}
// Y[:, 0:m) = x c^T + Y diag(beta): d_c, d_beta device arrays of m (d_beta NULL: beta = 1)
extern "C" int gcge_hip_rank1_update(int nrows, const double* d_x, long ldx, const double* d_c, const double* d_beta, double* d_y,
                                     long ldy, int m, void* stream) {
  if (nrows <= 0 || m <= 0) return 0;
  const int tpc = pow2_at_least(m, 64), rpi = 256 / tpc;
  long nb = ((long)nrows + 4L * rpi * 4 - 1) / (4L * rpi * 4);
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(rank1_update_kernel<4>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (long)nrows, d_x, ldx, d_c, d_beta,
                     d_y, ldy, m, tpc);
  return (int)hipGetLastError();
}
// x_k *= s; Y[:, 0:w) += x_k c^T (Y = the w columns right behind x_k in the same block, leading dimension ld);
// d_dots[i] = Y_i(new) . Y_0(new), i < w.  -1: w > 64 (the caller takes the separate kernels)
extern "C" int gcge_hip_mgs_step(int nrows, double* d_xk, long ld, double s, const double* d_c, int w, double* d_dots, void* stream) {
  if (w <= 0 || w > 64) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (nrows <= 0) return (int)hipMemsetAsync(d_dots, 0, w * sizeof(double), st);
  const int tpc = pow2_at_least(w, 64), rpi = 256 / tpc;
  long nb, rpb;
  panel_geometry(nrows, rpi, &nb, &rpb);      // the SAME slabs as gcge_hip_panel_dot1 would take for these w columns
  double* part = gcge_hip_partial_ws((size_t)nb * w);
  hipLaunchKernelGGL(mgs_step_kernel<4>, dim3((unsigned)nb), dim3(256), 0, st, (long)nrows, d_xk, ld, s, d_c, w, part, rpb, tpc);
  hipLaunchKernelGGL(reduce_partials, dim3(1), dim3(1024), 0, st, part, (int)nb, w, d_dots);
  return (int)hipGetLastError();
}
extern "C" int gcge_hip_colscale1(int nrows, double* d_y, long ldy, double s, void* stream) {
  if (nrows <= 0) return 0;
  long nb = ((long)nrows + 1023) / 1024;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(colscale1_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (long)nrows, d_y, ldy, s);
  return (int)hipGetLastError();
}

extern "C" int gcge_hip_coldots(int nrows, const double* d_x, long ldx, const double* d_y, long ldy, int m,
                                double* d_out, void* stream) {
  gcge_hip_apply_pending();
  if (m <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (nrows <= 0) return (int)hipMemsetAsync(d_out, 0, m * sizeof(double), st);
  long nb = ((long)nrows + 255) / 256;
  if (nb > 2048) nb = 2048;
  const long rpb = (((long)nrows + nb - 1) / nb + 3) / 4 * 4;
  nb = ((long)nrows + rpb - 1) / rpb;
  double* part = gcge_hip_partial_ws((size_t)nb * m);
  hipLaunchKernelGGL(coldots_partial, dim3((unsigned)nb), dim3(256), 0, st, (long)nrows, d_x, ldx, d_y, ldy, m,
                     part, rpb);
  hipLaunchKernelGGL(reduce_partials, dim3((m + 63) / 64), dim3(1024), 0, st, part, (int)nb, m, d_out);
  return (int)hipGetLastError();
}

// d_out[0:m) = column dots x.y, d_out[m:2m) = y.y: the same slabs and the same summation order as two gcge_hip_coldots calls
extern "C" int gcge_hip_coldots2(int nrows, const double* d_x, long ldx, const double* d_y, long ldy, int m,
                                 double* d_out, void* stream) {
  gcge_hip_apply_pending();
  if (m <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (nrows <= 0) return (int)hipMemsetAsync(d_out, 0, 2 * (size_t)m * sizeof(double), st);
  long nb = ((long)nrows + 255) / 256;
  if (nb > 2048) nb = 2048;
  const long rpb = (((long)nrows + nb - 1) / nb + 3) / 4 * 4;
  nb = ((long)nrows + rpb - 1) / rpb;
  double* part = gcge_hip_partial_ws((size_t)nb * 2 * m);
  hipLaunchKernelGGL(coldots2_partial, dim3((unsigned)nb), dim3(256), 0, st, (long)nrows, d_x, ldx, d_y, ldy, m, part, rpb);
  hipLaunchKernelGGL(reduce_partials, dim3((2 * m + 63) / 64), dim3(1024), 0, st, part, (int)nb, 2 * m, d_out);
  return (int)hipGetLastError();
}

// d_out[j] = sum_r (w[r,j] - lambda_j x[r,j])^2, j < m (device, fixed summation order); d_lambda: m doubles on the device
extern "C" int gcge_hip_resid_sq(int nrows, const double* d_w, long ldw, const double* d_x, long ldx, int m, const double* d_lambda,
                                 double* d_out, void* stream) {
  gcge_hip_apply_pending();
  if (nrows <= 0 || m <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  long nb = ((long)nrows + 255) / 256; if (nb > 2048) nb = 2048;
  const long rpb = (((long)nrows + nb - 1) / nb + 3) / 4 * 4;
  nb = ((long)nrows + rpb - 1) / rpb;
  double* part = gcge_hip_partial_ws((size_t)nb * m);
  hipLaunchKernelGGL(resid_sq_partial, dim3((unsigned)nb), dim3(256), 0, st, (long)nrows, d_w, ldw, d_x, ldx, m, d_lambda, part, rpb);
  hipLaunchKernelGGL(reduce_partials, dim3((m + 63) / 64), dim3(1024), 0, st, part, (int)nb, m, d_out);
  return (int)hipGetLastError();
}

// npass slabs of [nblocks][m_ps] partials (m_ps <= 16 columns each; slab ps starts at partial + ps*slab_stride) ->
// out[16*ps + j]: one block per slab, 64 row lanes x 16 columns, fixed order
__global__ __launch_bounds__(1024) void reduce_partials16(const double* __restrict__ partial, int nblocks, long slab_stride,
    int ncols, double* __restrict__ out) {
  __shared__ double red[64][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c0 = 16 * blockIdx.x, m = min(16, ncols - c0);
  const double* src = partial + (long)blockIdx.x * slab_stride;
  double s0 = 0.0, s1 = 0.0;
  if (tx < m) {
    int b = ty;
    for (; b + 64 < nblocks; b += 128) { s0 += src[(long)b * m + tx]; s1 += src[(long)(b + 64) * m + tx]; }
    for (; b < nblocks; b += 64) s0 += src[(long)b * m + tx];
  }
  red[ty][tx] = s0 + s1;
  __syncthreads();
  for (int h = 32; h > 0; h >>= 1) {
    if (ty < h) red[ty][tx] += red[ty + h][tx];
    __syncthreads();
  }
  if (ty == 0 && tx < m) out[c0 + tx] = red[0][tx];
}
// same with cpp (16, 32 or 64) columns per slab
__global__ __launch_bounds__(1024) void reduce_partials_slabs(const double* __restrict__ partial, int nblocks, long slab_stride,
    int cpp, int ncols, double* __restrict__ out) {
  __shared__ double red[1024];
  const int tx = threadIdx.x % cpp, ty = threadIdx.x / cpp, nl = 1024 / cpp;
  const int c0 = cpp * blockIdx.x, m = min(cpp, ncols - c0);
  const double* src = partial + (long)blockIdx.x * slab_stride;
  double s0 = 0.0;
  if (tx < m) for (int b = ty; b < nblocks; b += nl) s0 += src[(long)b * m + tx];
  red[threadIdx.x] = s0;
  __syncthreads();
  for (int h = nl / 2; h > 0; h >>= 1) {
    if (ty < h) red[threadIdx.x] += red[threadIdx.x + h * cpp];
    __syncthreads();
  }
  if (ty == 0 && tx < m) out[c0 + tx] = red[tx];
}
extern "C" void gcge_hip_reduce_partials_slabs(const double* d_partial, int nblocks, long slab_stride, int cpp, int ncols,
                                               double* d_out, void* stream) {
  hipLaunchKernelGGL(reduce_partials_slabs, dim3((ncols + cpp - 1) / cpp), dim3(1024), 0, (hipStream_t)stream, d_partial,
                     nblocks, slab_stride, cpp, ncols, d_out);
}
extern "C" void gcge_hip_reduce_partials16(const double* d_partial, int nblocks, long slab_stride, int ncols, double* d_out,
                                           void* stream) {
  hipLaunchKernelGGL(reduce_partials16, dim3((ncols + 15) / 16), dim3(1024), 0, (hipStream_t)stream, d_partial, nblocks,
                     slab_stride, ncols, d_out);
}

// exported so other translation units reuse the same fixed-order reduction
extern "C" void gcge_hip_reduce_partials(const double* d_partial, int nblocks, int len, double* d_out, void* stream) {
  hipLaunchKernelGGL(reduce_partials, dim3((len + 63) / 64), dim3(1024), 0, (hipStream_t)stream, d_partial, nblocks,
                     len, d_out);
}

extern "C" int gcge_hip_fill_uniform(int nrows, long row_begin, long nglobal, double* d_y, long ldy, int c0,
                                     int m, unsigned long long seed, void* stream) {
  if (nrows <= 0 || m <= 0) return 0;
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for((long)nrows * m)), dim3(256), 0, (hipStream_t)stream,
                     (long)nrows, row_begin, nglobal, d_y, ldy, c0, m, seed);
  return (int)hipGetLastError();
}

extern "C" int gcge_hip_colmajor_to_rowmajor(int nrows, int m, const double* d_src, long lds, double* d_dst,
                                             long ldd, void* stream) {
  if (nrows <= 0 || m <= 0) return 0;
  dim3 grid((unsigned)((nrows + 31) / 32), (unsigned)((m + 31) / 32));
  hipLaunchKernelGGL(colmajor_to_rowmajor, grid, dim3(256), 0, (hipStream_t)stream, (long)nrows, m, d_src, lds,
                     d_dst, ldd);
  return (int)hipGetLastError();
}
extern "C" int gcge_hip_rowmajor_to_colmajor(int nrows, int m, const double* d_src, long lds, double* d_dst,
                                             long ldd, void* stream) {
  if (nrows <= 0 || m <= 0) return 0;
  dim3 grid((unsigned)((nrows + 31) / 32), (unsigned)((m + 31) / 32));
  hipLaunchKernelGGL(rowmajor_to_colmajor, grid, dim3(256), 0, (hipStream_t)stream, (long)nrows, m, d_src, lds,
                     d_dst, ldd);
  return (int)hipGetLastError();
}
