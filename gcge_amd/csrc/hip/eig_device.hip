// K7 on the device — all eigenpairs of the small dense symmetric matrix of the Rayleigh-Ritz step.
//
// Replaces dsyevx('V','A','U') of the reference (src/ops_eig_sol_gcg.c:1201-1203; (V-C) x (V-C) <= 656^2) for the
// sizes where the host solver (csrc/host/eig_sym.c, same algorithm) dominates an outer iteration: N = 512 costs
// 64-150 ms on the host, 0.75 s of a 2.3 s solve at BASELINE config 3's shape and 6.2 s at config 4's.
//
// Same method as the host code, so the same accuracy (eigenvalues to O(N eps ||A||), orthonormal vectors):
//   1. Householder reduction to tridiagonal form on the device.  The matrix (<= 3.4 MB) stays in L2; every step is
//      three small launches on the back-end's stream: reflector from column k (one block), p = beta A22 v (one wave per
//      column of the full symmetric A22), rank-2 update A22 -= v w^T + w v^T (w = p - K v formed on the fly).
//   2. Q = H_0 ... H_{N-3} accumulated on the device (one launch per reflector, one wave per column).
//   3. Implicit-shift QL on (d, e) on the HOST — O(N^2) scalar work, no parallelism to speak of — which RECORDS its
//      plane rotations instead of applying them.
//   4. The recorded rotations (~1.7 N^2 of them) are replayed on the device against the columns of Q: the rows of Q are
//      independent, one lane per row carries the running column through a whole sweep (1 load + 1 store per rotation,
//      the (c, s) pairs arrive through scalar loads).
//   5. Ascending sort on the host, columns gathered while they are copied back.
// Hooked into the GCG driver of libgcge_host.so through GCGE_SetSymEigHook (include/gcge_ops.h); OPS_HIP_Set registers
// it for N >= 192 (below that the launch count, ~4 N, costs more than the host solver).  The reference's own stack
// keeps calling its LAPACK.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <vector>

#include "gcge_hip.h"
#include "gcge_hip_internal.h"

namespace gcge {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int mk = 32; mk >= 1; mk >>= 1) {
    const int lo = __shfl_xor(__double2loint(v), mk, 64), hi = __shfl_xor(__double2hiint(v), mk, 64);
    v += __hiloint2double(hi, lo);
  }
  return v;
}
__device__ __forceinline__ double block_sum256(double v, double* red) {   // 256 threads; every thread gets the sum
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// reflector annihilating M(k+2.., k): v in vbuf[0..len) (also stored in the annihilated column), beta[k], d[k], e[k]
__global__ __launch_bounds__(256) void eig_house(int n, int k, double* __restrict__ m, double* __restrict__ d, double* __restrict__ e,
    double* __restrict__ betas, double* __restrict__ vbuf) {
  __shared__ double red[4];
  const int len = n - k - 1;
  double* x = m + (size_t)k * n + (k + 1);
  double sc = 0.0;
  for (int i = threadIdx.x; i < len; i += 256) sc += fabs(x[i]);
  const double scale = block_sum256(sc, red);
  if (scale == 0.0) {
    for (int i = threadIdx.x; i < len; i += 256) { vbuf[i] = 0.0; x[i] = 0.0; }
    if (threadIdx.x == 0) { d[k] = m[(size_t)k * n + k]; e[k] = 0.0; betas[k] = 0.0; }
    return;
  }
  double sg = 0.0;
  for (int i = threadIdx.x; i < len; i += 256) { const double t = x[i] / scale; sg += t * t; }
  const double sigma = block_sum256(sg, red);
  const double x0 = x[0] / scale;
  const double mu = (x0 >= 0.0) ? -sqrt(sigma) : sqrt(sigma);
  const double v0 = x0 - mu;
  const double vtv = sigma - x0 * x0 + v0 * v0;
  __syncthreads();                                   // every thread has read x[0] before it is overwritten
  for (int i = threadIdx.x; i < len; i += 256) { const double t = (i == 0) ? v0 : x[i] / scale; vbuf[i] = t; x[i] = t; }
  if (threadIdx.x == 0) { d[k] = m[(size_t)k * n + k]; e[k] = scale * mu; betas[k] = 2.0 / vtv; }
}

// p[j] = beta * sum_i A22(i, j) v[i]: one wave per column of the trailing block (full symmetric storage: contiguous)
__global__ __launch_bounds__(256) void eig_symv(int n, int k, const double* __restrict__ m, const double* __restrict__ betas,
    const double* __restrict__ vbuf, double* __restrict__ pbuf) {
  const int len = n - k - 1;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= len) return;
  const double* col = m + (size_t)(k + 1 + j) * n + (k + 1);
  double s = 0.0;
  for (int i = lane; i < len; i += 64) s = fma(col[i], vbuf[i], s);
  s = wave_sum(s);
  if (lane == 0) pbuf[j] = betas[k] * s;
}

// A22 -= v w^T + w v^T with w = p - K v, K = beta/2 v.p (every block recomputes the scalar: len <= 656 products)
__global__ __launch_bounds__(256) void eig_rank2(int n, int k, double* __restrict__ m, const double* __restrict__ betas,
    const double* __restrict__ vbuf, const double* __restrict__ pbuf) {
  __shared__ double red[4];
  const int len = n - k - 1;
  double t = 0.0;
  for (int i = threadIdx.x; i < len; i += 256) t = fma(vbuf[i], pbuf[i], t);
  const double K = 0.5 * betas[k] * block_sum256(t, red);
  // block = 8 columns x 32-row strips ... simple: blockIdx.x = column, threads over rows
  for (int j = blockIdx.x; j < len; j += gridDim.x) {
    double* col = m + (size_t)(k + 1 + j) * n + (k + 1);
    const double vj = vbuf[j], wj = pbuf[j] - K * vj;
    for (int i = threadIdx.x; i < len; i += 256) {
      const double vi = vbuf[i], wi = pbuf[i] - K * vi;
      col[i] -= vi * wj + wi * vj;
    }
  }
}

// the last two diagonal entries and the last coupling
__global__ void eig_tail(int n, const double* __restrict__ m, double* __restrict__ d, double* __restrict__ e) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (n >= 2) { d[n - 2] = m[(size_t)(n - 2) * n + (n - 2)]; e[n - 2] = m[(size_t)(n - 2) * n + (n - 1)]; }
  d[n - 1] = m[(size_t)(n - 1) * n + (n - 1)]; e[n - 1] = 0.0;
}

__global__ __launch_bounds__(256) void eig_identity(int n, double* __restrict__ q) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < (size_t)n * n) q[idx] = (idx / n == idx % n) ? 1.0 : 0.0;
}

// Q[k+1.., k+1..] <- H_k Q[k+1.., k+1..]: per column, t = beta v.col; col -= t v (one wave per column)
__global__ __launch_bounds__(256) void eig_apply_q(int n, int k, const double* __restrict__ m, const double* __restrict__ betas,
    double* __restrict__ q) {
  const int len = n - k - 1;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const double bk = betas[k];
  if (j >= len || bk == 0.0) return;
  const double* v = m + (size_t)k * n + (k + 1);
  double* col = q + (size_t)(k + 1 + j) * n + (k + 1);
  double t = 0.0;
  for (int i = lane; i < len; i += 64) t = fma(v[i], col[i], t);
  t = bk * wave_sum(t);
  for (int i = lane; i < len; i += 64) col[i] -= t * v[i];
}


// ---- fused replay: EIG_K consecutive sweeps in one pass over the columns ---------------------------------------------
// The plain replay above is latency-bound (one L2 round trip per eight rotations: 62 ms for the 4.3e5 rotations of
// N = 656).  Rotation p of sweep j (columns hi_c - p, hi_c - p + 1 of the group's padded range) depends only on
// rotation p - 1 of the same sweep and rotation p + 1 of the previous one, so at "time" t = p + 2 j the rotations
// (j, t - 2 j), j = 0 .. K-1, touch disjoint column pairs: a lane carries a window of 2 K columns of its row in
// registers, per time step takes ONE new column, applies K independent rotations and retires ONE finished column.
// Sweeps of a group are padded with identity rotations to a common column range; the coefficients come time-major.
// Columns and coefficients move between memory and LDS in chunks of EIG_TC time steps (64 independent loads per lane
// issued back to back, then 64 stores), so the stepping loop itself only touches LDS: a first version that loaded and
// stored one column per step from global memory ran at 1900 cycles per step — loads and stores share vmcnt on this
// target, every prefetched load waited for the stores issued after it, and the wave-uniform coefficient loads (SMEM
// returns out of order) forced lgkmcnt(0) in every step.  Register roles rotate with period 2 K: the loop is unrolled
// by that, every index is a compile-time constant.  One wave per block: no barriers, LDS traffic of a wave is ordered.
constexpr int EIG_K = 8, EIG_W = 2 * EIG_K, EIG_TC = 64;
struct EigGroup { int hi_c, R; long off; };      // columns hi_c + 1 - x, x = 0 .. R; coefficients at cs[off ..), Tpad steps x 2 K
__global__ __launch_bounds__(64) void eig_replay_fused(int n, double* __restrict__ q, const EigGroup* __restrict__ grp, int ngroup,
    const double* __restrict__ cs) {
  __shared__ double colb[EIG_TC][64];            // chunk of columns: read at its step, overwritten with the retired column
  __shared__ double cfb[EIG_TC][2 * EIG_K];      // (c, s) of the chunk's time steps
  const int lane = threadIdx.x;
  const int r = blockIdx.x * 64 + lane;
  const bool live = r < n;
  double* qr = q + (live ? r : n - 1);           // surplus lanes shadow the last row and never store
  for (int g = 0; g < ngroup; ++g) {
    const int hi_c = grp[g].hi_c, R = grp[g].R;
    const double* coef = cs + grp[g].off;
    const int Tp = R + 2 * EIG_K - 1;             // the host pads the coefficient block to a multiple of EIG_TC steps
    double w[EIG_W];
#pragma unroll
    for (int u = 0; u < EIG_W; ++u) w[u] = 0.0;
    w[0] = qr[(size_t)(hi_c + 1) * n];            // x = 0
    for (int tc = 0; tc < Tp; tc += EIG_TC) {
      // chunk in: columns x = tc + 1 + c (clamped into the range: the surplus meets identity rotations), coefficients
#pragma unroll 16
      for (int c = 0; c < EIG_TC; ++c) colb[c][lane] = qr[(size_t)(hi_c + 1 - min(tc + 1 + c, R)) * n];
      {
        const double* src = coef + (size_t)(tc + lane) * (2 * EIG_K);
#pragma unroll
        for (int e = 0; e < 2 * EIG_K; ++e) cfb[lane][e] = src[e];
      }
      for (int c0 = 0; c0 < EIG_TC; c0 += EIG_W) {
#pragma unroll
        for (int u = 0; u < EIG_W; ++u) {
          const int c = c0 + u;                   // t = tc + c;  t mod 2K == u because tc and c0 are multiples of 2K
          w[(u + 1) % EIG_W] = colb[c][lane];
#pragma unroll
          for (int j = 0; j < EIG_K; ++j) {
            const int ih = ((u - 2 * j) % EIG_W + EIG_W) % EIG_W, il = ((u - 2 * j + 1) % EIG_W + EIG_W) % EIG_W;
            const double cc = cfb[c][2 * j], ss = cfb[c][2 * j + 1];
            const double hv = w[ih], lv = w[il];
            w[ih] = fma(ss, lv, cc * hv);
            w[il] = fma(cc, lv, -ss * hv);
          }
          colb[c][lane] = w[((u - 2 * EIG_K + 2) % EIG_W + EIG_W) % EIG_W];   // column x = t - 2K + 2 is final
        }
      }
      // chunk out
#pragma unroll 16
      for (int c = 0; c < EIG_TC; ++c) {
        const int xo = tc + c - 2 * EIG_K + 2;
        if (xo >= 0 && xo <= R && live) qr[(size_t)(hi_c + 1 - xo) * n] = colb[c][lane];
      }
    }
  }
}

}  // namespace gcge

using namespace gcge;

// a recorded QL sweep: rotations in columns (i, i+1) for i = first .. last (descending), cs[2 (off + first - i)] = c, +1 = s
struct EigSweep { int first, last; long off; };
// implicit QL on the tridiagonal (d, e), e[k] couples k and k+1 — the iteration of csrc/host/eig_sym.c with the
// rotations recorded (descending column index inside a sweep) instead of applied.  0, or l+1 if eigenvalue l failed.
// sqrt(f^2 + g^2); the libm hypot (over/underflow-proof, ~50 ns) only outside the range where the squares are safe
static inline double pythag(double f, double g) {
  const double af = fabs(f), ag = fabs(g), mx = af > ag ? af : ag;
  if (mx < 1e150 && mx > 1e-150) return sqrt(f * f + g * g);
  return hypot(f, g);
}
static int ql_record(int n, double* d, double* e, std::vector<EigSweep>& sweeps, std::vector<double>& cs) {
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= DBL_EPSILON * dd) break;
      }
      if (m != l) {
        if (iter++ == 60) return l + 1;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        const long off = (long)(cs.size() / 2);
        int i, last = m;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          e[i + 1] = r = pythag(f, g);
          if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          cs.push_back(c); cs.push_back(s); last = i;
        }
        if (last <= m - 1) sweeps.push_back(EigSweep{m - 1, last, off});
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[m] = 0.0;
      }
    } while (m != l);
  }
  return 0;
}

struct EigWs { double *m, *q, *d, *e, *betas, *v, *p, *cs; EigSweep* sw; int cap_n; size_t cap_cs, cap_sw; double* h_pin; size_t cap_pin; double* h_cs; size_t cap_hcs; };
static EigWs g_eig = {};

// All eigenpairs of the symmetric n x n matrix a (column-major, ld lda; only the triangle `uplo` is read).
// w: ascending eigenvalues, z (ld ldz): the matching orthonormal eigenvectors (host memory, as GCGE_SymEig).
static long g_symeig_calls = 0;
extern "C" long gcge_hip_symeig_calls(void) { return g_symeig_calls; }   /* how often the device solver ran (tests: only on behalf of the HIP table) */
extern "C" int gcge_hip_symeig(char uplo, int n, const double* a, int lda, double* w, double* z, int ldz) {
  ++g_symeig_calls;
  if (n <= 0) return 0;
  if (n == 1) { w[0] = a[0]; z[0] = 1.0; return 0; }
  if (gcge_hip_init(-1) != 0) return -1;
  hipStream_t st = (hipStream_t)gcge_hip_stream();
  EigWs& g = g_eig;
  const size_t nn = (size_t)n * n;
  if (n > g.cap_n) {
    GCGE_HIP_CHECK(hipStreamSynchronize(st));
    if (g.m) { hipFree(g.m); hipFree(g.q); hipFree(g.d); }
    g.cap_n = n + 64;
    const size_t cn = (size_t)g.cap_n;
    GCGE_HIP_CHECK(hipMalloc(&g.m, cn * cn * sizeof(double)));
    GCGE_HIP_CHECK(hipMalloc(&g.q, cn * cn * sizeof(double)));
    GCGE_HIP_CHECK(hipMalloc(&g.d, 5 * cn * sizeof(double)));
    g.e = g.d + cn; g.betas = g.e + cn; g.v = g.betas + cn; g.p = g.v + cn;
  }
  if (nn + 2 * (size_t)n > g.cap_pin) {
    GCGE_HIP_CHECK(hipStreamSynchronize(st));
    if (g.h_pin) hipHostFree(g.h_pin);
    g.cap_pin = ((size_t)g.cap_n * g.cap_n + 2 * (size_t)g.cap_n);
    GCGE_HIP_CHECK(hipHostMalloc(&g.h_pin, g.cap_pin * sizeof(double)));
  }
  // full symmetric copy from the referenced triangle, then to the device
  const bool upper = (uplo == 'U' || uplo == 'u');
  GCGE_HIP_CHECK(hipStreamSynchronize(st));          // the pinned buffer may still feed the previous call's upload
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) {
      const double t = upper ? a[(size_t)i * lda + j] : a[(size_t)j * lda + i];
      g.h_pin[(size_t)j * n + i] = t; g.h_pin[(size_t)i * n + j] = t;
    }
  GCGE_HIP_CHECK(hipMemcpyAsync(g.m, g.h_pin, nn * sizeof(double), hipMemcpyHostToDevice, st));
  static const bool timing = getenv("GCGE_EIG_TIMING") != nullptr;   // phase times on stderr (tuning aid)
  hipEvent_t tev[5];
  if (timing) { for (auto& ev_ : tev) GCGE_HIP_CHECK(hipEventCreate(&ev_)); GCGE_HIP_CHECK(hipEventRecord(tev[0], st)); }
  // 1. tridiagonal reduction
  for (int k = 0; k < n - 2; ++k) {
    const int len = n - k - 1;
    hipLaunchKernelGGL(eig_house, dim3(1), dim3(256), 0, st, n, k, g.m, g.d, g.e, g.betas, g.v);
    hipLaunchKernelGGL(eig_symv, dim3((len + 3) / 4), dim3(256), 0, st, n, k, g.m, g.betas, g.v, g.p);
    hipLaunchKernelGGL(eig_rank2, dim3(len), dim3(256), 0, st, n, k, g.m, g.betas, g.v, g.p);
  }
  hipLaunchKernelGGL(eig_tail, dim3(1), dim3(64), 0, st, n, g.m, g.d, g.e);
  GCGE_HIP_CHECK(hipMemcpyAsync(g.h_pin, g.d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  GCGE_HIP_CHECK(hipMemcpyAsync(g.h_pin + n, g.e, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  if (timing) GCGE_HIP_CHECK(hipEventRecord(tev[1], st));
  // 2. Q (runs while the host works on the tridiagonal matrix)
  hipLaunchKernelGGL(eig_identity, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, n, g.q);
  hipEvent_t ev_de;
  GCGE_HIP_CHECK(hipEventCreateWithFlags(&ev_de, hipEventDisableTiming));
  GCGE_HIP_CHECK(hipEventRecord(ev_de, st));
  for (int k = n - 3; k >= 0; --k) {
    const int len = n - k - 1;
    hipLaunchKernelGGL(eig_apply_q, dim3((len + 3) / 4), dim3(256), 0, st, n, k, g.m, g.betas, g.q);
  }
  if (timing) GCGE_HIP_CHECK(hipEventRecord(tev[2], st));
  GCGE_HIP_CHECK(hipEventSynchronize(ev_de));
  GCGE_HIP_CHECK(hipEventDestroy(ev_de));
  // 3. QL on the host, rotations recorded
  auto wall = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e3 * ts.tv_sec + 1e-6 * ts.tv_nsec; };
  const double w0 = wall();
  std::vector<double> d(g.h_pin, g.h_pin + n), e(g.h_pin + n, g.h_pin + 2 * n);
  std::vector<EigSweep> sweeps; std::vector<double> cs;
  cs.reserve((size_t)4 * n * n);
  const int info = ql_record(n, d.data(), e.data(), sweeps, cs);
  if (info != 0) { GCGE_HIP_CHECK(hipStreamSynchronize(st)); return info; }
  const double w1 = wall();
  // 4. replay on the device: groups of EIG_K consecutive sweeps, padded to a common column range, coefficients time-major
  size_t n_rot = cs.size() / 2;
  if (!sweeps.empty()) {
    std::vector<EigGroup> groups;
    size_t total = 0;
    for (size_t a = 0; a < sweeps.size(); a += EIG_K) {
      const size_t bnd = std::min(sweeps.size(), a + EIG_K);
      int hi_c = 0, lo_c = n;
      for (size_t q = a; q < bnd; ++q) { hi_c = std::max(hi_c, sweeps[q].first); lo_c = std::min(lo_c, sweeps[q].last); }
      const int R = hi_c - lo_c + 1;
      groups.push_back(EigGroup{hi_c, R, (long)total});
      total += (size_t)((R + 2 * EIG_K - 1 + EIG_TC - 1) / EIG_TC * EIG_TC) * (2 * EIG_K);
    }
    if (total > g.cap_hcs) {
      GCGE_HIP_CHECK(hipStreamSynchronize(st));
      if (g.h_cs) hipHostFree(g.h_cs);
      g.cap_hcs = total * 2;
      GCGE_HIP_CHECK(hipHostMalloc(&g.h_cs, g.cap_hcs * sizeof(double)));
    }
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      const EigGroup& G = groups[gi];
      double* co = g.h_cs + G.off;
      const int Tp = (G.R + 2 * EIG_K - 1 + EIG_TC - 1) / EIG_TC * EIG_TC;
      for (int t = 0; t < Tp; ++t)
        for (int j = 0; j < EIG_K; ++j) { co[(size_t)t * 2 * EIG_K + 2 * j] = 1.0; co[(size_t)t * 2 * EIG_K + 2 * j + 1] = 0.0; }
      for (int j = 0; j < EIG_K && gi * EIG_K + j < sweeps.size(); ++j) {
        const EigSweep& S = sweeps[gi * EIG_K + j];
        for (int i = S.first; i >= S.last; --i) {
          const int pp = G.hi_c - i, t = pp + 2 * j;
          co[(size_t)t * 2 * EIG_K + 2 * j] = cs[2 * (S.off + S.first - i)];
          co[(size_t)t * 2 * EIG_K + 2 * j + 1] = cs[2 * (S.off + S.first - i) + 1];
        }
      }
    }
    if (total > g.cap_cs) { if (g.cs) { GCGE_HIP_CHECK(hipStreamSynchronize(st)); hipFree(g.cs); } g.cap_cs = total * 2; GCGE_HIP_CHECK(hipMalloc(&g.cs, g.cap_cs * sizeof(double))); }
    const size_t gbytes = groups.size() * sizeof(EigGroup);
    if (groups.size() > g.cap_sw) { if (g.sw) { GCGE_HIP_CHECK(hipStreamSynchronize(st)); hipFree(g.sw); } g.cap_sw = groups.size() * 2; GCGE_HIP_CHECK(hipMalloc(&g.sw, g.cap_sw * sizeof(EigGroup))); }
    GCGE_HIP_CHECK(hipMemcpyAsync(g.cs, g.h_cs, total * sizeof(double), hipMemcpyHostToDevice, st));
    GCGE_HIP_CHECK(hipMemcpyAsync(g.sw, groups.data(), gbytes, hipMemcpyHostToDevice, st));
    if (timing) GCGE_HIP_CHECK(hipEventRecord(tev[3], st));
    hipLaunchKernelGGL(eig_replay_fused, dim3((n + 63) / 64), dim3(64), 0, st, n, g.q, (const EigGroup*)g.sw, (int)groups.size(), g.cs);
    if (timing) GCGE_HIP_CHECK(hipEventRecord(tev[4], st));
    GCGE_HIP_CHECK(hipStreamSynchronize(st));          // `groups` is pageable and leaves scope here
  }
  // 5. back to the host, ascending
  GCGE_HIP_CHECK(hipMemcpyAsync(g.h_pin, g.q, nn * sizeof(double), hipMemcpyDeviceToHost, st));
  GCGE_HIP_CHECK(hipStreamSynchronize(st));          // (also: cs / sweeps are pageable and leave scope below)
  if (timing) {
    float t01 = 0, t12 = 0, t34 = 0;
    hipEventElapsedTime(&t01, tev[0], tev[1]); hipEventElapsedTime(&t12, tev[1], tev[2]);
    if (!sweeps.empty()) hipEventElapsedTime(&t34, tev[3], tev[4]);
    fprintf(stderr, "gcge_hip_symeig n=%d: tridiagonalisation %.2f ms, Q %.2f ms, replay of %zu rotations in %zu sweeps %.2f ms; host QL %.2f ms, "
            "QL end -> results on the host %.2f ms\n", n, t01, t12, n_rot, sweeps.size(), t34, w1 - w0, wall() - w1);
    for (auto& ev_ : tev) hipEventDestroy(ev_);
  }
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int i = 1; i < n; ++i) {                      // stable insertion on the permutation, as the host solver
    const int pi = perm[i]; const double key = d[pi]; int j;
    for (j = i - 1; j >= 0 && d[perm[j]] > key; --j) perm[j + 1] = perm[j];
    perm[j + 1] = pi;
  }
  for (int j = 0; j < n; ++j) {
    w[j] = d[perm[j]];
    memcpy(z + (size_t)j * ldz, g.h_pin + (size_t)perm[j] * n, (size_t)n * sizeof(double));
  }
  return 0;
}
