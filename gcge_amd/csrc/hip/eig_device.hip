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
#include "gcge_solver.h"

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

// One launch per column k of the Householder reduction.  The rank-2 update of step k-1, A22 -= v w^T + w v^T with
// w = p - K v and K = beta/2 v.p, is still PENDING when the launch starts; every block
//   1. rebuilds w of step k-1 from (v, p) of that step and applies the pending update to column k only (its own copy, in LDS),
//   2. forms the reflector of step k from that column (block 0 also stores it, with d[k], e[k], beta[k]),
//   3. for its columns j > k: applies the pending update, stores the column and reduces p_k[j] = beta_k (column . v_k) while
//      the column is in registers.
// The reduction needs one pass over the trailing block per column instead of two (symv, then rank-2 update) and one launch
// instead of three; what stays is the one device-wide dependency per column (p_k complete before anybody forms w_k), which is
// the launch boundary.  Steps 1-2 are redundant in every block: ~3 n loads from L2 and four block reductions.
constexpr int EIG_MAX_N = 1184;                    // a 16-row slab of Q with its margins fills the 160 KB of LDS (eig_apply_tiles)
__global__ __launch_bounds__(256) void eig_step(int n, int k, double* __restrict__ m, double* __restrict__ d, double* __restrict__ e,
    double* __restrict__ betas, double* __restrict__ vs, const double* __restrict__ pprev, double* __restrict__ pnext) {
  __shared__ double red[4];
  __shared__ double vp[EIG_MAX_N], wp[EIG_MAX_N], vk[EIG_MAX_N];
  const int len = n - k - 1, lp = n - k;           // rows k+1 .. n-1 (this step), rows k .. n-1 (the pending one)
  const bool pending = k > 0;
  if (pending) {
    const double* vprev = vs + (size_t)(k - 1) * n;
    double t = 0.0;
    for (int i = threadIdx.x; i < lp; i += 256) { const double a = vprev[i], b = pprev[i]; vp[i] = a; wp[i] = b; t = fma(a, b, t); }
    const double K = 0.5 * betas[k - 1] * block_sum256(t, red);
    for (int i = threadIdx.x; i < lp; i += 256) wp[i] -= K * vp[i];      // (each thread rewrites the entries it wrote)
    __syncthreads();
  }
  // column k, rows k+1 .. : x[i] (row k + 1 + i), and the diagonal entry
  const double* colk = m + (size_t)k * n + k;
  const double vp0 = pending ? vp[0] : 0.0, wp0 = pending ? wp[0] : 0.0;
  double sc = 0.0;
  for (int i = threadIdx.x; i < len; i += 256) {
    double x = colk[i + 1];
    if (pending) x -= vp[i + 1] * wp0 + wp[i + 1] * vp0;
    vk[i] = x; sc += fabs(x);
  }
  const double scale = block_sum256(sc, red);      // (its barriers also publish vk)
  double beta_k = 0.0;
  if (scale == 0.0) {
    for (int i = threadIdx.x; i < len; i += 256) vk[i] = 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) e[k] = 0.0;
  } else {
    double sg = 0.0;
    for (int i = threadIdx.x; i < len; i += 256) { const double t = vk[i] / scale; sg += t * t; }
    const double sigma = block_sum256(sg, red);
    const double x0 = vk[0] / scale;
    const double mu = (x0 >= 0.0) ? -sqrt(sigma) : sqrt(sigma);
    const double v0 = x0 - mu;
    const double vtv = sigma - x0 * x0 + v0 * v0;
    beta_k = 2.0 / vtv;
    __syncthreads();                               // every thread has read vk[0]
    for (int i = threadIdx.x; i < len; i += 256) vk[i] = (i == 0) ? v0 : vk[i] / scale;
    if (blockIdx.x == 0 && threadIdx.x == 0) e[k] = scale * mu;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    double* vout = vs + (size_t)k * n;
    for (int i = threadIdx.x; i < len; i += 256) vout[i] = vk[i];
    if (threadIdx.x == 0) {
      double dk = colk[0];
      if (pending) dk -= 2.0 * vp0 * wp0;
      d[k] = dk; betas[k] = beta_k;
    }
  }
  // the trailing block: one wave per column
  const int lane = threadIdx.x & 63;
  for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < len; j += gridDim.x * 4) {
    double* col = m + (size_t)(k + 1 + j) * n + (k + 1);
    const double vj = pending ? vp[j + 1] : 0.0, wj = pending ? wp[j + 1] : 0.0;
    double s = 0.0;
    for (int i = lane; i < len; i += 64) {
      double a = col[i];
      if (pending) { a -= vp[i + 1] * wj + wp[i + 1] * vj; col[i] = a; }
      s = fma(a, vk[i], s);
    }
    s = wave_sum(s);
    if (lane == 0) pnext[j] = beta_k * s;
  }
}

// the trailing 2 x 2 block with the last pending update (step n-3), or the whole matrix for n == 2
__global__ void eig_tail(int n, const double* __restrict__ m, double* __restrict__ d, double* __restrict__ e, const double* __restrict__ betas,
    const double* __restrict__ vs, const double* __restrict__ plast) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a11 = m[(size_t)(n - 2) * n + (n - 2)], a21 = m[(size_t)(n - 2) * n + (n - 1)], a22 = m[(size_t)(n - 1) * n + (n - 1)];
  if (n >= 3) {
    const double* v = vs + (size_t)(n - 3) * n;
    const double K = 0.5 * betas[n - 3] * (v[0] * plast[0] + v[1] * plast[1]);
    const double w0 = plast[0] - K * v[0], w1 = plast[1] - K * v[1];
    a11 -= 2.0 * v[0] * w0; a21 -= v[1] * w0 + w1 * v[0]; a22 -= 2.0 * v[1] * w1;
  }
  d[n - 2] = a11; e[n - 2] = a21; d[n - 1] = a22; e[n - 1] = 0.0;
}

__global__ __launch_bounds__(256) void eig_identity(int n, double* __restrict__ q) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < (size_t)n * n) q[idx] = (idx / n == idx % n) ? 1.0 : 0.0;
}

// Q[k+1.., k+1..] <- H_k Q[k+1.., k+1..]: per column, t = beta v.col; col -= t v (one wave per column)
__global__ __launch_bounds__(256) void eig_apply_q(int n, int k, const double* __restrict__ vs, const double* __restrict__ betas,
    double* __restrict__ q) {
  const int len = n - k - 1;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const double bk = betas[k];
  if (j >= len || bk == 0.0) return;
  const double* v = vs + (size_t)k * n;
  double* col = q + (size_t)(k + 1 + j) * n + (k + 1);
  double t = 0.0;
  for (int i = lane; i < len; i += 64) t = fma(v[i], col[i], t);
  t = bk * wave_sum(t);
  for (int i = lane; i < len; i += 64) col[i] -= t * v[i];
}


// ---- blocked replay: the rotations are accumulated into 64 x 64 orthogonal factors, the factors meet Q on the MFMA pipe ------
// Rotation (s, i) — sweep s, columns (i, i + 1) — has to follow (s, i + 1) and (s - 1, i - 1), nothing else.  EIG_B
// consecutive sweeps form a group; inside a group, with j = s mod EIG_B and v = i - j, the rotations with v in one window
// of EIG_B values form a TILE: it only depends on the tile of the next-higher window and on the previous group, and it
// touches the 2 EIG_B = 64 columns [cbase, cbase + 64).  So
//   * eig_form_tiles: one wave per tile applies its <= 1024 rotations to the 64 x 64 identity (a lane carries one row; a
//     sweep's 33 columns sit in registers with compile-time indices, the (c, s) pairs are loaded 32 at a time and handed
//     round with v_readlane) — every tile of the whole replay at once, ~700 independent waves for N = 656;
//   * eig_apply_tiles: one wave per 16 rows of Q keeps its rows in LDS and multiplies the tiles in, in order (groups
//     ascending, windows descending): 64 v_mfma_f64_16x16x4 per tile, the next tile's factor prefetched into registers.
// The one-lane-per-row replay this replaces was a chain of 7e5 dependent rotations on 11 waves: 17 ms at N = 656.
constexpr int EIG_B = 32, EIG_T = 2 * EIG_B, EIG_LO_MARGIN = EIG_B, EIG_HI_MARGIN = EIG_T;
struct EigSweepDev { int first, last; long off; };   // rotations in columns (i, i + 1), i = first .. last descending; (c, s) at cs[2 (off + first - i)]
struct EigTask { int g, cbase; };                   // sweeps g EIG_B .. ; columns cbase .. cbase + 63 (may stick out of [0, n): identity there)

typedef double eig_v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double eig_bcast(double v, int src_lane) {   // src_lane is a compile-time constant: v_readlane_b32 x 2
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void eig_form_tiles(const EigTask* __restrict__ tasks, const EigSweepDev* __restrict__ sweeps, int nsweeps,
    const double* __restrict__ cs, double* __restrict__ ug) {
  __shared__ double ul[EIG_T * (EIG_T + 1)];        // U[row = lane][col] at col * 65 + lane
  const int lane = threadIdx.x;
  const int g = tasks[blockIdx.x].g, cbase = tasks[blockIdx.x].cbase;
#pragma unroll 8
  for (int c = 0; c < EIG_T; ++c) ul[c * (EIG_T + 1) + lane] = (c == lane) ? 1.0 : 0.0;
  for (int j = 0; j < EIG_B; ++j) {
    const int s = g * EIG_B + j;
    if (s >= nsweeps) break;
    const int first = sweeps[s].first, last = sweeps[s].last;
    const long off = sweeps[s].off;
    // rotation u of this sweep's part of the tile: v = EIG_B - 1 - u, column i = cbase + v + j
    if (cbase + j > first || cbase + j + EIG_B - 1 < last) continue;      // (wave-uniform)
    double cv = 1.0, sv = 0.0;
    {
      const int i = cbase + (EIG_B - 1 - lane) + j;
      if (lane < EIG_B && i >= last && i <= first) { const double* p = cs + 2 * (off + first - i); cv = p[0]; sv = p[1]; }
    }
    double x[EIG_B + 1];
#pragma unroll
    for (int t = 0; t <= EIG_B; ++t) x[t] = ul[(j + t) * (EIG_T + 1) + lane];
    double hv = x[EIG_B];
#pragma unroll
    for (int u = 0; u < EIG_B; ++u) {
      const int v = EIG_B - 1 - u;
      const double cc = eig_bcast(cv, u), ss = eig_bcast(sv, u), lv = x[v];
      x[v + 1] = fma(ss, lv, cc * hv);
      hv = fma(cc, lv, -ss * hv);
    }
    x[0] = hv;
#pragma unroll
    for (int t = 0; t <= EIG_B; ++t) ul[(j + t) * (EIG_T + 1) + lane] = x[t];
  }
  double* out = ug + (size_t)blockIdx.x * (EIG_T * EIG_T);           // row-major 64 x 64
#pragma unroll 8
  for (int r = 0; r < EIG_T; ++r) out[r * EIG_T + lane] = ul[lane * (EIG_T + 1) + r];
}

// Q[rows, cbase .. cbase+63] <- Q[rows, same] U for every tile in order.  Computed transposed, D^T = U^T Q^T, so that both the
// operand reads and the result writes of the row slab are 64 consecutive doubles of LDS:
//   A: lane l holds U[k0 + (l >> 4)][c0 + (l & 15)],  B: lane l holds Q[row0 + (l & 15)][cbase + k0 + (l >> 4)],
//   D: lane l, register t holds the new Q[row0 + (l & 15)][cbase + c0 + 4 t + (l >> 4)].
__device__ __forceinline__ void eig_load_u(double (&ua)[4][16], const double* __restrict__ u, int lane) {
  const double* p = u + (lane >> 4) * EIG_T + (lane & 15);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) ua[a][kt] = p[(4 * kt) * EIG_T + 16 * a];
}
__device__ __forceinline__ void eig_apply_one(double* __restrict__ slab, int cbase, const double (&ua)[4][16], int lane) {
  double* base = slab + ((size_t)(cbase + EIG_LO_MARGIN + (lane >> 4))) * 16 + (lane & 15);
  double bq[16];
#pragma unroll
  for (int kt = 0; kt < 16; ++kt) bq[kt] = base[(4 * kt) * 16];
  eig_v4d acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    acc[a] = (eig_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[a][kt], bq[kt], acc[a], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int t = 0; t < 4; ++t) base[(16 * a + 4 * t) * 16] = acc[a][t];
}
__global__ __launch_bounds__(64) void eig_apply_tiles(int n, double* __restrict__ q, const EigTask* __restrict__ tasks, int ntasks,
    const double* __restrict__ ug) {
  extern __shared__ __align__(16) double slab[];   // [EIG_LO_MARGIN + n + EIG_HI_MARGIN columns][16 rows]
  const int lane = threadIdx.x, row0 = blockIdx.x * 16;
  const int r = row0 + (lane & 15);
  const int ncols = EIG_LO_MARGIN + n + EIG_HI_MARGIN;
  for (int c = (lane >> 4); c < ncols; c += 4) {
    const int qc = c - EIG_LO_MARGIN;
    slab[(size_t)c * 16 + (lane & 15)] = (qc >= 0 && qc < n && r < n) ? q[(size_t)qc * n + r] : 0.0;
  }
  double u0[4][16], u1[4][16];
  if (ntasks > 0) eig_load_u(u0, ug, lane);
  int t = 0;
  for (; t + 1 < ntasks; t += 2) {
    eig_load_u(u1, ug + (size_t)(t + 1) * (EIG_T * EIG_T), lane);
    eig_apply_one(slab, tasks[t].cbase, u0, lane);
    eig_load_u(u0, ug + (size_t)(t + 2 < ntasks ? t + 2 : t + 1) * (EIG_T * EIG_T), lane);
    eig_apply_one(slab, tasks[t + 1].cbase, u1, lane);
  }
  if (t < ntasks) eig_apply_one(slab, tasks[t].cbase, u0, lane);
  if (r < n)
    for (int c = (lane >> 4); c < n; c += 4) q[(size_t)c * n + r] = slab[(size_t)(c + EIG_LO_MARGIN) * 16 + (lane & 15)];
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

struct EigWs { double *m, *q, *vs, *d, *e, *betas, *v, *p, *cs, *ug; EigSweep* sw; EigTask* tk; int cap_n; size_t cap_cs, cap_sw, cap_tk, lds_set; double* h_pin; size_t cap_pin; };
static EigWs g_eig = {};

// All eigenpairs of the symmetric n x n matrix a (column-major, ld lda; only the triangle `uplo` is read).
// w: ascending eigenvalues, z (ld ldz): the matching orthonormal eigenvectors (host memory, as GCGE_SymEig).
static long g_symeig_calls = 0;
extern "C" long gcge_hip_symeig_calls(void) { return g_symeig_calls; }   /* how often the device solver ran (tests: only on behalf of the HIP table) */
extern "C" int gcge_hip_symeig(char uplo, int n, const double* a, int lda, double* w, double* z, int ldz) {
  ++g_symeig_calls;
  if (n <= 0) return 0;
  if (n == 1) { w[0] = a[0]; z[0] = 1.0; return 0; }
  if (n > EIG_MAX_N) {   // a 16-row slab of Q no longer fits the LDS of a CU
    std::vector<double> work(2 * (size_t)n);
    return GCGE_SymEigHost(uplo, n, a, lda, w, z, ldz, work.data());
  }
  if (gcge_hip_init(-1) != 0) return -1;
  hipStream_t st = (hipStream_t)gcge_hip_stream();
  EigWs& g = g_eig;
  const size_t nn = (size_t)n * n;
  if (n > g.cap_n) {
    GCGE_HIP_CHECK(hipStreamSynchronize(st));
    if (g.m) { hipFree(g.m); hipFree(g.q); hipFree(g.vs); hipFree(g.d); }
    g.cap_n = n + 64;
    const size_t cn = (size_t)g.cap_n;
    GCGE_HIP_CHECK(hipMalloc(&g.m, cn * cn * sizeof(double)));
    GCGE_HIP_CHECK(hipMalloc(&g.q, cn * cn * sizeof(double)));
    GCGE_HIP_CHECK(hipMalloc(&g.vs, cn * cn * sizeof(double)));
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
    hipLaunchKernelGGL(eig_step, dim3((len + 3) / 4), dim3(256), 0, st, n, k, g.m, g.d, g.e, g.betas, g.vs,
                       (const double*)(k & 1 ? g.v : g.p), k & 1 ? g.p : g.v);
  }
  hipLaunchKernelGGL(eig_tail, dim3(1), dim3(64), 0, st, n, (const double*)g.m, g.d, g.e, (const double*)g.betas, (const double*)g.vs,
                     (const double*)((n - 3) & 1 ? g.p : g.v));
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
    hipLaunchKernelGGL(eig_apply_q, dim3((len + 3) / 4), dim3(256), 0, st, n, k, (const double*)g.vs, (const double*)g.betas, g.q);
  }
  if (timing) GCGE_HIP_CHECK(hipEventRecord(tev[2], st));
  GCGE_HIP_CHECK(hipEventSynchronize(ev_de));
  GCGE_HIP_CHECK(hipEventDestroy(ev_de));
  // 3. QL on the host, rotations recorded
  auto wall = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e3 * ts.tv_sec + 1e-6 * ts.tv_nsec; };
  const double w0 = wall();
  // (the buffers live across calls: a fresh 4 n^2 vector costs ~3 000 page faults per call)
  static std::vector<double> d, e, cs;
  static std::vector<EigSweep> sweeps;
  d.assign(g.h_pin, g.h_pin + n); e.assign(g.h_pin + n, g.h_pin + 2 * n);
  sweeps.clear(); cs.clear();
  if (cs.capacity() < (size_t)4 * n * n) cs.reserve((size_t)4 * n * n);
  const int info = ql_record(n, d.data(), e.data(), sweeps, cs);
  if (info != 0) { GCGE_HIP_CHECK(hipStreamSynchronize(st)); return info; }
  const double w1 = wall();
  // 4. replay on the device: tiles of EIG_B sweeps x EIG_B windows (see eig_form_tiles), groups ascending, windows descending
  size_t n_rot = cs.size() / 2, n_tasks = 0;
  if (!sweeps.empty()) {
    std::vector<EigTask> tasks;
    const int ngroup = (int)((sweeps.size() + EIG_B - 1) / EIG_B);
    for (int gi = 0; gi < ngroup; ++gi) {
      const size_t a = (size_t)gi * EIG_B, bnd = std::min(sweeps.size(), a + EIG_B);
      int hi_c = 0, lo_c = n;
      for (size_t q = a; q < bnd; ++q) { hi_c = std::max(hi_c, sweeps[q].first); lo_c = std::min(lo_c, sweeps[q].last); }
      for (int B = (hi_c - lo_c + EIG_B - 1) / EIG_B; B >= 0; --B) {
        const int cbase = EIG_B * B + lo_c - (EIG_B - 1);
        bool any = false;
        for (size_t q = a; q < bnd && !any; ++q) {
          const int j = (int)(q - a);
          any = !(cbase + j > sweeps[q].first || cbase + j + EIG_B - 1 < sweeps[q].last);
        }
        if (any) tasks.push_back(EigTask{gi, cbase});
      }
    }
    n_tasks = tasks.size();
    const size_t need_cs = cs.size(), need_sw = sweeps.size(), need_tk = tasks.size();
    if (need_cs > g.cap_cs || need_sw > g.cap_sw || need_tk > g.cap_tk) {
      GCGE_HIP_CHECK(hipStreamSynchronize(st));
      if (need_cs > g.cap_cs) { if (g.cs) hipFree(g.cs); g.cap_cs = need_cs * 2; GCGE_HIP_CHECK(hipMalloc(&g.cs, g.cap_cs * sizeof(double))); }
      if (need_sw > g.cap_sw) { if (g.sw) hipFree(g.sw); g.cap_sw = need_sw * 2; GCGE_HIP_CHECK(hipMalloc(&g.sw, g.cap_sw * sizeof(EigSweepDev))); }
      if (need_tk > g.cap_tk) {
        if (g.tk) { hipFree(g.tk); hipFree(g.ug); }
        g.cap_tk = need_tk * 2;
        GCGE_HIP_CHECK(hipMalloc(&g.tk, g.cap_tk * sizeof(EigTask)));
        GCGE_HIP_CHECK(hipMalloc(&g.ug, g.cap_tk * (size_t)(EIG_T * EIG_T) * sizeof(double)));
      }
    }
    static_assert(sizeof(EigSweep) == sizeof(EigSweepDev), "the recorded sweeps go to the device as they are");
    GCGE_HIP_CHECK(hipMemcpyAsync(g.cs, cs.data(), need_cs * sizeof(double), hipMemcpyHostToDevice, st));
    GCGE_HIP_CHECK(hipMemcpyAsync(g.sw, sweeps.data(), need_sw * sizeof(EigSweep), hipMemcpyHostToDevice, st));
    GCGE_HIP_CHECK(hipMemcpyAsync(g.tk, tasks.data(), need_tk * sizeof(EigTask), hipMemcpyHostToDevice, st));
    if (timing) GCGE_HIP_CHECK(hipEventRecord(tev[3], st));
    hipLaunchKernelGGL(eig_form_tiles, dim3((unsigned)need_tk), dim3(64), 0, st, (const EigTask*)g.tk, (const EigSweepDev*)g.sw,
                       (int)need_sw, (const double*)g.cs, g.ug);
    const size_t lds = (size_t)(EIG_LO_MARGIN + n + EIG_HI_MARGIN) * 16 * sizeof(double);
    if (lds > g.lds_set) {
      GCGE_HIP_CHECK(hipFuncSetAttribute((const void*)eig_apply_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      g.lds_set = lds;
    }
    hipLaunchKernelGGL(eig_apply_tiles, dim3((n + 15) / 16), dim3(64), lds, st, n, g.q, (const EigTask*)g.tk, (int)need_tk, (const double*)g.ug);
    if (timing) GCGE_HIP_CHECK(hipEventRecord(tev[4], st));
    GCGE_HIP_CHECK(hipStreamSynchronize(st));          // `tasks` is pageable and leaves scope here
  }
  // 5. back to the host, ascending
  GCGE_HIP_CHECK(hipMemcpyAsync(g.h_pin, g.q, nn * sizeof(double), hipMemcpyDeviceToHost, st));
  GCGE_HIP_CHECK(hipStreamSynchronize(st));          // (also: cs / sweeps are pageable and leave scope below)
  if (timing) {
    float t01 = 0, t12 = 0, t34 = 0;
    hipEventElapsedTime(&t01, tev[0], tev[1]); hipEventElapsedTime(&t12, tev[1], tev[2]);
    if (!sweeps.empty()) hipEventElapsedTime(&t34, tev[3], tev[4]);
    fprintf(stderr, "gcge_hip_symeig n=%d: tridiagonalisation %.2f ms, Q %.2f ms, replay of %zu rotations in %zu sweeps as %zu tiles %.2f ms; host QL %.2f ms, "
            "QL end -> results on the host %.2f ms\n", n, t01, t12, n_rot, sweeps.size(), n_tasks, t34, w1 - w0, wall() - w1);
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
