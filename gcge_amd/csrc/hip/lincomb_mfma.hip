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
// No load in the main loop is predicated (a select or branch next to a load makes hipcc wait with
// vmcnt(0) after every single load: measured 23 TF): rows beyond nrows and columns beyond k are
// CLAMPED to the last valid one, and the coefficient block is first copied into a zero-padded
// (k rounded to 32) x (16 NT) workspace, so a clamped X column meets a zero coefficient row and a
// clamped X row only feeds output rows that are never stored.
//
// In-place use is safe, with disjoint column ranges (ops_orth.c:70,90,253) and with the output columns INSIDE the
// input range (X = X R^-1, P = V[:, N..W) coef; declared to the solver stack by GCGE_SetInplaceLinearComb): a block /
// wave reads only the rows it writes (rows past the end are clamped to the last row, but feed output rows that are
// never stored) and stores them after its last read of them; x is therefore NOT __restrict__.
// Roofline: 2 n k m flops (FP64 MFMA) vs 8 n (k + m [+ m]) bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gcge_hip_internal.h"

namespace gcge {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int LC_KT = 32;          // k-tile
constexpr int LC_XS = LC_KT + 2;   // LDS row stride of the X tile (doubles)

// NT 16-column output fragments per wave: m <= 16 NT.  RF 16-row fragments per wave: a block owns 64 RF rows and every
// coefficient fragment read from LDS feeds RF MFMAs (RF = 2: 8 MFMAs per 6 LDS reads at m = 64 instead of 4 per 5; PMC
// showed the RF = 1 form issuing MFMAs at 52-62 % of the pipe rate, profiles/r01_dense/10).
template <int NT, int RF>
__global__ __launch_bounds__(256, (RF == 2 && NT <= 4) ? 2 : 1) void lincomb_kernel(long nrows, const double* x, long ldx, int k,
    const double* __restrict__ cpad, int m, const double* __restrict__ beta, double* y, long ldy, int cs) {
  extern __shared__ __align__(16) double lds[];
  constexpr int BR = 64 * RF;       // rows per block
  double* xs = lds;                 // [BR][LC_XS]
  double* cst = lds + BR * LC_XS;   // [LC_KT][cs]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const long r0 = (long)blockIdx.x * BR;

  v4d acc[RF][NT];
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[f][t] = (v4d){0.0, 0.0, 0.0, 0.0};

  // register double-buffering: the next k-tile is fetched from global memory while the MFMAs of the
  // current one run; it is written to LDS after the barrier that ends the current tile
  constexpr int XE = BR * LC_KT / 256;            // X elements per thread and tile (8 RF)
  constexpr int CE = LC_KT * 16 * NT / 256;       // C elements per thread and tile (2 NT)
  double xr[XE], cr[CE];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < CE; ++q) {
      const int e = threadIdx.x + 256 * q, row = e / (16 * NT), col = e % (16 * NT);
      cr[q] = cpad[(long)(k0 + row) * (16 * NT) + col];
    }
#pragma unroll
    for (int q = 0; q < XE; ++q) {
      const int e = threadIdx.x + 256 * q, row = e / LC_KT, col = e % LC_KT;
      const long gr = min(r0 + row, nrows - 1);
      const int gc = min(k0 + col, k - 1);
      xr[q] = x[gr * ldx + gc];
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int q = 0; q < XE; ++q) { const int e = threadIdx.x + 256 * q; xs[(e / LC_KT) * LC_XS + e % LC_KT] = xr[q]; }
#pragma unroll
    for (int q = 0; q < CE; ++q) {
      const int e = threadIdx.x + 256 * q, row = e / (16 * NT), col = e % (16 * NT);
      cst[row * cs + col] = cr[q];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < k; k0 += LC_KT) {
    stash();
    __syncthreads();
    if (k0 + LC_KT < k) fetch(k0 + LC_KT);
#pragma unroll
    for (int s = 0; s < LC_KT; s += 4) {
      double a[RF];
#pragma unroll
      for (int f = 0; f < RF; ++f) a[f] = xs[(16 * RF * wave + 16 * f + li) * LC_XS + s + kk];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double b = cst[(s + kk) * cs + 16 * t + li];
#pragma unroll
        for (int f = 0; f < RF; ++f) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[f], b, acc[f][t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // epilogue: Y = acc + beta_j * Y.  The old Y values are fetched with clamped (row, column) and no
  // predicate so that all 4 NT loads of a lane are in flight together; only the stores are guarded.
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    const long rw = r0 + 16 * RF * wave + 16 * f;     // first row of this fragment
    if (beta != nullptr) {
      double yv[NT][4], bj[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int colc = min(16 * t + li, m - 1);
        bj[t] = beta[colc];
#pragma unroll
        for (int u = 0; u < 4; ++u) yv[t][u] = y[min(rw + 4 * u + kk, nrows - 1) * ldy + colc];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[f][t][u] = fma(bj[t], yv[t][u], acc[f][t][u]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 16 * t + li;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long row = rw + 4 * u + kk;
        if (col < m && row < nrows) y[row * ldy + col] = acc[f][t][u];
      }
    }
  }
}


// ---- direct form: A fragments straight from global memory, accumulators pinned to AGPRs ------------------------------
// The X tile does not go through LDS at all.  Lane (li = l & 15, kk = l >> 4) of a wave loads 16 bytes
//     X[r + li][k0 + 8 j + 2 kk + {0,1}],  j = 0..3:   four loads cover the 32 columns of a k-tile for 16 rows,
// and because the k index of an MFMA operand is just a summation index, the two halves of such a load feed two
// MFMAs whose B fragments are the matching coefficient rows 8 j + 2 kk + h (h = 0, 1) — any assignment of the 32
// k values of a tile to (j, kk, h) is a valid one.  What this removes against lincomb_kernel above: the register ->
// LDS -> register round trip of X (16 ds_write_b64 + 2 barriers per tile) and the 8-byte global loads (an 8-byte lane
// load costs the address path as much as a 16-byte one).  Only the coefficient tile (32 x 16 NT doubles, L2-resident)
// is staged through LDS, double-buffered: one barrier per k-tile.  A wave owns RF = 2 row fragments (32 rows) and all
// NT column fragments; the 2 NT accumulator tiles live in a[0 : 16 NT) by name (agpr_tiles.inc), so no v_accvgpr
// shuttling.  k odd: the pair (k-1, k) would read one column past the operand; the last k-tile therefore blends the
// second half of its loads with 0 by an integer mask (after the loads have been issued: no predicate near them).
#include "agpr_tiles.inc"
typedef double v2d_lc __attribute__((ext_vector_type(2)));

template <int NT, int RF, int MINB, int KTD, bool ASM = true>
__global__ __launch_bounds__(256, MINB) void lincomb_direct_kernel(long nrows, const double* x, long ldx, int k,
    const double* __restrict__ cpad, int m, const double* __restrict__ beta, double* y, long ldy, int cs) {
  extern __shared__ __align__(16) double lds[];       // [2][KTD][cs]
  constexpr int NJ = KTD / 8;                          // 16-byte loads per row fragment and k-tile
  constexpr int CE = (KTD * 16 * NT + 255) / 256;      // coefficient elements per thread and tile (NT odd at k-tiles of 8: the last round is half
  constexpr bool CE_RAGGED = (KTD * 16 * NT) % 256 != 0;   //  used — its loads run into the next tile's rows, which the workspace holds; only the LDS stores are guarded)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int li = lane & 15, kk = lane >> 4;
  const long r0 = (long)blockIdx.x * (64 * RF) + 16 * RF * wave;
  // Accumulators.  NT >= 4: 8 or 16 tiles pinned to AGPRs by name and driven by inline-asm MFMAs; a tile is then
  // touched again only 7 or 15 MFMAs later.  NT <= 2 has 2 or 4 tiles, i.e. dependent MFMAs one or three instructions
  // apart: the hardware does NOT interlock a DGEMM MFMA reading SrcC against the previous one still writing it (the
  // compiler's hazard recogniser inserts the wait states for the builtin, but it does not see inline asm — measured:
  // results with stale low/high words).  Those widths are bandwidth-bound anyway and take the builtin.
  constexpr bool PIN = ASM && RF * NT >= 8;
  v4d accv[PIN ? 1 : RF][PIN ? 1 : NT];
  if constexpr (PIN) {
#pragma unroll
    for (int T = 0; T < RF * NT; ++T) agpr_tile_zero(T);
  } else {
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int t = 0; t < NT; ++t) accv[f][t] = (v4d){0.0, 0.0, 0.0, 0.0};
  }

  // per-lane row bases (clamped: rows past the end feed output rows that are never stored)
  const double* xr[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) xr[f] = x + min(r0 + 16 * f + li, nrows - 1) * ldx + 2 * kk;
  const int kpairs = (k + 1) / 2;                      // 16-byte column pairs that hold at least one valid column
  auto fetch_a = [&](v2d_lc (&a)[RF][NJ], int k0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int pr = min((k0 + 8 * j) / 2 + kk, kpairs - 1) - kk;   // pair index, clamped into the operand (meets a zero coefficient row)
#pragma unroll
      for (int f = 0; f < RF; ++f) a[f][j] = *reinterpret_cast<const v2d_lc*>(xr[f] + 2 * pr);
    }
  };
  double cr[CE];
  auto fetch_c = [&](int k0) {
#pragma unroll
    for (int q = 0; q < CE; ++q) {
      const int e = threadIdx.x + 256 * q, row = e / (16 * NT), col = e % (16 * NT);
      cr[q] = cpad[(long)(k0 + row) * (16 * NT) + col];
    }
  };
  auto stash_c = [&](int buf) {
    double* cst = lds + buf * KTD * cs;
#pragma unroll
    for (int q = 0; q < CE; ++q) {
      const int e = threadIdx.x + 256 * q, row = e / (16 * NT), col = e % (16 * NT);
      if (!CE_RAGGED || e < KTD * 16 * NT) cst[row * cs + col] = cr[q];
    }
  };
  auto mfmas = [&](const v2d_lc (&a)[RF][NJ], int buf) {
    const double* cst = lds + buf * KTD * cs + li;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j > 0) __builtin_amdgcn_sched_barrier(0);   // keeps the coefficient reads of later groups from being hoisted (registers)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const double b = cst[(8 * j + 2 * kk + h) * cs + 16 * t];
#pragma unroll
          for (int f = 0; f < RF; ++f) {
            if constexpr (PIN) agpr_tile_mfma(f * NT + t, h ? a[f][j].y : a[f][j].x, b);
            else accv[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(h ? a[f][j].y : a[f][j].x, b, accv[f][t], 0, 0, 0);
          }
        }
    }
  };
  const int ntile = (k + KTD - 1) / KTD;
  v2d_lc a0[RF][NJ], a1[RF][NJ];
  fetch_c(0);
  fetch_a(a0, 0);
  stash_c(0);
  __syncthreads();
  // tiles in pairs with two register sets (no copy, no branch inside); the last tile is peeled for the odd-k blend
  int tl = 0;
  for (; tl + 2 < ntile; tl += 2) {
    fetch_a(a1, (tl + 1) * KTD); fetch_c((tl + 1) * KTD);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stash_c(1);
    __syncthreads();
    fetch_a(a0, (tl + 2) * KTD); fetch_c((tl + 2) * KTD);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a1, 1);
    __builtin_amdgcn_sched_barrier(0);
    stash_c(0);
    __syncthreads();
  }
  // here: a0 / LDS buffer 0 hold tile tl; 1 or 2 tiles remain
  const long odd_mask = (k & 1) ? 0L : -1L;            // all ones: keep the second half of the last valid pair
  auto blend_last = [&](v2d_lc (&a)[RF][NJ], int k0) {   // zero x[.., k] where the pair (k-1, k) straddles the end
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool straddles = 2 * ((k0 + 8 * j) / 2 + kk) + 1 >= k;
      const long keep = straddles ? odd_mask : -1L;
#pragma unroll
      for (int f = 0; f < RF; ++f) a[f][j].y = __longlong_as_double(__double_as_longlong(a[f][j].y) & keep);
    }
  };
  if (tl + 2 == ntile) {
    fetch_a(a1, (tl + 1) * KTD); fetch_c((tl + 1) * KTD);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(a0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stash_c(1);
    __syncthreads();
    blend_last(a1, (tl + 1) * KTD);
    mfmas(a1, 1);
  } else {
    blend_last(a0, tl * KTD);
    mfmas(a0, 0);
  }
  // the MFMAs are inline asm, invisible to the hazard recogniser: let the last ones retire before the tiles are read
  if constexpr (PIN) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    const long rw = r0 + 16 * f;
    double acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if constexpr (PIN) acc[t][u] = agpr_tile_read(f * NT + t, u);
        else acc[t][u] = accv[f][t][u];
      }
    if (beta != nullptr) {
      double yv[NT][4], bj[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int colc = min(16 * t + li, m - 1);
        bj[t] = beta[colc];
#pragma unroll
        for (int u = 0; u < 4; ++u) yv[t][u] = y[min(rw + 4 * u + kk, nrows - 1) * ldy + colc];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = fma(bj[t], yv[t][u], acc[t][u]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 16 * t + li;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long row = rw + 4 * u + kk;
        if (col < m && row < nrows) y[row * ldy + col] = acc[t][u];
      }
    }
  }
}

}  // namespace gcge

// cpad (kp x mp, row-major, zero outside rows [shift, shift + k) x columns [0, m)) <- c (k x m, row-major)
__global__ __launch_bounds__(256) void lincomb_pad_c(const double* __restrict__ c, int k, int m, double* __restrict__ cpad,
                                                     int kp, int mp, int shift) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= kp * mp) return;
  const int row = e / mp - shift, col = e % mp;
  cpad[e] = (row >= 0 && row < k && col < m) ? c[(long)row * m + col] : 0.0;
}

using namespace gcge;

static double* g_cpad = nullptr;
static size_t g_cpad_len = 0;
static int g_lc_rf = 0;   // 0 automatic (the direct form for panels of >= 33 columns where the operand allows 16-byte loads);
                          // 1 / 2: the LDS-staged kernel with that many row fragments per wave; >= 3: the direct form forced, see lc_launch
static int g_lc_wide_only = 0;   // 1: panels of 65 .. 96 columns take the 128-column kernel as before round 4 (measurements)
extern "C" void gcge_hip_lincomb_wide_only(int on) { g_lc_wide_only = on != 0; }
extern "C" void gcge_hip_lincomb_tune(int row_fragments) { if (row_fragments >= 0 && row_fragments <= 14) g_lc_rf = row_fragments; }

template <int NT>
static int lc_launch(int nrows, const double* x, long ldx, int k, const double* c, int m,
                     const double* beta, double* y, long ldy, hipStream_t st) {
  // An operand that starts on an odd column of a 16-byte aligned block (the solver's X / P / W ranges move with the
  // number of converged pairs) is widened by the column in front of it, which meets a zero coefficient row: the direct
  // form's 16-byte loads stay aligned (the staged kernel it fell back to: 25.0 against 19.5 ms at k = 192, m = 128).
  // That column belongs to the same block of vectors (blocks are zero-filled at creation: finite).
  int shift = 0;
  if ((((uintptr_t)x & 15) == 8) && (ldx % 2 == 0) && ((g_lc_rf == 0 && NT >= 4) || g_lc_rf >= 3)) { shift = 1; x -= 1; k += 1; }
  const int kp = (k + LC_KT - 1) / LC_KT * LC_KT, mp = 16 * NT;
  if ((size_t)(kp + LC_KT) * mp > g_cpad_len) {   // grows rarely; freeing synchronises with kernels still reading the old one (one k-tile of slack: see CE_RAGGED)
    if (g_cpad) GCGE_HIP_CHECK(hipFree(g_cpad));
    g_cpad_len = (size_t)(kp + LC_KT) * mp * 2;
    GCGE_HIP_CHECK(hipMalloc(&g_cpad, g_cpad_len * sizeof(double)));
  }
  hipLaunchKernelGGL(lincomb_pad_c, dim3((kp * mp + 255) / 256), dim3(256), 0, st, c, k - shift, m, g_cpad, kp, mp, shift);
  // direct form: X read with 16-byte lane loads straight into MFMA operands (needs a 16-byte aligned operand)
  // Measured at n = 2^24 (profiles/r02_dense/06): what decides is waves per SIMD, and short k-tiles buy them — the operand
  // registers of a k-tile shrink with it while the accumulator tiles stay.  k = 256, m = 128: k-tiles of 32 at one wave per
  // SIMD 47.2 TF, of 16 at two waves 56.2, of 8 at two waves 57.9 (staged form 45.9; the vendor GEMM 63.8); m = 64:
  // k-tiles of 32 at two waves 46.4, of 16 at two waves 50.9, of 8 at three waves with compiler-managed accumulators 53.2
  // (staged 49.4; vendor 30.5).  Panels of <= 32 columns are bandwidth-bound and stay with the staged kernel.
  // Accumulators pinned to AGPRs BY NAME are only used where the compiler is not short of registers: the clobber lists
  // keep it from holding values in a named tile ACROSS a statement that uses it, not from parking a temporary there
  // BETWEEN two such statements — the m = 64 variants at three / four waves per SIMD (84 / 64 VGPRs) did exactly that and
  // returned wrong panels (caught by test_lincomb_row_fragment_variants_vs_oracle at k = 192, m = 64).  Those widths take
  // the builtin MFMA (the compiler owns the accumulators); m = 128 at two waves per SIMD keeps the named tiles and is
  // covered by the same test with all 16 tiles live.
  if (((g_lc_rf == 0 && NT >= 4) || g_lc_rf >= 3) && (((uintptr_t)x & 15) == 0) && (ldx % 2 == 0) && k >= 2) {
    const int csd = 16 * NT + 8;   // coefficient rows 2 apart land on the other half of the 64 LDS banks
#define GCGE_LCD(RFV, MB, KTV, ASMV) hipLaunchKernelGGL((lincomb_direct_kernel<NT, RFV, MB, KTV, ASMV>), dim3((unsigned)(((long)nrows + 64 * RFV - 1) / (64 * RFV))), dim3(256), \
                                             (size_t)2 * KTV * csd * sizeof(double), st, (long)nrows, x, ldx, k, g_cpad, m, beta, y, ldy, csd)
    // tuning codes (gcge_hip_lincomb_tune / GCGE_LINCOMB_RF): 3 the automatic direct form also for narrow panels;
    // 7: k-tiles of 32 (named tiles); 10: k-tiles of 16 at two waves per SIMD (named tiles); 13 / 14: builtin, k-tiles of 8 / 16
    if constexpr (NT == 8) {
      if (g_lc_rf == 7) GCGE_LCD(2, 1, 32, true); else if (g_lc_rf == 10) GCGE_LCD(2, 2, 16, true);
      else if (g_lc_rf == 13) GCGE_LCD(2, 2, 8, false); else if (g_lc_rf == 14) GCGE_LCD(2, 2, 16, false);
      else GCGE_LCD(2, 2, 8, true);
    } else if constexpr (NT == 4) {
      if (g_lc_rf == 7) GCGE_LCD(2, 2, 32, true); else if (g_lc_rf == 10) GCGE_LCD(2, 2, 16, true);
      else if (g_lc_rf == 14) GCGE_LCD(2, 3, 16, false);
      else GCGE_LCD(2, 3, 8, false);
    } else if constexpr (NT >= 5) {   // panels of 65 .. 96 columns (round 4): 5 or 6 column fragments instead of 8 with up to 3/8 of the MFMAs on zero
      // columns — k = 256: m = 72 35.1 -> 47.5 TF, 80 39.0 -> 51.8, 96 46.2 -> 56.1 (profiles/r04_bench); builtin MFMAs only (the named
      // tiles were written for 8 and 16 of them).  Seven fragments (m <= 112) measured no faster than eight: those panels stay there.
      if (g_lc_rf == 14) GCGE_LCD(2, 2, 16, false); else GCGE_LCD(2, 2, 8, false);
    } else {
      GCGE_LCD(2, 2, 16, false);
    }
#undef GCGE_LCD
    return 0;
  }
  const int cs = (16 * NT + 31) / 32 * 32 + 16;  // row stride of the C tile: 16 mod 32 doubles
  // two row fragments per wave once there are enough rows to fill the chip with 128-row blocks several times over
  // and enough MFMA work per tile to pay for the larger register set (n = 2^24, k = 256: m = 128 27.1 -> 23.9 ms =
  // 46 TF, m = 64 12.0 -> 11.3 ms = 48.5 TF with two waves per SIMD; k = 64 and narrower panels: no gain)
  const int rf = (g_lc_rf == 0 || g_lc_rf >= 3) ? (((NT == 8 || (NT == 4 && k >= 128)) && (long)nrows >= 128L * 256 * 8) ? 2 : 1) : g_lc_rf;
  const size_t shmem = (size_t)(64 * rf * LC_XS + LC_KT * cs) * sizeof(double);
  const unsigned grid = (unsigned)(((long)nrows + 64 * rf - 1) / (64 * rf));
  if (rf == 2)
    hipLaunchKernelGGL((lincomb_kernel<NT, 2>), dim3(grid), dim3(256), shmem, st, (long)nrows, x, ldx, k, g_cpad, m, beta,
                       y, ldy, cs);
  else
    hipLaunchKernelGGL((lincomb_kernel<NT, 1>), dim3(grid), dim3(256), shmem, st, (long)nrows, x, ldx, k, g_cpad, m, beta,
                       y, ldy, cs);
  return 0;
}

// d_c: row-major k x m coefficient block on the device; d_beta: m scale factors or NULL
extern "C" int gcge_hip_lincomb(int nrows, const double* d_x, long ldx, int k, const double* d_c, int m,
                                const double* d_beta, double* d_y, long ldy, void* stream) {
  gcge_hip_apply_pending();
  if (nrows <= 0 || m <= 0 || k <= 0) return 0;
  if (m > 128) return -2;  // callers split wider panels
  hipStream_t st = (hipStream_t)stream;
  if (m <= 16) lc_launch<1>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  else if (m <= 32) lc_launch<2>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  else if (m <= 64) lc_launch<4>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  else if (m <= 80 && g_lc_wide_only == 0) lc_launch<5>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  else if (m <= 96 && g_lc_wide_only == 0) lc_launch<6>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  else lc_launch<8>(nrows, d_x, ldx, k, d_c, m, d_beta, d_y, ldy, st);
  return (int)hipGetLastError();
}
