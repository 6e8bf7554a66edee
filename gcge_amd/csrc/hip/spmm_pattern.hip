// K1 (pattern path) — SpMM for matrices whose rows repeat a small set of stencils.
//
// Same contract as spmm.hip / spmm_pad8.hip (Y[:, y0:y0+m) = A X[:, x0:x0+m), reference
// app/app_ccs.c:50-139).  The finite-difference and finite-element matrices the reference is
// exercised with (test_app_ccs.c:33-83 3-D Laplacian, the cube4 P1 pair) have only a few dozen
// distinct rows when a row is written as {(column - row, value)}.  The back-end detects that
// at matrix creation (app_hip.hip: build_patterns) and keeps
//     pid[r]            (16 bit)      which pattern row r follows
//     tab[p][0..LT)     (16 bytes)    (value, column offset), padded with (0.0, 0)
// so the matrix stream shrinks from 12 B per non-zero to 2 B per ROW and the (col,val)
// loads and their cross-lane broadcasts disappear (measured upper bound of this format in
// profiles/r01_spmm_explore: 4.0-4.2 ms against 5.4-5.9 ms for the generic kernels at 256^3 x 64).
// Matrices without such structure keep the pad-8 path.
//
// Kernel shape (what the counters asked for, profiles/r01_spmm_explore/README.md):
//   * passes of 16 columns: a grid plane of X (N^2 rows x 128 B) then fits the L2s, so the +-N^2
//     neighbours are L2 hits instead of second and third trips over the fabric;
//   * lane l -> row slot g = l >> 3 of an 8-row slice, column pair i = l & 7: one 16-byte load
//     per lane fetches the 128-byte X segments of 8 rows; no cross-lane reduction at all;
//   * a block works on TILES of 4 slices, one per wave: 4 consecutive slices by default (32-row
//     chunks stay interleaved over the XCDs like a one-chunk-per-block launch); optionally `line`
//     rows apart (the second longest stencil offset) so that the +-N neighbours of one wave are
//     the centre rows of the next — measured, no gain, kept as a tuning hook;
//   * a block walks its tiles (tile b, b + G, ...) with two register sets: the X rows of the NEXT
//     tile are requested before the current one is reduced, pattern ids two tiles ahead;
//   * no load sits next to a branch or select (hipcc would wait with vmcnt(0) after each);
//     surplus iterations re-do the wave's last slice and only their stores are predicated.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "gcge_hip_internal.h"

extern "C" double* gcge_hip_partial_ws(size_t len);
extern "C" void gcge_hip_reduce_partials(const double* d_partial, int nblocks, int len, double* d_out, void* stream);
extern "C" void gcge_hip_reduce_partials16(const double* d_partial, int nblocks, long slab_stride, int ncols, double* d_out,
                                           void* stream);
extern "C" void gcge_hip_reduce_partials_slabs(const double* d_partial, int nblocks, long slab_stride, int cpp, int ncols,
                                               double* d_out, void* stream);
extern "C" int gcge_hip_ring_pass(int mode, int nrows, const unsigned short* d_pid, const void* d_tab, int npat, long L, int nw,
                                  long nb, const double* d_x, long ldx, int m, double* part, long yyo,
                                  const double* d_lambda, void* stream, long maxoff, double* d_y, long ldy, int gy);

namespace gcge {

typedef double v2d __attribute__((ext_vector_type(2)));
struct PatEntry { double val; long off; };   // 16 bytes: one ds_read_b128

// MODE (the kernels below share it):
//   0  Y = A X
//   1  Y = A X and dot_partial[block][j] = sum over the block's rows of X[r,j] * Y[r,j] (+ Y[r,j]^2 behind yy_offset)
//   2  the two sums of mode 1 only, nothing is stored (first pass of the block CG, block_pcg.hip)
//   3  second pass of the block CG: w = A X is recomputed in registers and consumed on the spot,
//        R[r,j] -= alpha_j w[r,j] ;  PNEW[r,j] = cr_j R[r,j] + cb_j X[r,j] ;  partial: sum_r cr_j R[r,j]^2
//      with (alpha, cb, cr) = flag_j ? (alpha_j, beta_j, 1) : (0, 1, 0) exactly as cg_update_rp (retired columns are
//      copied).  X = p_k, PNEW = p_{k+1} must be different blocks: neighbours still read X.
//   7  mode 3 without a stored residual: the block passed as R holds p_{k-1}, cg.b the previous iteration's beta, and
//      r_k = p_k - beta_{k-1} p_{k-1} is rebuilt on the spot (p_k = r_k + beta_{k-1} p_{k-1} is how p_k was formed); only
//      PNEW is written: 3 block streams instead of 4.  Retired columns (flag 0) are copied as in mode 3.
//   5  start of the block CG: R[r,j] = B[r,j] - (A X)[r,j], PNEW = R (p_0 = r_0), partial: sum_r R[r,j]^2, with
//      B = cg.b the right-hand sides and X the initial guess: one sweep instead of product, axpby, column dots and copy
//   6  mode 5 with the right-hand side B = X diag(scale) (scale = cg.alpha) formed on the fly from the row's own X value:
//      the GCG driver's systems A w = (lambda + sigma) x start from w = x, so neither B nor a second read is needed
//   4  residual norms of Ritz pairs (standard problem): partial: sum_r ((A X)[r,j] - lambda_j X[r,j])^2 with
//      lambda = cg.alpha; nothing is stored (CheckConvergence of the GCG driver, one read of X instead of 11 streams)
struct CgArgs { double* r; size_t ldr; double* pnew; size_t ldp; const double* alpha; const double* beta; const int* flag; const double* b; size_t ldb;
                const double* rowval; };   // rowval != NULL (kernels built with VALS): the table holds offsets only, the values of row r are rowval[8 r + slot]
struct CgCoef { double al0, al1, cb0, cb1, cr0, cr1, bp0, bp1; };
__device__ __forceinline__ CgCoef cg_coef(const CgArgs& cg, int j, bool act) {
  CgCoef c = {0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0};
  if (act) {
    const int f0 = cg.flag[j], f1 = cg.flag[j + 1];
    if (f0) { c.al0 = cg.alpha[j]; c.cb0 = cg.beta[j]; c.cr0 = 1.0; }
    if (f1) { c.al1 = cg.alpha[j + 1]; c.cb1 = cg.beta[j + 1]; c.cr1 = 1.0; }
  }
  return c;
}

// VALS: lane (g, i) of an 8-lane row group holds the value of table slot i of its row; slot t for the whole group
__device__ __forceinline__ double group_bcast(double v, int lane, int t) {
  const int src = (lane & ~7) | t;
  return __hiloint2double(__shfl(__double2hiint(v), src, 64), __shfl(__double2loint(v), src, 64));
}

template <int LT, int MODE, bool VALS = false>
__global__ __launch_bounds__(256) void spmm_pattern_kernel(
    long nrows, const unsigned short* __restrict__ pid, const PatEntry* __restrict__ tab, int ntab,
    const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int m, long ntiles, long line,
    double* __restrict__ dot_partial, long yy_offset, CgArgs cg) {
  constexpr int DOT = MODE != 0;       // the row's own X value rides in buf[LT]
  constexpr int UPD = MODE == 3 || MODE == 5 || MODE == 7;   // its R value (MODE 5: right-hand side, MODE 7: p_{k-1}) in buf[LT + 1]
  constexpr int RES = MODE == 4;
  if (gridDim.y > 1) {   // the 16-column passes of one operation in one launch (see spmm_pattern_chain2_kernel): pass = blockIdx.y
    const int c0 = 16 * (int)blockIdx.y;
    x += c0; if (y != nullptr) y += c0;
    if (dot_partial != nullptr) dot_partial += (size_t)blockIdx.y * gridDim.x * 16;
    if (cg.r != nullptr) cg.r += c0;
    if (cg.pnew != nullptr) cg.pnew += c0;
    if (cg.alpha != nullptr) cg.alpha += c0;
    if (cg.beta != nullptr) cg.beta += c0;
    if (cg.flag != nullptr) cg.flag += c0;
    if (cg.b != nullptr) cg.b += c0;
    m = min(m - c0, 16);
  }
  extern __shared__ __align__(16) unsigned char smem_raw[];
  PatEntry* s_tab = reinterpret_cast<PatEntry*>(smem_raw);
  for (int e = threadIdx.x; e < ntab; e += 256) s_tab[e] = tab[e];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 3, i = lane & 7;
  const bool act = 2 * i < m;
  const double* __restrict__ xl = x + (act ? 2 * i : 0);   // idle lanes of a narrow last pass re-read column 0
  const double* __restrict__ rl = UPD ? (MODE == 5 ? cg.b : cg.r) + (act ? 2 * i : 0) : nullptr;
  const size_t ldrl = MODE == 5 ? cg.ldb : cg.ldr;
  CgCoef cf = (MODE == 3 || MODE == 7) ? cg_coef(cg, 2 * i, act) : CgCoef{0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0};
  if (MODE == 7 && act) { cf.bp0 = cg.b[2 * i]; cf.bp1 = cg.b[2 * i + 1]; }   // beta of the previous iteration
  if ((RES || MODE == 6) && act) { cf.al0 = cg.alpha[2 * i]; cf.al1 = cg.alpha[2 * i + 1]; }   // lambda / rhs scale of this lane's column pair
  double d0 = 0.0, d1 = 0.0, e0 = 0.0, e1 = 0.0;   // x.y and y.y column sums (DOT)

  // tile t = (group of 4 lines q, slice a inside the line); wave w takes line 4q + w
  const long G = gridDim.x, aslices = line / 8;
  auto first_row = [&](long t) { const long q = t / aslices, a = t - q * aslices; return ((4 * q + wave) * line) + 8 * a; };
  if ((long)blockIdx.x < ntiles) {   // block-uniform
    const long cnt = (ntiles - blockIdx.x + G - 1) / G;
    auto row_of = [&](long it) { return min(first_row(blockIdx.x + min(it, cnt - 1) * G) + g, nrows - 1); };   // clamped
    auto issue = [&](v2d (&buf)[LT + DOT + UPD], double (&val)[LT], long row, int p) {
      if (VALS) val[0] = cg.rowval[(size_t)row * 8 + i];   // my slot's value; finish() hands the slots round the row group
#pragma unroll
      for (int t = 0; t < LT; ++t) {
        const PatEntry e = s_tab[p * LT + t];
        if (!VALS) val[t] = e.val;
        buf[t] = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e.off) * ldx);
      }
      if (DOT) buf[LT] = *reinterpret_cast<const v2d*>(xl + (size_t)row * ldx);
      if (UPD) buf[LT + DOT] = *reinterpret_cast<const v2d*>(rl + (size_t)row * ldrl);
    };
    auto finish = [&](const v2d (&buf)[LT + DOT + UPD], const double (&val)[LT], long it) {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int t = 0; t < LT; ++t) {
        const double vt = VALS ? group_bcast(val[0], lane, t) : val[t];
        a0 = fma(vt, buf[t].x, a0); a1 = fma(vt, buf[t].y, a1);
      }
      const long row = first_row(blockIdx.x + it * G) + g;   // unclamped: surplus iterations and tail rows store nothing
      const bool ok = it < cnt && row < nrows && act;
      if ((MODE <= 1 && ok) || ((MODE == 2 || RES) && ok && y != nullptr)) {   // MODE 2, 4: y == NULL, see chain2_body
        v2d o = {a0, a1};
        __builtin_nontemporal_store(o, reinterpret_cast<v2d*>(y + (size_t)row * ldy + 2 * i));
      }
      const double wgt = ok ? 1.0 : 0.0;
      if (MODE == 1 || MODE == 2) {
        d0 = fma(a0 * wgt, buf[LT].x, d0); d1 = fma(a1 * wgt, buf[LT].y, d1);
        e0 = fma(a0 * wgt, a0, e0); e1 = fma(a1 * wgt, a1, e1);
      }
      if (MODE == 5 || MODE == 6) {
        // MODE 6: the product is rounded on its own (as the column scaling that used to form B did), then subtracted
        v2d rv = buf[LT + DOT + UPD - 1];
        if (MODE == 6) {
#pragma clang fp contract(off)   // no fma(scale, x, -Ax): __dmul_rn is a plain product in the HIP headers and would be contracted
          rv = v2d{cf.al0 * buf[LT].x, cf.al1 * buf[LT].y};
        }
        v2d rn = {rv.x - a0, rv.y - a1};
        if (ok) {
          __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(cg.r + (size_t)row * cg.ldr + 2 * i));
          if (cg.pnew != cg.r) __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(cg.pnew + (size_t)row * cg.ldp + 2 * i));   // (equal: the residual alone, a V-cycle's r = b - A x)
        }
        d0 = fma(wgt * rn.x, rn.x, d0); d1 = fma(wgt * rn.y, rn.y, d1);
      }
      if (MODE == 3 || MODE == 7) {
        const v2d pv = buf[LT];
        v2d rv = buf[LT + DOT];
        if (MODE == 7) rv = v2d{fma(-cf.bp0, rv.x, pv.x), fma(-cf.bp1, rv.y, pv.y)};   // r_k = p_k - beta_{k-1} p_{k-1}
        v2d rn = {fma(-cf.al0, a0, rv.x), fma(-cf.al1, a1, rv.y)};
        v2d pn = {fma(cf.cb0, pv.x, cf.cr0 * rn.x), fma(cf.cb1, pv.y, cf.cr1 * rn.y)};
        if (ok) {
          if (MODE == 3) __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(cg.r + (size_t)row * cg.ldr + 2 * i));
          __builtin_nontemporal_store(pn, reinterpret_cast<v2d*>(cg.pnew + (size_t)row * cg.ldp + 2 * i));
        }
        d0 = fma(cf.cr0 * wgt * rn.x, rn.x, d0); d1 = fma(cf.cr1 * wgt * rn.y, rn.y, d1);
      }
      if (RES) {
        const double q0 = fma(-cf.al0, buf[LT].x, a0), q1 = fma(-cf.al1, buf[LT].y, a1);
        d0 = fma(q0 * wgt, q0, d0); d1 = fma(q1 * wgt, q1, d1);
      }
    };
    v2d b0[LT + DOT + UPD], b1[LT + DOT + UPD];
    double v0[LT], v1[LT];
    int p0 = pid[row_of(0)], p1 = pid[row_of(1)];
    issue(b0, v0, row_of(0), p0);
    for (long it = 0; it < cnt; it += 2) {
      const int p2 = pid[row_of(it + 2)];
      issue(b1, v1, row_of(it + 1), p1);
      __builtin_amdgcn_sched_barrier(0);
      finish(b0, v0, it);
      __builtin_amdgcn_sched_barrier(0);
      const int p3 = pid[row_of(it + 3)];
      issue(b0, v0, row_of(it + 2), p2);
      __builtin_amdgcn_sched_barrier(0);
      finish(b1, v1, it + 1);
      __builtin_amdgcn_sched_barrier(0);
      p1 = p3;
    }
  }
  if (DOT) {
    // sum the 8 row slots of the wave (lanes l, l^8, l^16, l^32 share a column pair), then the 4 waves
    auto sx = [](double v, int mask) {
      int lo = __shfl_xor(__double2loint(v), mask, 64), hi = __shfl_xor(__double2hiint(v), mask, 64);
      return __hiloint2double(hi, lo);
    };
    d0 += sx(d0, 8);  d1 += sx(d1, 8);  e0 += sx(e0, 8);  e1 += sx(e1, 8);
    d0 += sx(d0, 16); d1 += sx(d1, 16); e0 += sx(e0, 16); e1 += sx(e1, 16);
    d0 += sx(d0, 32); d1 += sx(d1, 32); e0 += sx(e0, 32); e1 += sx(e1, 32);
    __shared__ double sred[4][32];
    if (lane < 8) {
      sred[wave][2 * lane] = d0; sred[wave][2 * lane + 1] = d1;
      sred[wave][16 + 2 * lane] = e0; sred[wave][16 + 2 * lane + 1] = e1;
    }
    __syncthreads();
    const int tq = threadIdx.x & 15;
    if (threadIdx.x < 32 && tq < m) {   // threads 0-15: x.y partials, 16-31: y.y partials (second half of the workspace)
      const double v = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
      dot_partial[(threadIdx.x < 16 ? 0 : yy_offset) + (long)blockIdx.x * m + tq] = v;
    }
  }
}


// ---- chain variant: the +-S rows stay in registers ------------------------------------------------------------
// Table layout per pattern (built by app_hip.hip when the matrix qualifies): slot 0: offset -S, slot 1: offset 0,
// slot 2: offset +S, slots 3..LT-1: the remaining offsets, where S = the row stride of a wave between two of its
// iterations (blocks = S / 32).  Then the row fetched through slot 2 at iteration it, L(it) = X[row + S], IS the
// centre row of iteration it+1 and the "-S" row of iteration it+2: one new load per iteration replaces three
// (7 -> 5 loads per row on the 7-point stencil; the column dot gets its own-row value for free).  Four rotating
// registers hold L(it-2) .. L(it+1) (the loop is unrolled four times so the rotation is pure renaming), the other
// slots keep their two alternating register sets.  Offsets of slots 0 and 2 that would leave the matrix are stored
// as 0 with value 0 (patterns are split by that validity), so every lane whose NEXT row exists has loaded exactly
// that row through slot 2.
// LPR lanes serve one row (16 bytes each): a pass covers 2 LPR columns, a wave instruction 64 / LPR rows, a tile
// (4 waves) 256 / LPR rows.  With the +-S rows in registers the L2 no longer has to hold a grid plane, so wide
// passes are possible again (fewer table look-ups and launches per byte).
template <int LT, int DOT, int LPR>
__global__ __launch_bounds__(256) void spmm_pattern_chain_kernel(
    long nrows, const unsigned short* __restrict__ pid, const PatEntry* __restrict__ tab, int ntab,
    const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int m, long ntiles,
    double* __restrict__ dot_partial, long yy_offset) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  PatEntry* s_tab = reinterpret_cast<PatEntry*>(smem_raw);
  for (int e = threadIdx.x; e < ntab; e += 256) s_tab[e] = tab[e];
  __syncthreads();
  constexpr int NO = LT - 3;   // slots outside the chain

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPS = 64 / LPR, TR = 4 * RPS;   // rows per slice (wave instruction) and per tile (block iteration)
  const int g = lane / LPR, i = lane % LPR;
  const bool act = 2 * i < m;
  const double* __restrict__ xl = x + (act ? 2 * i : 0);
  double d0 = 0.0, d1 = 0.0, e0 = 0.0, e1 = 0.0;   // x.y and y.y column sums (DOT)
  const long G = gridDim.x;
  if ((long)blockIdx.x < ntiles) {
    const long cnt = (ntiles - blockIdx.x + G - 1) / G;
    auto row_at = [&](long it) { return (blockIdx.x + it * G) * TR + RPS * wave + g; };              // unclamped
    auto row_of = [&](long it) { return min(row_at(min(it, cnt - 1)), nrows - 1); };               // clamped
    auto issue = [&](v2d& lnew, v2d (&oth)[NO], double (&val)[LT], long row, int p) {
      const PatEntry* e = s_tab + p * LT;
      val[0] = e[0].val; val[1] = e[1].val;
      { const PatEntry c = e[2]; val[2] = c.val; lnew = *reinterpret_cast<const v2d*>(xl + (size_t)(row + c.off) * ldx); }
#pragma unroll
      for (int t = 0; t < NO; ++t) {
        const PatEntry c = e[3 + t];
        val[3 + t] = c.val;
        oth[t] = *reinterpret_cast<const v2d*>(xl + (size_t)(row + c.off) * ldx);
      }
    };
    auto finish = [&](const v2d& a, const v2d& b, const v2d& c, const v2d (&oth)[NO], const double (&val)[LT], long it) {
      double a0 = val[0] * a.x, a1 = val[0] * a.y;
      a0 = fma(val[1], b.x, a0); a1 = fma(val[1], b.y, a1);
      a0 = fma(val[2], c.x, a0); a1 = fma(val[2], c.y, a1);
#pragma unroll
      for (int t = 0; t < NO; ++t) { a0 = fma(val[3 + t], oth[t].x, a0); a1 = fma(val[3 + t], oth[t].y, a1); }
      const long row = row_at(it);
      const bool ok = it < cnt && row < nrows && act;
      if (ok) {
        v2d o = {a0, a1};
        __builtin_nontemporal_store(o, reinterpret_cast<v2d*>(y + (size_t)row * ldy + 2 * i));
      }
      if (DOT) {
        const double wgt = ok ? 1.0 : 0.0;
        d0 = fma(a0 * wgt, b.x, d0); d1 = fma(a1 * wgt, b.y, d1);
        e0 = fma(a0 * wgt, a0, e0); e1 = fma(a1 * wgt, a1, e1);
      }
    };
    v2d r0, r1, r2, r3, o0[NO], o1[NO];
    double v0[LT], v1[LT];
    int p0 = pid[row_of(0)], p1 = pid[row_of(1)];
    {   // start of the chain: L(-2) = X[row0 - S] and L(-1) = X[row0] through slots 0 and 1 of the first row's pattern
      const long row = row_of(0);
      const PatEntry* e = s_tab + p0 * LT;
      r2 = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e[0].off) * ldx);
      r3 = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e[1].off) * ldx);
    }
    issue(r0, o0, v0, row_of(0), p0);
    for (long it = 0; it < cnt; it += 4) {
      const int p2 = pid[row_of(it + 2)];
      issue(r1, o1, v1, row_of(it + 1), p1);
      __builtin_amdgcn_sched_barrier(0);
      finish(r2, r3, r0, o0, v0, it);
      __builtin_amdgcn_sched_barrier(0);
      const int p3 = pid[row_of(it + 3)];
      issue(r2, o0, v0, row_of(it + 2), p2);
      __builtin_amdgcn_sched_barrier(0);
      finish(r3, r0, r1, o1, v1, it + 1);
      __builtin_amdgcn_sched_barrier(0);
      const int p4 = pid[row_of(it + 4)];
      issue(r3, o1, v1, row_of(it + 3), p3);
      __builtin_amdgcn_sched_barrier(0);
      finish(r0, r1, r2, o0, v0, it + 2);
      __builtin_amdgcn_sched_barrier(0);
      const int p5 = pid[row_of(it + 5)];
      issue(r0, o0, v0, row_of(it + 4), p4);
      __builtin_amdgcn_sched_barrier(0);
      finish(r1, r2, r3, o1, v1, it + 3);
      __builtin_amdgcn_sched_barrier(0);
      p1 = p5;
    }
  }
  if (DOT) {
    auto sx = [](double v, int mask) {
      int lo = __shfl_xor(__double2loint(v), mask, 64), hi = __shfl_xor(__double2hiint(v), mask, 64);
      return __hiloint2double(hi, lo);
    };
#pragma unroll
    for (int mk = LPR; mk < 64; mk <<= 1) { d0 += sx(d0, mk); d1 += sx(d1, mk); e0 += sx(e0, mk); e1 += sx(e1, mk); }
    constexpr int CPP = 2 * LPR;   // columns per pass
    __shared__ double sred[4][2 * CPP];
    if (lane < LPR) {
      sred[wave][2 * lane] = d0; sred[wave][2 * lane + 1] = d1;
      sred[wave][CPP + 2 * lane] = e0; sred[wave][CPP + 2 * lane + 1] = e1;
    }
    __syncthreads();
    const int tq = threadIdx.x % CPP;
    if (threadIdx.x < 2 * CPP && tq < m) {   // first CPP threads: x.y partials, next CPP: y.y partials (second half of the workspace)
      const double v = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
      dot_partial[(threadIdx.x < CPP ? 0 : yy_offset) + (long)blockIdx.x * m + tq] = v;
    }
  }
}


// ---- chain + line exchange: the +-S rows in registers, the +-L rows through LDS ------------------------------
// Table layout: slots 0,1,2 = offsets -S, 0, +S as above, slots 3,4 = offsets -L, +L (L = the second longest
// offset of the interior stencil: N on an N^3 grid), slots 5.. the rest.  A block of NW waves works on NW slices
// that lie L rows apart (wave w on grid line NW q + w), so the "-L" row of wave w IS the centre row of wave w-1 of
// the same block in the same iteration, lane for lane.  The centre row is known one iteration ahead (it arrived as
// the "+S" row), so every wave posts it in LDS, one barrier later the neighbours read it: only the lowest wave
// still loads its -L row and the highest its +L row from memory.  Loads per row on the 7-point stencil:
// 1 (+S) + 2 (+-1) + 2/NW, against 5 for the chain alone and 7 without it.  The barrier carries no memory fence
// (raw s_barrier after lgkmcnt(0)): the global loads of the NEXT iteration stay in flight across it.
// ROLE: 0 lowest wave, 1 inner wave, 2 highest wave (three copies of the loop: no branch near a load).
template <int LT, int MODE, int NW, int ROLE, bool VALS>
__device__ __forceinline__ void chain2_body(
    long nrows, const unsigned short* __restrict__ pid, const PatEntry* s_tab, v2d (*xch)[NW][64],
    const double* __restrict__ xl, size_t ldx, double* __restrict__ y, size_t ldy, bool act, int i, int g, int wave, int lane,
    long ntiles, long line, int xcd_runs, double& d0, double& d1, double& e0, double& e1, const CgArgs& cg, const v2d* s_cf) {
  constexpr int UPD = MODE == 3 || MODE == 5 || MODE == 7;
  const double* __restrict__ rl = UPD ? (MODE == 5 ? cg.b : cg.r) + (act ? 2 * i : 0) : nullptr;
  const size_t ldrl = MODE == 5 ? cg.ldb : cg.ldr;
  constexpr int NO = LT - 5;               // slots that are neither chain nor line
  constexpr int NE = (ROLE == 1) ? 0 : 1;  // line row still loaded from memory
  // Tuning hook (off): blocks are dealt round-robin to the 8 XCDs; tiles that are neighbours along a grid line share
  // the rows at their common edge, so one could give each XCD a CONTIGUOUS run of the G tiles of a sweep (block b
  // takes tile (b % 8) * G/8 + b / 8).  Measured: slower (3.43 vs 3.19 ms at 256^3), same at 240^3.
  const long G = gridDim.x, asl = line / 8;
  // xcd_runs == 2: runs of 4 consecutive tiles (neighbours along a grid line) per XCD, the runs dealt round-robin:
  // tile = 32 (b / 32) + 4 (b % 8) + (b / 8) % 4; tiles 32 apart (the +-L neighbours at 256-row lines) stay on one XCD
  const long bx = blockIdx.x;
  const long b0 = (xcd_runs == 2 && G % 32 == 0) ? ((bx & ~31L) | ((bx & 7) << 2) | ((bx >> 3) & 3))
                : (G % 8 == 0 && xcd_runs == 1) ? ((bx & 7) * (G >> 3) + (bx >> 3)) : bx;
  const long cnt = (ntiles - b0 + G - 1) / G;
  auto row_at = [&](long it) {
    const long t = b0 + it * G, q = t / asl, a = t - q * asl;
    return ((long)NW * q + wave) * line + 8 * a + g;
  };
  auto row_of = [&](long it) { return min(row_at(min(it, cnt - 1)), nrows - 1); };
  // the stencil values are looked up again when the rows are reduced (7 LDS reads) instead of being carried in
  // 2 x LT registers from issue to finish: keeps the kernel at 4 waves per SIMD with the dot accumulators
  auto issue = [&](v2d& lnew, v2d (&edge)[NE + 1], v2d (&oth)[NO + 1 + UPD], long row, int p, double& vv) {
    const PatEntry* e = s_tab + p * LT;
    lnew = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e[2].off) * ldx);
    if (VALS) vv = cg.rowval[(size_t)row * 8 + i];   // the values of this row, one slot per lane of its group (LT <= 8)
    if (UPD) oth[NO + UPD] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(rl + (size_t)row * ldrl));   // the row's residual / right-hand side
    if (ROLE == 0) edge[0] = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e[3].off) * ldx);
    if (ROLE == 2) edge[0] = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e[4].off) * ldx);
#pragma unroll
    for (int t = 0; t < NO; ++t) oth[t] = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e[5 + t].off) * ldx);
  };
  auto finish = [&](const v2d& a, const v2d& b, const v2d& c, const v2d (&edge)[NE + 1], const v2d (&oth)[NO + 1 + UPD],
                    int p, long it, int buf, double vv) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    double val[LT];
#pragma unroll
    for (int t = 0; t < LT; ++t) val[t] = VALS ? group_bcast(vv, lane, t) : s_tab[p * LT + t].val;
    const v2d vm = (ROLE == 0) ? edge[0] : xch[buf][wave - (ROLE == 0 ? 0 : 1)][lane];
    const v2d vp = (ROLE == 2) ? edge[0] : xch[buf][wave + (ROLE == 2 ? 0 : 1)][lane];
    double a0 = val[0] * a.x, a1 = val[0] * a.y;
    a0 = fma(val[1], b.x, a0); a1 = fma(val[1], b.y, a1);
    a0 = fma(val[2], c.x, a0); a1 = fma(val[2], c.y, a1);
    a0 = fma(val[3], vm.x, a0); a1 = fma(val[3], vm.y, a1);
    a0 = fma(val[4], vp.x, a0); a1 = fma(val[4], vp.y, a1);
#pragma unroll
    for (int t = 0; t < NO; ++t) { a0 = fma(val[5 + t], oth[t].x, a0); a1 = fma(val[5 + t], oth[t].y, a1); }
    const long row = row_at(it);
    const bool ok = it < cnt && row < nrows && act;
    // MODE 2 stores nothing (the launcher passes y == NULL), but keeps the never-taken branch: without it hipcc
    // schedules the straight-line body into 200+ VGPRs / spills (118 with it), measured on the resource remarks
    if ((MODE <= 1 && ok) || ((MODE == 2 || MODE == 4) && ok && y != nullptr)) {
      v2d o = {a0, a1};
      __builtin_nontemporal_store(o, reinterpret_cast<v2d*>(y + (size_t)row * ldy + 2 * i));
    }
    const double wgt = ok ? 1.0 : 0.0;
    if (MODE == 1 || MODE == 2) {
      d0 = fma(a0 * wgt, b.x, d0); d1 = fma(a1 * wgt, b.y, d1);
      e0 = fma(a0 * wgt, a0, e0); e1 = fma(a1 * wgt, a1, e1);
    }
    if (MODE == 5 || MODE == 6) {
      v2d rv;
      if (MODE == 5) rv = oth[NO + UPD];
      else {   // B = X diag(scale), rounded like the column scaling (no contraction into fma(scale, x, -Ax))
#pragma clang fp contract(off)
        const v2d sc = s_cf[i]; rv = v2d{sc.x * b.x, sc.y * b.y};
      }
      v2d rn = {rv.x - a0, rv.y - a1};
      if (ok) {
        __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(cg.r + (size_t)row * cg.ldr + 2 * i));
        if (cg.pnew != cg.r) __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(cg.pnew + (size_t)row * cg.ldp + 2 * i));   // (equal: the residual alone)
      }
      d0 = fma(wgt * rn.x, rn.x, d0); d1 = fma(wgt * rn.y, rn.y, d1);
    }
    if (MODE == 3 || MODE == 7) {   // coefficients of this lane's column pair from LDS (kept out of the registers: 4 waves per SIMD)
      const v2d al = s_cf[i], cb = s_cf[8 + i], cr = s_cf[16 + i];
      v2d rv = oth[NO + UPD];
      if (MODE == 7) { const v2d bp = s_cf[24 + i]; rv = v2d{fma(-bp.x, rv.x, b.x), fma(-bp.y, rv.y, b.y)}; }   // r_k = p_k - beta_{k-1} p_{k-1}
      v2d rn = {fma(-al.x, a0, rv.x), fma(-al.y, a1, rv.y)};
      v2d pn = {fma(cb.x, b.x, cr.x * rn.x), fma(cb.y, b.y, cr.y * rn.y)};
      if (ok) {
        if (MODE == 3) __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(cg.r + (size_t)row * cg.ldr + 2 * i));
        __builtin_nontemporal_store(pn, reinterpret_cast<v2d*>(cg.pnew + (size_t)row * cg.ldp + 2 * i));
      }
      d0 = fma(cr.x * wgt * rn.x, rn.x, d0); d1 = fma(cr.y * wgt * rn.y, rn.y, d1);
    }
    if (MODE == 4) {
      const v2d lam = s_cf[i];
      const double q0 = fma(-lam.x, b.x, a0), q1 = fma(-lam.y, b.y, a1);
      d0 = fma(q0 * wgt, q0, d0); d1 = fma(q1 * wgt, q1, d1);
    }
    xch[buf ^ 1][wave][lane] = c;   // the centre row of my next iteration
  };
  v2d r0, r1, r2, r3, ed0[NE + 1], ed1[NE + 1], o0[NO + 1 + UPD], o1[NO + 1 + UPD];
  double vv0 = 0.0, vv1 = 0.0;
  int p0 = pid[row_of(0)], p1 = pid[row_of(1)];
  {
    const long row = row_of(0);
    const PatEntry* e = s_tab + p0 * LT;
    r2 = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e[0].off) * ldx);
    r3 = *reinterpret_cast<const v2d*>(xl + (size_t)(row + e[1].off) * ldx);
    xch[0][wave][lane] = r3;
  }
  issue(r0, ed0, o0, row_of(0), p0, vv0);
  for (long it = 0; it < cnt; it += 4) {   // pattern ids: p0 = row(it), p1 = row(it+1)
    const int p2 = pid[row_of(it + 2)];
    issue(r1, ed1, o1, row_of(it + 1), p1, vv1);
    __builtin_amdgcn_sched_barrier(0);
    finish(r2, r3, r0, ed0, o0, p0, it, 0, vv0);
    __builtin_amdgcn_sched_barrier(0);
    const int p3 = pid[row_of(it + 3)];
    issue(r2, ed0, o0, row_of(it + 2), p2, vv0);
    __builtin_amdgcn_sched_barrier(0);
    finish(r3, r0, r1, ed1, o1, p1, it + 1, 1, vv1);
    __builtin_amdgcn_sched_barrier(0);
    const int p4 = pid[row_of(it + 4)];
    issue(r3, ed1, o1, row_of(it + 3), p3, vv1);
    __builtin_amdgcn_sched_barrier(0);
    finish(r0, r1, r2, ed0, o0, p2, it + 2, 0, vv0);
    __builtin_amdgcn_sched_barrier(0);
    const int p5 = pid[row_of(it + 5)];
    issue(r0, ed0, o0, row_of(it + 4), p4, vv0);
    __builtin_amdgcn_sched_barrier(0);
    finish(r1, r2, r3, ed1, o1, p3, it + 3, 1, vv1);
    __builtin_amdgcn_sched_barrier(0);
    p0 = p4; p1 = p5;
  }
}

template <int LT, int MODE, int NW, bool VALS = false>
__global__ __launch_bounds__(64 * NW) void spmm_pattern_chain2_kernel(
    long nrows, const unsigned short* __restrict__ pid, const PatEntry* __restrict__ tab, int ntab,
    const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int m, long ntiles, long line,
    double* __restrict__ dot_partial, long yy_offset, int xcd_runs, CgArgs cg) {
  constexpr int DOT = MODE != 0;
  // gridDim.y > 1: the 16-column passes of one operation in ONE launch (grids that would leave CUs idle pass by pass: the coarse
  // levels of a multigrid hierarchy); pass blockIdx.y works on columns [16 y, 16 y + 16) of every operand, m = all the columns
  if (gridDim.y > 1) {
    const int c0 = 16 * (int)blockIdx.y;
    x += c0; if (y != nullptr) y += c0;
    if (dot_partial != nullptr) dot_partial += (size_t)blockIdx.y * gridDim.x * 16;
    if (cg.r != nullptr) cg.r += c0;
    if (cg.pnew != nullptr) cg.pnew += c0;
    if (cg.alpha != nullptr) cg.alpha += c0;
    if (cg.beta != nullptr) cg.beta += c0;
    if (cg.flag != nullptr) cg.flag += c0;
    if (cg.b != nullptr) cg.b += c0;
    m = min(m - c0, 16);
  }
  extern __shared__ __align__(16) unsigned char smem_raw[];
  PatEntry* s_tab = reinterpret_cast<PatEntry*>(smem_raw);
  __shared__ v2d xch[2][NW][64];
  __shared__ v2d s_cf[32];   // MODE 3 / 7: (alpha, cb, cr [, previous beta]) of the 8 column pairs of this pass
  for (int e = threadIdx.x; e < ntab; e += 64 * NW) s_tab[e] = tab[e];
  if ((MODE == 3 || MODE == 7) && threadIdx.x < 8) {
    const CgCoef c = cg_coef(cg, 2 * threadIdx.x, 2 * (int)threadIdx.x < m);
    s_cf[threadIdx.x] = v2d{c.al0, c.al1}; s_cf[8 + threadIdx.x] = v2d{c.cb0, c.cb1}; s_cf[16 + threadIdx.x] = v2d{c.cr0, c.cr1};
    if (MODE == 7) s_cf[24 + threadIdx.x] = (2 * (int)threadIdx.x < m) ? v2d{cg.b[2 * threadIdx.x], cg.b[2 * threadIdx.x + 1]} : v2d{0.0, 0.0};
  }
  if ((MODE == 4 || MODE == 6) && threadIdx.x < 8)
    s_cf[threadIdx.x] = (2 * (int)threadIdx.x < m) ? v2d{cg.alpha[2 * threadIdx.x], cg.alpha[2 * threadIdx.x + 1]} : v2d{0.0, 0.0};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 3, i = lane & 7;
  const bool act = 2 * i < m;
  const double* __restrict__ xl = x + (act ? 2 * i : 0);
  double d0 = 0.0, d1 = 0.0, e0 = 0.0, e1 = 0.0;
  const long gq = gridDim.x;
  const long bq = blockIdx.x;
  const long bperm = (xcd_runs == 2 && gq % 32 == 0) ? ((bq & ~31L) | ((bq & 7) << 2) | ((bq >> 3) & 3))
                   : (gq % 8 == 0 && xcd_runs == 1) ? ((bq & 7) * (gq >> 3) + (bq >> 3)) : bq;
  if (bperm < ntiles) {   // block-uniform: every wave of the block runs the same number of barriers
    if (wave == 0) chain2_body<LT, MODE, NW, 0, VALS>(nrows, pid, s_tab, xch, xl, ldx, y, ldy, act, i, g, wave, lane, ntiles, line, xcd_runs, d0, d1, e0, e1, cg, s_cf);
    else if (wave == NW - 1) chain2_body<LT, MODE, NW, 2, VALS>(nrows, pid, s_tab, xch, xl, ldx, y, ldy, act, i, g, wave, lane, ntiles, line, xcd_runs, d0, d1, e0, e1, cg, s_cf);
    else chain2_body<LT, MODE, NW, 1, VALS>(nrows, pid, s_tab, xch, xl, ldx, y, ldy, act, i, g, wave, lane, ntiles, line, xcd_runs, d0, d1, e0, e1, cg, s_cf);
  }
  if (DOT) {
    auto sx = [](double v, int mask) {
      int lo = __shfl_xor(__double2loint(v), mask, 64), hi = __shfl_xor(__double2hiint(v), mask, 64);
      return __hiloint2double(hi, lo);
    };
    d0 += sx(d0, 8);  d1 += sx(d1, 8);  e0 += sx(e0, 8);  e1 += sx(e1, 8);
    d0 += sx(d0, 16); d1 += sx(d1, 16); e0 += sx(e0, 16); e1 += sx(e1, 16);
    d0 += sx(d0, 32); d1 += sx(d1, 32); e0 += sx(e0, 32); e1 += sx(e1, 32);
    __shared__ double sred[NW][32];
    if (lane < 8) {
      sred[wave][2 * lane] = d0; sred[wave][2 * lane + 1] = d1;
      sred[wave][16 + 2 * lane] = e0; sred[wave][16 + 2 * lane + 1] = e1;
    }
    __syncthreads();
    const int tq = threadIdx.x & 15;
    if (threadIdx.x < 32 && tq < m) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += sred[w][threadIdx.x];
      dot_partial[(threadIdx.x < 16 ? 0 : yy_offset) + (long)blockIdx.x * m + tq] = v;
    }
  }
}

}  // namespace gcge

using namespace gcge;

// Blocks per pass.  A wave handles slices w, w + W, w + 2W, ... (W = 4 * blocks), i.e. rows 32 * blocks apart.
// Measured on the 256^3 Laplacian (64 columns, profiles/r01_spmm_explore/13_pattern_grid.log):
//     blocks  512    768    1024   1536   2048   4096   8192
//     ms      4.78   6.18   5.02   6.29   4.55   5.23   6.28
// Fast whenever the row stride divides the matrix's longest column offset (here N^2 = 65536 rows = 2048
// blocks): then the X rows a wave fetched as "+N^2" neighbours are the rows it needs itself one (or a few)
// iterations later, so the reuse is private to the wave / CU and does not depend on blocks advancing in
// lock-step.  Otherwise fast blocks drift planes ahead of slow ones and the reuse is lost from L2.
// Hence: blocks = span / 32 / j with the smallest j that divides span / 32 and keeps the grid below ~3072 blocks.
static int g_pat_grid = 0;   // > 0: forced (tuning)
extern "C" void gcge_hip_spmm_pattern_tune(int grid) { g_pat_grid = grid > 0 ? (grid + 7) / 8 * 8 : 0; }
// tiles of 4 slices `line` rows apart: groups of 4 lines x (line / 8) slices per line
static long pat_ntiles(long nrows, long line) {
  const long nlines = (nrows + line - 1) / line;
  return (nlines + 3) / 4 * (line / 8);
}
// Measured (profiles/r01_spmm_explore/16_pattern_line_tiles.log): tiles one grid line apart do NOT pay — 4.98 vs
// 5.02 ms at 256^3, 3.05 vs 2.84 ms on the 200^3 FE matrix — the four waves of a block are not in step closely
// enough for the +-N rows to still be in the 32 KB L1.  Default: 4 consecutive slices (line = 8).
// chain + line exchange: most waves per block to try (16, 8, 4; measured 3.20 / 3.37 / 3.67 ms at 256^3 x 64);
// 0: use the plain chain kernel
static int g_pass_streams = 0;   // > 1: the 16-column passes of a CG sweep on that many side streams (gcge_hip_pattern_cg_vals)
extern "C" void gcge_hip_cg_pass_streams(int n) { g_pass_streams = n < 0 ? 0 : (n > 4 ? 4 : n); }
// Column passes of one operation merged into ONE launch (gridDim.y = passes) when a pass alone has at most this many blocks:
// a 128^3 level of a multigrid hierarchy has 128 blocks of 16 waves per pass — half the CUs idle, 64^3 an eighth of them.
// Same blocks, same partial sums, same results bit for bit; the finest level of config 2 (512 blocks) keeps its passes apart
// (a pass's working set per XCD is what its L2 holds).  The plain pattern kernel (blocks of 4 waves: the FE pair of config 3, 312
// blocks per pass at n = 10^6) follows the same rule counted in waves: merged while a pass has at most 16 x this many.  0: never.
static int g_pass_merge_blocks = -1;   // -1: not set yet (GCGE_PASS_MERGE in the environment, else 256)
extern "C" void gcge_hip_spmm_pass_merge(int max_blocks) { g_pass_merge_blocks = max_blocks < 0 ? 0 : max_blocks; }
static int pass_merge_blocks() {
  if (g_pass_merge_blocks < 0) { const char* e = getenv("GCGE_PASS_MERGE"); g_pass_merge_blocks = e ? atoi(e) : 256; if (g_pass_merge_blocks < 0) g_pass_merge_blocks = 0; }
  return g_pass_merge_blocks;
}
static int g_chain2_nw = 16;
static int g_chain2_xcd = 2;   // 2: runs of 4 neighbouring tiles per XCD (pass 2 6.50 -> 6.42 ms, fabric reads down); 1: one contiguous eighth of
                               // the tiles per XCD (slower: 3.43 vs 3.19 ms at 256^3); 0: tiles in block order
extern "C" void gcge_hip_spmm_chain2_xcd(int on) { g_chain2_xcd = on; }
extern "C" void gcge_hip_spmm_chain2_tune(int waves) { if (waves == 0 || waves == 4 || waves == 8 || waves == 16) g_chain2_nw = waves; }
static int g_chain_lpr = 8;    // chain variant: lanes per row = half the columns per pass (8, 16, 32)
extern "C" void gcge_hip_spmm_chain_tune(int lanes_per_row) { if (lanes_per_row == 8 || lanes_per_row == 16 || lanes_per_row == 32) g_chain_lpr = lanes_per_row; }
static int g_pat_line = 8;    // tuning: -1 = from the stencil's second longest offset, 8 = consecutive slices
extern "C" void gcge_hip_spmm_pattern_tune_line(int line) { g_pat_line = line; }
static long pat_grid(long span, long ntiles) {
  long g;
  if (g_pat_grid > 0) g = g_pat_grid;
  else {
    const long target = span / 32;
    if (target < 256) g = 1024;                          // short reuse distances live in L2 anyway
    else {   // smallest j that divides the plane into an integral number of strides of at most ~3072 blocks
      long j = (target + 3071) / 3072;
      while (j < 64 && target % j != 0) ++j;
      g = (target % j == 0) ? target / j : (target / j + 7) / 8 * 8;
    }
  }
  return g < ntiles ? g : ntiles;
}

template <int LT, int MODE, bool VALS>
static long pat_launch(long nrows, const unsigned short* pid, const void* tab, int npat, const double* x, size_t ldx,
                       double* y, size_t ldy, int m, double* partial, long yy_off, long nb, long line, hipStream_t st,
                       long cline, int nw, const CgArgs& cg, int gy = 1) {
  const int ntab = npat * LT;
  if (gy > 1 && cline <= 0 && line < 0) return -1;   // merged column passes: the chain2 and the plain kernel
  if (cline > 0) {   // chain + line exchange: nw waves per block, lines of `cline` rows
    if (LT < 5) return -1;
    const long nlines = (nrows + cline - 1) / cline, ntl = (nlines + nw - 1) / nw * (cline / 8);
#define GCGE_C2(NWV) hipLaunchKernelGGL((spmm_pattern_chain2_kernel<(LT < 5 ? 5 : LT), MODE, NWV, VALS>), dim3((unsigned)nb, (unsigned)gy), dim3(64 * NWV), \
                       (size_t)ntab * sizeof(PatEntry), st, nrows, pid, (const PatEntry*)tab, ntab, x, ldx, y, ldy, m, ntl, cline, partial, yy_off, g_chain2_xcd, cg)
    if (nw == 16) GCGE_C2(16); else if (nw == 8) GCGE_C2(8); else GCGE_C2(4);
#undef GCGE_C2
    return nb;
  }
  if (line < 0) {   // chain variant: consecutive slices, the caller fixed nb = S / 32
    if constexpr (MODE >= 2 || VALS) return -1;   // the CG passes and the streamed values exist for the chain2 and the plain kernel
    else {
      const long lpr = -line;   // chain variant: lanes per row is passed as -line (8, 16 or 32)
      const long tr = 256 / lpr, ntl = (nrows + tr - 1) / tr;
#define GCGE_CH(L) hipLaunchKernelGGL((spmm_pattern_chain_kernel<LT, MODE, L>), dim3((unsigned)nb), dim3(256), \
                       (size_t)ntab * sizeof(PatEntry), st, nrows, pid, (const PatEntry*)tab, ntab, x, ldx, y, ldy, m, ntl, partial, yy_off)
      if (lpr == 32) GCGE_CH(32); else if (lpr == 16) GCGE_CH(16); else GCGE_CH(8);
#undef GCGE_CH
      return nb;
    }
  }
  hipLaunchKernelGGL((spmm_pattern_kernel<LT, MODE, VALS>), dim3((unsigned)nb, (unsigned)gy), dim3(256), (size_t)ntab * sizeof(PatEntry), st,
                     nrows, pid, (const PatEntry*)tab, ntab, x, ldx, y, ldy, m, pat_ntiles(nrows, line), line, partial, yy_off, cg);
  return nb;
}

template <int MODE>
static long pat_dispatch(int lt, long nrows, const unsigned short* pid, const void* tab, int npat, const double* x,
                         size_t ldx, double* y, size_t ldy, int m, double* partial, long yy_off, long nb, long line, hipStream_t st,
                         long cline = 0, int nw = 4, const CgArgs& cg = CgArgs{}, int gy = 1) {
  if (cg.rowval != nullptr) {   // offsets-only table, values streamed per row (tables of at most 8 slots)
    switch (lt) {
      case 7: return pat_launch<7, MODE, true>(nrows, pid, tab, npat, x, ldx, y, ldy, m, partial, yy_off, nb, line, st, cline, nw, cg, gy);
      case 8: return pat_launch<8, MODE, true>(nrows, pid, tab, npat, x, ldx, y, ldy, m, partial, yy_off, nb, line, st, cline, nw, cg, gy);
      default: return -1;
    }
  }
  switch (lt) {
    case 7: return pat_launch<7, MODE, false>(nrows, pid, tab, npat, x, ldx, y, ldy, m, partial, yy_off, nb, line, st, cline, nw, cg, gy);
    case 8: return pat_launch<8, MODE, false>(nrows, pid, tab, npat, x, ldx, y, ldy, m, partial, yy_off, nb, line, st, cline, nw, cg, gy);
    case 16: return pat_launch<16, MODE, false>(nrows, pid, tab, npat, x, ldx, y, ldy, m, partial, yy_off, nb, line, st, cline, nw, cg, gy);
    default: return -1;
  }
}

// table entries per pattern the kernels are built for (rows are padded up to one of these), 0: too long
extern "C" int gcge_hip_pattern_width(int max_row_len) {
  if (max_row_len <= 7) return 7;
  if (max_row_len <= 8) return 8;
  if (max_row_len <= 16) return 16;
  return 0;
}

// Y[:,0:ncols) = A X[:,0:ncols); d_dots != NULL: also d_dots[j] = sum_r X[r,j] Y[r,j] and, if d_dots_yy != NULL,
// d_dots_yy[j] = sum_r Y[r,j]^2 (both free: the kernel has the rows in registers).
// d_tab: npat * lt entries of {double value; long column_offset}; span / span2: the longest and second longest
// |column_offset| of the interior stencil (launch geometry only; 0 if unknown).  span2 == -1: the table is in
// CHAIN layout (slots 0,1,2 = offsets -span, 0, +span; see spmm_pattern_chain_kernel) and span is a multiple of 32;
// span2 == -L <= -8: additionally slots 3,4 = offsets -L, +L (spmm_pattern_chain2_kernel).
// -1: alignment contract not met.
extern "C" int gcge_hip_pattern_spmm_vals(int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                          long span, long span2, const double* d_x, long ldx, double* d_y, long ldy, int ncols,
                                          double* d_dots, double* d_dots_yy, void* stream, long near, const double* d_rowval);
extern "C" int gcge_hip_pattern_spmm_near(int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                          long span, long span2, const double* d_x, long ldx, double* d_y, long ldy, int ncols,
                                          double* d_dots, double* d_dots_yy, void* stream, long near) {
  gcge_hip_apply_pending();
  return gcge_hip_pattern_spmm_vals(nrows, d_pid, d_tab, npat, lt, span, span2, d_x, ldx, d_y, ldy, ncols, d_dots, d_dots_yy, stream, near, nullptr);
}
extern "C" int gcge_hip_pattern_spmm(int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                     long span, long span2, const double* d_x, long ldx, double* d_y, long ldy, int ncols,
                                     double* d_dots, double* d_dots_yy, void* stream) {
  gcge_hip_apply_pending();
  return gcge_hip_pattern_spmm_near(nrows, d_pid, d_tab, npat, lt, span, span2, d_x, ldx, d_y, ldy, ncols, d_dots, d_dots_yy, stream, 0);
}
// near > 0: as in gcge_hip_pattern_cg_near — the product may take the LDS-ring sweep (spmm_ring.hip)
// d_rowval != NULL: the table was built from the rows' OFFSETS only and the values of row r are d_rowval[8 r + slot]
// (app_hip.hip build_patterns, by_offsets): chain + line-exchange tables and plain tables of at most 8 slots
extern "C" int gcge_hip_pattern_spmm_vals(int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                          long span, long span2, const double* d_x, long ldx, double* d_y, long ldy, int ncols,
                                          double* d_dots, double* d_dots_yy, void* stream, long near, const double* d_rowval) {
  if (nrows <= 0 || ncols <= 0) return 0;
  if (d_rowval != nullptr && lt > 8) return -1;
  CgArgs cgv = CgArgs{}; cgv.rowval = d_rowval;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15)) return -1;
  if ((size_t)npat * lt * sizeof(PatEntry) > 64 * 1024) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int npass = (ncols + 15) / 16;
  // lines of `span2` rows when they tile the matrix exactly (a plane = a whole number of 4-line groups)
  long line = 8;
  // chain layout with line exchange: L = -span2 rows per grid line, nw waves per block (the most that tile a plane)
  int nw = 0;
  if (span2 <= -8 && (-span2) % 8 == 0 && lt >= 5)
    for (int cand = g_chain2_nw; cand >= 4; cand /= 2)
      if (span % (cand * -span2) == 0 && (long)nrows >= cand * -span2) { nw = cand; break; }
  if (nw > 0) {
    const long L = -span2;
    const long nbc = std::min(span / (8L * nw), ((((long)nrows + L - 1) / L + nw - 1) / nw) * (L / 8));
    const int npassc = (ncols + 15) / 16;
    hipStream_t stc = (hipStream_t)stream;
    double* partc = d_dots ? gcge_hip_partial_ws((size_t)nbc * 16 * npassc * 2) : nullptr;
    const long yyc = (long)nbc * 16 * npassc;
    bool ring = near > 0 && lt == 7 && d_x != d_y && d_rowval == nullptr;
    const bool merge = npassc > 1 && nbc * nw <= 16L * pass_merge_blocks();   // all passes in one launch (gridDim.y): a pass alone fills at most 16 waves per CU
    for (int c0 = 0, ps = 0; c0 < ncols; c0 += 16, ++ps) {
      const int m = merge ? ncols : ((ncols - c0 < 16) ? ncols - c0 : 16);
      const int gy = merge ? npassc : 1;
      double* pp = partc ? partc + (size_t)ps * nbc * 16 : nullptr;
      if (ring) {
        if (gcge_hip_ring_pass(d_dots ? 1 : 0, nrows, d_pid, d_tab, npat, L, nw, nbc, d_x + c0, ldx, m, pp, yyc, nullptr, stc, near,
                               d_y + c0, ldy, gy) == 0) { if (merge) break; continue; }
        ring = false;   // declined (first pass): the chain2 kernel below
      }
      long rcl = d_dots ? pat_dispatch<1>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, pp, yyc, nbc, 8, stc, L, nw, cgv, gy)
                        : pat_dispatch<0>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, nullptr, 0, nbc, 8, stc, L, nw, cgv, gy);
      if (rcl < 0) return -1;
      if (merge) break;
    }
    if (d_dots) gcge_hip_reduce_partials16(partc, (int)nbc, nbc * 16, ncols, d_dots, stc);
    if (d_dots && d_dots_yy) gcge_hip_reduce_partials16(partc + yyc, (int)nbc, nbc * 16, ncols, d_dots_yy, stc);
    return (int)hipGetLastError();
  }
  if (span2 <= -1 && d_rowval == nullptr) {   // chain layout: the wave stride must be exactly `span` rows (streamed values: the plain kernel below)
    if (span % 32 != 0 || lt < 4) return -1;
    // pass width: 16, 32 or 64 columns (lanes per row 8 / 16 / 32); the wave stride must stay exactly `span` rows
    int lpr = g_chain_lpr;
    while (lpr > 8 && (2 * lpr > ((ncols + 15) / 16) * 16 || span % (256 / lpr) != 0)) lpr /= 2;
    const int cpp = 2 * lpr;
    const long tr = 256 / lpr;
    const long nbc = std::min(span / tr, ((long)nrows + tr - 1) / tr);
    const int npassc = (ncols + cpp - 1) / cpp;
    hipStream_t stc = (hipStream_t)stream;
    double* partc = d_dots ? gcge_hip_partial_ws((size_t)nbc * cpp * npassc * 2) : nullptr;
    const long yyc = (long)nbc * cpp * npassc;   // the y.y partials follow the x.y partials
    for (int c0 = 0, ps = 0; c0 < ncols; c0 += cpp, ++ps) {
      const int m = (ncols - c0 < cpp) ? ncols - c0 : cpp;
      double* pp = partc ? partc + (size_t)ps * nbc * cpp : nullptr;
      long rcl = d_dots ? pat_dispatch<1>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, pp, yyc, nbc, -lpr, stc)
                        : pat_dispatch<0>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, nullptr, 0, nbc, -lpr, stc);
      if (rcl < 0) return -1;
    }
    if (d_dots) gcge_hip_reduce_partials_slabs(partc, (int)nbc, nbc * cpp, cpp, ncols, d_dots, stc);   // all passes in one launch
    if (d_dots && d_dots_yy) gcge_hip_reduce_partials_slabs(partc + yyc, (int)nbc, nbc * cpp, cpp, ncols, d_dots_yy, stc);
    return (int)hipGetLastError();
  }
  if (g_pat_line < 0 && span2 >= 8 && span2 % 8 == 0 && span > span2 && span % (4 * span2) == 0) line = span2;   // (span2 < 0 here: chain-layout table with streamed values, plain kernel)
  if (g_pat_line >= 8 && g_pat_line % 8 == 0) line = g_pat_line;
  const long nb = pat_grid(span, pat_ntiles(nrows, line));
  double* part = d_dots ? gcge_hip_partial_ws((size_t)nb * 16 * npass * 2) : nullptr;
  const long yyo = (long)nb * 16 * npass;
  const bool mergep = npass > 1 && nb * 4 <= 16L * pass_merge_blocks();   // plain kernel: blocks of 4 waves
  for (int c0 = 0, ps = 0; c0 < ncols; c0 += 16, ++ps) {
    const int m = mergep ? ncols : ((ncols - c0 < 16) ? ncols - c0 : 16);
    const int gy = mergep ? npass : 1;
    if (d_dots) {
      double* pp = part + (size_t)ps * nb * 16;
      if (pat_dispatch<1>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, pp, yyo, nb, line, st, 0, 4, cgv, gy) < 0) return -1;
    } else if (pat_dispatch<0>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, d_y + c0, (size_t)ldy, m, nullptr, 0, nb, line, st, 0, 4, cgv, gy) < 0) {
      return -1;
    }
    if (mergep) break;
  }
  if (d_dots) gcge_hip_reduce_partials16(part, (int)nb, nb * 16, ncols, d_dots, st);   // all passes in one launch
  if (d_dots && d_dots_yy) gcge_hip_reduce_partials16(part + yyo, (int)nb, nb * 16, ncols, d_dots_yy, st);
  return (int)hipGetLastError();
}

// The two passes of a block-CG iteration on a pattern matrix (MODE 2 and 3 above; block_pcg.hip):
//   mode 2: d_dots[j] = sum_r X[r,j] (A X)[r,j], d_dots_yy[j] = sum_r (A X)[r,j]^2; nothing is stored
//   mode 3: R -= (A X) diag(alpha); PNEW = R diag(cr) + X diag(cb); d_dots[j] = sum_r cr_j R[r,j]^2  (d_dots_yy unused)
//   mode 4: d_dots[j] = sum_r ((A X)[r,j] - alpha_j X[r,j])^2  (residuals of Ritz pairs, alpha = the Ritz values)
//   mode 5: R = B - A X; PNEW = R; d_dots[j] = sum_r R[r,j]^2  (start of the CG; d_b / ldb: the right-hand sides)
// Geometry as gcge_hip_pattern_spmm; a chain-layout table without line exchange runs through the plain kernel
// (its table is a valid generic one).  -1: not applicable (alignment), the caller keeps the unfused recurrence.
// near > 0: the table's slots are [-S, 0, +S, -L, +L, -1, +1] (7-point stencil) and near is the largest |offset| in it
// (GCGE_HIP_MAT_::pat_near): modes 2 and 4 may take the LDS-ring sweep of spmm_ring.hip.
extern "C" int gcge_hip_pattern_cg_vals(int mode, int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                        long span, long span2, const double* d_x, long ldx, double* d_r, long ldr, double* d_pnew,
                                        long ldp, int ncols, const double* d_alpha, const double* d_beta, const int* d_flag,
                                        double* d_dots, double* d_dots_yy, void* stream, const double* d_b, long ldb, long near,
                                        const double* d_rowval);
extern "C" int gcge_hip_pattern_cg_near(int mode, int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                        long span, long span2, const double* d_x, long ldx, double* d_r, long ldr, double* d_pnew,
                                        long ldp, int ncols, const double* d_alpha, const double* d_beta, const int* d_flag,
                                        double* d_dots, double* d_dots_yy, void* stream, const double* d_b, long ldb, long near) {
  gcge_hip_apply_pending();
  return gcge_hip_pattern_cg_vals(mode, nrows, d_pid, d_tab, npat, lt, span, span2, d_x, ldx, d_r, ldr, d_pnew, ldp, ncols, d_alpha, d_beta,
                                  d_flag, d_dots, d_dots_yy, stream, d_b, ldb, near, nullptr);
}
extern "C" int gcge_hip_pattern_cg_vals(int mode, int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                        long span, long span2, const double* d_x, long ldx, double* d_r, long ldr, double* d_pnew,
                                        long ldp, int ncols, const double* d_alpha, const double* d_beta, const int* d_flag,
                                        double* d_dots, double* d_dots_yy, void* stream, const double* d_b, long ldb, long near,
                                        const double* d_rowval) {
  if (d_rowval != nullptr && lt > 8) return -1;
  if (mode != 2 && mode != 3 && mode != 4 && mode != 5 && mode != 6 && mode != 7) return -1;
  if (nrows <= 0 || ncols <= 0) return 0;
  if ((ncols & 1) || (ldx & 1) || ((uintptr_t)d_x & 15) || d_dots == nullptr) return -1;
  if (mode == 6 && d_alpha == nullptr) return -1;
  if (mode == 7 && (d_b == nullptr || (ldr & 1) || (ldp & 1) || ((uintptr_t)d_r & 15) || ((uintptr_t)d_pnew & 15) || d_pnew == d_x || d_pnew == d_r)) return -1;
  if ((mode == 3 || mode == 5 || mode == 6) && ((ldr & 1) || (ldp & 1) || ((uintptr_t)d_r & 15) || ((uintptr_t)d_pnew & 15) || d_pnew == d_x || d_r == d_x)) return -1;
  if (mode == 5 && (d_b == nullptr || (ldb & 1) || ((uintptr_t)d_b & 15))) return -1;
  if ((size_t)npat * lt * sizeof(PatEntry) > 64 * 1024) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int npass = (ncols + 15) / 16;
  int nw = 0;
  if (span2 <= -8 && (-span2) % 8 == 0 && lt >= 5)
    for (int cand = g_chain2_nw; cand >= 4; cand /= 2)
      if (span % (cand * -span2) == 0 && (long)nrows >= cand * -span2) { nw = cand; break; }
  const long L = -span2;
  long nb, line = 8, cline = 0;
  if (nw > 0) { nb = std::min(span / (8L * nw), ((((long)nrows + L - 1) / L + nw - 1) / nw) * (L / 8)); cline = L; }
  else nb = pat_grid(span, pat_ntiles(nrows, line));
  double* part = gcge_hip_partial_ws((size_t)nb * 16 * npass * 2);
  const long yyo = (long)nb * 16 * npass;
  // read-only passes on a [-S, 0, +S, -L, +L, -1, +1] table: the LDS-ring sweep (spmm_ring.hip), same geometry and workspace
  bool ring = near && lt == 7 && nw >= 4 && (mode == 2 || mode == 4) && d_rowval == nullptr;
  // The column passes are independent (their own columns, their own partial sums): g_pass_streams > 1 puts them on that many
  // side streams between a fork and a join on `stream` (gcge_hip_cg_pass_streams; measured in profiles/r04_bench).
  static hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
  static hipEvent_t ev_fork = nullptr, ev_join[4];
  const hipStream_t st_main = st;
  const bool merge = npass > 1 && nb * (nw > 0 ? nw : 4) <= 16L * pass_merge_blocks();   // all passes in one launch (gridDim.y); plain kernel: blocks of 4 waves
  const int gy = merge ? npass : 1;
  const int nside = (g_pass_streams > 1 && npass > 1 && !merge) ? std::min(g_pass_streams, std::min(npass, 4)) : 0;
  if (nside > 0) {
    if (ev_fork == nullptr) {
      GCGE_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
      for (int q = 0; q < 4; ++q) { GCGE_HIP_CHECK(hipStreamCreateWithFlags(&side[q], hipStreamNonBlocking)); GCGE_HIP_CHECK(hipEventCreateWithFlags(&ev_join[q], hipEventDisableTiming)); }
    }
    GCGE_HIP_CHECK(hipEventRecord(ev_fork, st_main));
    for (int q = 0; q < nside; ++q) GCGE_HIP_CHECK(hipStreamWaitEvent(side[q], ev_fork, 0));
  }
  auto join = [&]() {
    for (int q = 0; q < nside; ++q) { GCGE_HIP_CHECK(hipEventRecord(ev_join[q], side[q])); GCGE_HIP_CHECK(hipStreamWaitEvent(st_main, ev_join[q], 0)); }
  };
  for (int c0 = 0, ps = 0; c0 < ncols; c0 += 16, ++ps) {
    const int m = merge ? ncols : ((ncols - c0 < 16) ? ncols - c0 : 16);
    double* pp = part + (size_t)ps * nb * 16;
    long rc;
    if (nside > 0) st = side[ps % nside];
    if (ring) {
      if (gcge_hip_ring_pass(mode, nrows, d_pid, d_tab, npat, L, nw, nb, d_x + c0, ldx, m, pp, yyo, mode == 4 ? d_alpha + c0 : nullptr, st, near, nullptr, 0, gy) == 0) {
        if (merge) break;
        continue;
      }
      ring = false;   // declined (first pass): the chain2 kernel below
    }
    if (mode == 2) { CgArgs cg = CgArgs{}; cg.rowval = d_rowval; rc = pat_dispatch<2>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, nullptr, 0, m, pp, yyo, nb, line, st, cline, nw, cg, gy); }
    else if (mode == 4) {
      const CgArgs cg = {nullptr, 0, nullptr, 0, d_alpha + c0, nullptr, nullptr, nullptr, 0, d_rowval};
      rc = pat_dispatch<4>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, nullptr, 0, m, pp, yyo, nb, line, st, cline, nw, cg, gy);
    } else if (mode == 5) {
      const CgArgs cg = {d_r + c0, (size_t)ldr, d_pnew + c0, (size_t)ldp, nullptr, nullptr, nullptr, d_b + c0, (size_t)ldb, d_rowval};
      rc = pat_dispatch<5>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, nullptr, 0, m, pp, yyo, nb, line, st, cline, nw, cg, gy);
    } else if (mode == 6) {
      const CgArgs cg = {d_r + c0, (size_t)ldr, d_pnew + c0, (size_t)ldp, d_alpha + c0, nullptr, nullptr, nullptr, 0, d_rowval};
      rc = pat_dispatch<6>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, nullptr, 0, m, pp, yyo, nb, line, st, cline, nw, cg, gy);
    } else if (mode == 7) {   // d_r: p_{k-1} (read only), d_b: the previous iteration's beta
      const CgArgs cg = {d_r + c0, (size_t)ldr, d_pnew + c0, (size_t)ldp, d_alpha + c0, d_beta + c0, d_flag + c0, d_b + c0, 0, d_rowval};
      rc = pat_dispatch<7>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, nullptr, 0, m, pp, yyo, nb, line, st, cline, nw, cg, gy);
    } else {
      const CgArgs cg = {d_r + c0, (size_t)ldr, d_pnew + c0, (size_t)ldp, d_alpha + c0, d_beta + c0, d_flag + c0, nullptr, 0, d_rowval};
      rc = pat_dispatch<3>(lt, nrows, d_pid, d_tab, npat, d_x + c0, (size_t)ldx, nullptr, 0, m, pp, yyo, nb, line, st, cline, nw, cg, gy);
    }
    if (rc < 0) { join(); return -1; }
    if (merge) break;
  }
  join();
  st = st_main;
  gcge_hip_reduce_partials16(part, (int)nb, nb * 16, ncols, d_dots, st);
  if (mode == 2 && d_dots_yy) gcge_hip_reduce_partials16(part + yyo, (int)nb, nb * 16, ncols, d_dots_yy, st);
  return (int)hipGetLastError();
}
extern "C" int gcge_hip_pattern_cg(int mode, int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                   long span, long span2, const double* d_x, long ldx, double* d_r, long ldr, double* d_pnew,
                                   long ldp, int ncols, const double* d_alpha, const double* d_beta, const int* d_flag,
                                   double* d_dots, double* d_dots_yy, void* stream, const double* d_b, long ldb) {
  gcge_hip_apply_pending();
  return gcge_hip_pattern_cg_near(mode, nrows, d_pid, d_tab, npat, lt, span, span2, d_x, ldx, d_r, ldr, d_pnew, ldp, ncols, d_alpha,
                                  d_beta, d_flag, d_dots, d_dots_yy, stream, d_b, ldb, 0);
}
