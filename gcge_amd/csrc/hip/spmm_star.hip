// K1 (grid path) — rows that carry exactly a star stencil on a lexicographic 3-D grid, swept plane by plane.
//
// Same contract as the other K1 kernels (Y[:, 0:m) = A X[:, 0:m), reference app/app_ccs.c:50-139).  The matrices behind
// BASELINE config 5 (real-space DFT Hamiltonians, test_eig_sol_SiO2_MAT.c of the reference) are a high-order finite-difference
// Laplacian — a star of 6 R + 1 points, R = 6: 37 entries with the SAME 3 R coefficients in every row, a diagonal of its own per
// row (the local potential), Dirichlet truncation at the faces — plus dense blocks where the atoms sit.  The tile form
// (spmm_tile.hip) stages 8.3 X rows per row for such a star and streams 10 B per entry; here the rows whose off-diagonal
// entries ARE that star, bit for bit, are taken out of the CSR arrays altogether ("clean" rows: a per-row diagonal and a flag
// are all that is stored) and multiplied by a 2.5-D sweep:
//   * a workgroup owns a 16 x 16 patch of one z-range and 8 columns; it walks z, one plane per step;
//   * the 13 z-neighbours of a point live in REGISTERS of the lanes that own the point (a queue of 13 planes, rotated by
//     unrolling 13 steps: every index is a compile-time constant);
//   * the x / y arms of the current plane come from LDS: the patch's own values are written there from the queue, the 2 x 6
//     halo strips on each side are loaded from global memory one step ahead (64-byte segments, 4 lanes per row);
//   * per step and patch 640 X rows are fetched for 256 results (2.5 x; the patch core of plane z + 6 and the arms of plane z),
//     no matrix entry is read at all.
// EVERY row takes star + diagonal from the sweep (star_build_host: A = S + D + R); what a row holds beyond that — the entries of the
// atom blocks, a star entry that differs from the coefficient — is the REMAINDER R, a CSR matrix of those rows alone, ADDED to the
// sweep's result by the kernels that follow it on the stream (dense blocks in layers + a pad-8 list, spmm_dense.hip).  One writer per
// row and launch: bit-reproducible.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <unordered_map>
#include <vector>
#include "gcge_hip_internal.h"

namespace gcge {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int STAR_R = 6;                       // arm length the kernel is built for (shorter stars: zero coefficients)
constexpr int STAR_T = 16;                      // patch edge
constexpr int STAR_PW = STAR_T + 2 * STAR_R;    // 28: patch with halo
constexpr int STAR_NP = STAR_PW * STAR_PW;      // slots of the plane image (corners unused)
constexpr int STAR_Q = 2 * STAR_R + 1;          // 13 planes in the register queue

struct StarCoef { double cx[STAR_R + 1], cy[STAR_R + 1], cz[STAR_R + 1]; };   // [k]: coefficient of the neighbours k steps away ([0] unused)
// Where the planes of the grid live in a block of vectors.  One rank: the slab is the grid.  A row slab of a partitioned matrix
// (planes [zs, ze) of nz; cuts on plane boundaries) finds the planes below and above among its halo rows: the halo rows are
// ascending by global index, so the up to R whole planes on either side are two contiguous runs of them — row of grid point
// (in-plane offset i, plane zz) = off + plane_rows * zz + i with off = mid_off inside the slab, lo_off below, hi_off above;
// planes outside [zmin, zmax) are outside the grid (Dirichlet: zero) or beyond the star's arms (coefficient zero): never loaded.
struct StarGeom { int nx, ny, nz, zs, ze, zmin, zmax; long lo_off, mid_off, hi_off; };
struct StarMat {
  StarGeom g; int R; bool iso; long nclean, nrows; StarCoef c; double* d_diag; unsigned char* d_clean;   // d_clean[local row]: 1 = nothing but the star and a diagonal
  int* d_map;   // masked grids: d_map[box index] = row, -1 where the point is not a row (NULL: every grid point is a row)
  int* d_prange;   // masked grids: for every 16 x 16 patch the first / last + 1 plane in which it has rows   // d_diag[local row]: the row's diagonal entry, NaN: not a clean row
  void* d_lines;   // masked grids whose lines are runs of rows: (base, xs | xe << 16) per line (y, z) — the third form (spmm_star3m_kernel); NULL: second form
  int* d_prange3;  // ... and the plane ranges of its 16 x 8 patches
  int* d_order3; int npatch3;   // ... and those of them that have rows, longest range first
  bool masked_slab;             // a row slab of a masked grid (halo rows among the lines: third form only)
  int shared_lo, shared_hi;     // ... 1: its first / last plane holds lines of the neighbouring slab as well (the cut runs between lines)
};

// staging plan of a thread: unit u = tid + 1024 q, point u >> 2, 16-byte part u & 3
struct StarUnit { int src; int dst; };   // src: in-plane row offset (x + nx y) or -1 (outside the grid: zero); dst: LDS index (v2d units), -1: no unit

__device__ __forceinline__ v2d star_ld(const double* __restrict__ x, size_t ldx, long row, int col) {
  return *reinterpret_cast<const v2d*>(x + (size_t)row * ldx + col);
}


// ---------------------------------------------------------------------------------------------- the sweep, second form
// Same patch (16 x 16 points, 8 columns, one z range), same register queue along z — another assignment of lanes and one
// barrier per plane instead of two:
//   * the 4 lanes that hold the 4 column pairs of a point are NEIGHBOURS in a wave, a wave is one grid line of the patch (16
//     points): a lane's own 16 bytes of plane z + 7 come straight from global memory into its queue register and its result goes
//     straight back (64-byte segments per point either way) — no "core in" / "result out" tiles in LDS;
//   * the plane image is point-major (64 B per point): what a wave reads for an x or y neighbour is ONE contiguous kilobyte —
//     the part-major image of the first form lost half of its LDS cycles to bank conflicts (4 line segments 448 B apart);
//   * two plane images: step z reads image z, writes image z + 1 (own values from the queue, halo strips loaded TWO steps earlier
//     into registers), one barrier separates the steps; 14 queue slots (13 planes live + the one in flight), unrolled 14 steps so
//     that every slot index and the image in use are compile-time constants;
//   * one coefficient set for all axes (ISO): the six neighbours at distance k are summed first, one multiply-add per k
//     (72 instead of 144 double-precision operations per step).
__device__ __forceinline__ v2d star2_and(v2d v, unsigned long long m) {   // m: all ones or zero
  typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
  u2 b = __builtin_bit_cast(u2, v);
  b.x &= m; b.y &= m;
  return __builtin_bit_cast(v2d, b);
}
constexpr int STAR2_Q = 14;
#define S2SLOT(U, k) (((U) + (k) + STAR_R + 2 * STAR2_Q) % STAR2_Q)
// LPP: lanes (of 16 bytes) per grid point = columns per pass / 2.  4: a 16 x 16 patch, 8 columns — 64-byte pieces of the rows;
// 8: a 16 x 8 patch, 16 columns — 128-byte pieces, i.e. whole cache lines.  tools/seg_bench.hip: on this chip 64-byte pieces of
// 512-byte rows come in at 3.1-3.9 TB/s (2.8 with the write-back) whatever is done, 128-byte pieces at 6.2 (5.4) TB/s, and the
// sweep's time is proportional to its traffic (tools/star_dbg_probe.py: LDS reads and barriers cost nothing, every X row fetched
// costs its share): the smaller patch stages 3.25 rows per row instead of 2.5 but moves them at the full rate.
template <int LPP> struct Star2Geom {
  static constexpr int TY = 64 / LPP;                    // lines of the patch (16 points each)
  static constexpr int ROWS = TY + 2 * STAR_R;           // image rows
  static constexpr int IMG = ROWS * STAR_PW * LPP;       // v2d entries of one plane image
  static constexpr int HPTS = 2 * STAR_R * TY + 2 * STAR_R * STAR_T;   // halo points per plane: side strips, then top / bottom strips
  static constexpr int HQ = (HPTS * LPP + 1023) / 1024;  // halo units per thread
};
// MAPPED: a masked grid (the points inside a sphere, say): only some points of the nx x ny x nz box are rows; map[box index] = the
// row of X / Y / diag of that point, -1 where there is none (reads as zero, nothing written).  Every access to a row then goes
// through a 4-byte lookup requested a step before the row itself is: one more stage in the same pipeline (one rank, no halo planes).
template <bool DOT, bool ISO, bool SLAB, int LPP, int dbg = 0, bool MAPPED = false>
__global__ __launch_bounds__(1024) void spmm_star2_kernel(int nx, int ny, int zs, int ze, int zmin, int zmax, long dlo, long dhi, StarCoef cf,
    const double* __restrict__ diag, const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int ncols,
    int zlo, int zhi, int zlen, int ntx, double* __restrict__ partial, const unsigned char* __restrict__ cleanf, const int* __restrict__ map,
    const int* __restrict__ prange) {
  typedef Star2Geom<LPP> GEO;
  constexpr int TY = GEO::TY, IMG = GEO::IMG, HQ = GEO::HQ, SIDE = 2 * STAR_R * TY;
  __shared__ v2d img[2 * IMG];                  // img[b][(row * 28 + col) * LPP + part]
  const int tid = threadIdx.x;
  const int part = tid % LPP, px = (tid / LPP) & 15, py = tid / (16 * LPP);
  const int tile_x = blockIdx.x % ntx, tile_y = blockIdx.x / ntx;
  const int x0 = tile_x * STAR_T, y0 = tile_y * TY;
  // MAPPED: only the planes in which this patch has rows at all (a ball fills half of its box: the rest would be swept for nothing)
  const int z0 = MAPPED ? max(zlo + (int)blockIdx.y * zlen, prange[2 * blockIdx.x]) : zlo + blockIdx.y * zlen;
  const int z1 = MAPPED ? min(min(zhi, zlo + ((int)blockIdx.y + 1) * zlen), prange[2 * blockIdx.x + 1]) : min(zhi, z0 + zlen);
  const int c0 = 2 * LPP * blockIdx.z;
  const long plane_rows = (long)nx * ny;
  auto plane_row0 = [&](int zz) -> long { return plane_rows * zz + (SLAB ? (zz < zs ? dlo : zz >= ze ? dhi : 0L) : 0L); };
  const int gx = x0 + px, gy = y0 + py;
  const bool cvalid = c0 + 2 * part < ncols;
  if (MAPPED && z0 >= z1) {          // (nothing of this patch in this z range: uniform over the workgroup, before any barrier)
    if (DOT && tid < LPP && c0 + 2 * tid < ncols) {
      double* out = partial + ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 2 * ncols;
      out[c0 + 2 * tid] = 0.0; out[c0 + 2 * tid + 1] = 0.0; out[ncols + c0 + 2 * tid] = 0.0; out[ncols + c0 + 2 * tid + 1] = 0.0;
    }
    return;
  }
  const int col = cvalid ? c0 + 2 * part : c0;                           // (a valid address whatever the lane: see star2_and)
  const bool inside = gx < nx && gy < ny && cvalid;
  const int own_i = inside ? gx + nx * gy : -1;                          // my point's offset inside a plane (< 0: none)
  const int slot = ((py + STAR_R) * STAR_PW + (px + STAR_R)) * LPP + part;
  // Every load below is UNCONDITIONAL at a clamped, valid address; what must read as zero (points outside the grid, planes
  // outside [zmin, zmax), columns past the block) is masked to zero bit-wise afterwards.  A predicate next to a load makes
  // hipcc branch round it and wait vmcnt(0) behind it ("no predicate anywhere near a load"): the requests of the
  // NEXT steps would be drained at every step.
#define own ((long)max(own_i, 0))
#define own_m (~(unsigned long long)(long)(own_i >> 31))
  // halo units of this thread: unit u = tid + 1024 q, point u / LPP, part u % LPP
  int hsrc[HQ], hdst[HQ];                        // hsrc < 0: reads as zero (outside the grid / the block's columns)
#pragma unroll
  for (int q = 0; q < HQ; ++q) {
    const int pt = (tid + 1024 * q) / LPP;
    int xx, yy;
    if (pt < SIDE) { const int ry = pt / 12, a = pt % 12; yy = ry; xx = a < 6 ? a - 6 : 10 + a; }
    else if (pt < GEO::HPTS) { const int j = pt - SIDE, d = j >> 4; xx = j & 15; yy = d < 6 ? d - 6 : TY - 6 + d; }
    else { xx = 0; yy = 0; }
    const int ax = x0 + xx, ay = y0 + yy;
    const bool ok = pt < GEO::HPTS && ax >= 0 && ax < nx && ay >= 0 && ay < ny && cvalid;
    hdst[q] = pt < GEO::HPTS ? ((yy + STAR_R) * STAR_PW + (xx + STAR_R)) * LPP + part : -1;
    hsrc[q] = ok ? ax + nx * ay : -1;
  }
  // lookups (MAPPED): the row of my point / of halo unit q in plane zz (clamped plane, a valid address whatever the lane)
  auto mp_own = [&](int zz) -> int { return map[plane_rows * min(max(zz, zmin), zmax - 1) + own]; };
  auto mp_halo = [&](int q, int zz) -> int { return map[plane_rows * min(max(zz, zmin), zmax - 1) + max(hsrc[q], 0)]; };
  auto zmask = [&](int zz) -> unsigned long long { return (zz >= zmin && zz < zmax) ? ~0ull : 0ull; };
  auto rmask = [&](int row) -> unsigned long long { return ~(unsigned long long)(long)(row >> 31); };   // all ones unless row < 0
  auto ld_own = [&](int zz) -> v2d {                                  // (not MAPPED)
    const int zc = min(max(zz, zmin), zmax - 1);
    return star2_and(star_ld(x, ldx, own + plane_row0(zc), col), own_m & zmask(zz));
  };
  auto ld_own_m = [&](int row, int zz) -> v2d {                       // MAPPED: row = mp_own(zz), looked up a step earlier
    return star2_and(star_ld(x, ldx, (long)max(row, 0), col), own_m & zmask(zz) & rmask(row));
  };
  auto ld_halo = [&](int q, int zz) -> v2d {
    const int zc = min(max(zz, zmin), zmax - 1);
    return star2_and(star_ld(x, ldx, (long)max(hsrc[q], 0) + plane_row0(zc), col), rmask(hsrc[q]) & zmask(zz));
  };
  auto ld_halo_m = [&](int q, int row, int zz) -> v2d {
    return star2_and(star_ld(x, ldx, (long)max(row, 0), col), rmask(hsrc[q]) & zmask(zz) & rmask(row));
  };
  // the diagonal of my point in plane zz (NaN: not my plane / no point of mine / no row there), same rule
  auto ld_diag_at = [&](long row, unsigned long long m) -> double {
    unsigned long long b = __builtin_bit_cast(unsigned long long, diag[row]);
    b = (b & m) | (0x7ff8000000000000ull & ~m);
    return __builtin_bit_cast(double, b);
  };
  auto ld_diag = [&](int zz) -> double {                              // (not MAPPED)
    const int zc = min(max(zz, z0), max(z1 - 1, z0));
    return ld_diag_at(own + plane_rows * zc, (zz < z1 ? ~0ull : 0ull) & own_m);
  };
  // ---- prologue: planes z0 - 6 .. z0 + 6 into slots 0 .. 12; image 0 = plane z0; halo strips of plane z0 + 1 in flight
  v2d qv[STAR2_Q];
#pragma unroll
  for (int t = 0; t < STAR2_Q - 1; ++t) qv[t] = MAPPED ? ld_own_m(mp_own(z0 - STAR_R + t), z0 - STAR_R + t) : ld_own(z0 - STAR_R + t);
  qv[STAR2_Q - 1] = v2d{0.0, 0.0};
  v2d st[HQ];                                    // halo strips in flight: requested at the end of a step, written to LDS at the end of the next
  int hrow[HQ];                                  // MAPPED: rows of the halo units in the plane requested NEXT (looked up a step ahead)
  {
#pragma unroll
    for (int q = 0; q < HQ; ++q) st[q] = MAPPED ? ld_halo_m(q, mp_halo(q, z0), z0) : ld_halo(q, z0);
    img[slot] = qv[STAR_R];
#pragma unroll
    for (int q = 0; q < HQ; ++q) if (hdst[q] >= 0) img[hdst[q]] = st[q];
#pragma unroll
    for (int q = 0; q < HQ; ++q) st[q] = MAPPED ? ld_halo_m(q, mp_halo(q, z0 + 1), z0 + 1) : ld_halo(q, z0 + 1);
#pragma unroll
    for (int q = 0; q < HQ; ++q) hrow[q] = MAPPED ? mp_halo(q, z0 + 2) : 0;
  }
  // MAPPED: rows of my point in the output plane and the next one (diag / clean / Y), and in plane z + 7 (the next X request)
  int rcur = MAPPED ? mp_own(z0) : 0, rnext = MAPPED ? mp_own(z0 + 1) : 0, r7 = MAPPED ? mp_own(z0 + STAR_R + 1) : 0;
  const unsigned long long in0 = (z0 < z1 ? ~0ull : 0ull) & own_m;
  double dg = MAPPED ? ld_diag_at((long)max(rcur, 0), in0 & rmask(rcur)) : ld_diag(z0);   // diagonal of my point in the output plane (NaN: none)
  // DOT: the sums run over the rows that are COMPLETE after the sweep (nothing but the star and a diagonal); the other rows get
  // the rest of their product from the kernels that follow, and their share of the sums from star_coldots2_rows
  auto ld_clean = [&](int zz) -> int { return DOT ? (int)cleanf[own + plane_rows * min(max(zz, z0), max(z1 - 1, z0))] : 1; };
  int cl = DOT ? (MAPPED ? (int)cleanf[max(rcur, 0)] : ld_clean(z0)) : 1;
  v2d spw = v2d{0.0, 0.0}, sww = v2d{0.0, 0.0};
  __syncthreads();

  for (int zb = z0; zb < z1; zb += STAR2_Q) {
#define STAR2_STEP(U)                                                                                                       \
    {                                                                                                                       \
      const int z = zb + (U);                                                                                               \
      if (z >= z1) break;                                                                                                   \
      constexpr int CUR = ((U) & 1) * IMG, NXT = (((U) + 1) & 1) * IMG;                                                     \
      /* request: my point in plane z + 7 (first used at step z + 1); MAPPED: and the lookups of the next step */          \
      if (!(dbg & 8)) qv[S2SLOT(U, STAR_R + 1)] = MAPPED ? ld_own_m(r7, z + STAR_R + 1) : ld_own(z + STAR_R + 1);            \
      int r7n = 0, rnn = 0;                                                                                                 \
      if (MAPPED) { r7n = mp_own(z + STAR_R + 2); rnn = mp_own(z + 2); }                                                    \
      const unsigned long long inn = (z + 1 < z1 ? ~0ull : 0ull) & own_m;                                                   \
      const double dnext = MAPPED ? ld_diag_at((long)max(rnext, 0), inn & rmask(rnext)) : ld_diag(z + 1);                   \
      const int cnext = DOT ? (MAPPED ? (int)cleanf[max(rnext, 0)] : ld_clean(z + 1)) : 1;                                  \
      const double d0 = dg == dg ? dg : 0.0;                                                                                \
      v2d acc = qv[S2SLOT(U, 0)] * d0;                                                                                      \
      const v2d* pl = img + CUR + slot;                                                                                     \
      _Pragma("unroll") for (int k = 1; k <= STAR_R; ++k) {                                                                 \
        const v2d zsum = qv[S2SLOT(U, -k)] + qv[S2SLOT(U, k)];                                                              \
        v2d xsum = v2d{0.0, 0.0}, ysum = v2d{0.0, 0.0};                                                                     \
        if (!(dbg & 2)) { xsum = pl[-LPP * k] + pl[LPP * k]; ysum = pl[-LPP * k * STAR_PW] + pl[LPP * k * STAR_PW]; }        \
        if (ISO) {                                                                                                          \
          const v2d s6 = (zsum + xsum) + ysum;                                                                              \
          acc.x = fma(cf.cz[k], s6.x, acc.x); acc.y = fma(cf.cz[k], s6.y, acc.y);                                            \
        } else {                                                                                                            \
          acc.x = fma(cf.cz[k], zsum.x, acc.x); acc.y = fma(cf.cz[k], zsum.y, acc.y);                                        \
          acc.x = fma(cf.cx[k], xsum.x, acc.x); acc.y = fma(cf.cx[k], xsum.y, acc.y);                                        \
          acc.x = fma(cf.cy[k], ysum.x, acc.x); acc.y = fma(cf.cy[k], ysum.y, acc.y);                                        \
        }                                                                                                                   \
      }                                                                                                                     \
      if (dg == dg && (!(dbg & 4) || acc.x == 12345.678)) {                                                                 \
        const size_t yrow = MAPPED ? (size_t)max(rcur, 0) : (size_t)(own + plane_rows * z);                                 \
        __builtin_nontemporal_store(acc, reinterpret_cast<v2d*>(y + yrow * ldy + col));                                     \
        if (DOT && cl != 0) {                                                                                               \
          const v2d xc = qv[S2SLOT(U, 0)];                                                                                  \
          spw.x = fma(xc.x, acc.x, spw.x); spw.y = fma(xc.y, acc.y, spw.y);                                                  \
          sww.x = fma(acc.x, acc.x, sww.x); sww.y = fma(acc.y, acc.y, sww.y);                                                \
        }                                                                                                                   \
      }                                                                                                                     \
      /* image of plane z + 1: my value from the queue, the strips requested a step ago; then the request for plane z + 2's */ \
      img[NXT + slot] = qv[S2SLOT(U, 1)];                                                                                   \
      _Pragma("unroll") for (int q = 0; q < HQ; ++q) if (hdst[q] >= 0) img[NXT + hdst[q]] = st[q];                           \
      if (!(dbg & 1)) { _Pragma("unroll") for (int q = 0; q < HQ; ++q) st[q] = MAPPED ? ld_halo_m(q, hrow[q], z + 2) : ld_halo(q, z + 2); } \
      if (MAPPED) { _Pragma("unroll") for (int q = 0; q < HQ; ++q) hrow[q] = mp_halo(q, z + 3); }                            \
      dg = dnext; cl = cnext;                                                                                               \
      if (MAPPED) { rcur = rnext; rnext = rnn; r7 = r7n; }                                                                  \
      if (!(dbg & 16)) __syncthreads();                                                                                     \
    }
    STAR2_STEP(0) STAR2_STEP(1) STAR2_STEP(2) STAR2_STEP(3) STAR2_STEP(4) STAR2_STEP(5) STAR2_STEP(6)
    STAR2_STEP(7) STAR2_STEP(8) STAR2_STEP(9) STAR2_STEP(10) STAR2_STEP(11) STAR2_STEP(12) STAR2_STEP(13)
#undef STAR2_STEP
  }
  if (DOT) {
    // fixed-order reduction over the points of every column pair: through the plane images (2 IMG >= 2 x 1024 v2d);
    // a lane's partner at distance h in POINT index is LPP h lanes away
    __syncthreads();
    img[tid] = spw; img[IMG + tid] = sww;
    __syncthreads();
    const int pidx = tid / LPP;
    for (int h = 512 / LPP; h > 0; h >>= 1) {
      if (pidx < h) {
        const v2d a = img[tid + LPP * h], b = img[IMG + tid + LPP * h];
        img[tid].x += a.x; img[tid].y += a.y; img[IMG + tid].x += b.x; img[IMG + tid].y += b.y;
      }
      __syncthreads();
    }
    if (pidx == 0 && cvalid) {
      double* out = partial + ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 2 * ncols;
      out[col] = img[tid].x; out[col + 1] = img[tid].y;
      out[ncols + col] = img[IMG + tid].x; out[ncols + col + 1] = img[IMG + tid].y;
    }
  }
}
#undef S2SLOT
#undef own
#undef own_m

// ---------------------------------------------------------------------------------------------- the sweep, third form
// 16-column passes (128-byte pieces of the rows: the full rate of the memory system, tools/seg_bench.hip) on a 16 x 8 patch, with
// what the second form could not afford in registers: TWO planes of prefetch.  The halo strips of plane z + 2 come by LDS-DMA
// (global_load_lds_dwordx4: no register is involved) into a ring of two strip buffers while plane z is computed; at the start of
// step z every thread copies its three units of ring slot z & 1 into the plane image (masking what must read as zero) next to
// its own value from the queue; the own point of plane z + 8 is requested at step z into a 15-slot register queue.  One plane
// image, two barriers per plane (barriers and LDS traffic are free here: profiles/r04_star/06_...).
// The DMA pieces and the compiler's own loads share the in-order vmcnt queue.  Hand-written waits count the MINIMUM number of
// operations issued since (a result store may be skipped by a wave: the stricter count is the safe one); hipcc's waits for its
// loads do not know the DMA pieces and so wait for MORE than they need — harmless as long as every load hipcc waits for was
// issued at least two steps before its use (the queue, the diagonal two planes ahead): the newest operations it then still
// allows in flight cover the pieces issued in the current step.  No spill may exist in this kernel (a scratch access is a
// vector-memory operation hipcc waits for at once).
constexpr int STAR3_Q = 15;
constexpr int STAR3_TY = 8, STAR3_LPP = 8;
constexpr int STAR3_ROWS = STAR3_TY + 2 * STAR_R;                           // 20 image rows
constexpr int STAR3_IMG = STAR3_ROWS * STAR_PW * STAR3_LPP;                 // v2d entries of the plane image (71 680 B)
constexpr int STAR3_HPTS = 2 * STAR_R * STAR3_TY + 2 * STAR_R * STAR_T;     // 288 halo points per plane
constexpr int STAR3_RING = STAR3_HPTS * STAR3_LPP;                          // v2d entries of one strip buffer (36 864 B)
constexpr int STAR3_PIECES = STAR3_RING / 64;                               // 36 pieces of 64 lanes x 16 B
constexpr unsigned STAR3_LDS = (STAR3_IMG + 2 * STAR3_RING) * 16;           // 145 408 B
#define S3SLOT(U, k) (((U) + (k) + STAR_R + 2 * STAR3_Q) % STAR3_Q)
__device__ __forceinline__ void star3_dma(const char* base, unsigned voff, unsigned lds_dst) {   // one piece: lane l -> lds_dst + 16 l
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// Workgroup -> (patch, z range, column pass) in an XCD-aware order (gcge_hip_spmm_star_xcd; OFF by default).  The dispatcher deals
// linear workgroup ids round-robin to the 8 XCDs, each with its own L2, and a patch's halo strips are its neighbours' own points —
// dealt in launch order, neighbouring patches sit on different XCDs and every strip comes over the fabric again (3.25 rows per
// row).  Tried: XCD k = id % 8 takes the k-th contiguous eighth of the patch-fastest order (xcd = 1), or runs of G consecutive
// patches dealt round-robin (xcd = G).  Measured at 171^3 x 64 (profiles/r04_star/16): launch order 2.84 ms, runs of 2 / 4 / 11 / 22
// 2.83 / 2.83 / 2.82 / 2.78, contiguous eighths 2.95 — the strips of neighbours on other XCDs are served by the memory-side cache
// at the rate the sweep consumes them; keeping them in one L2 buys nothing and contiguous eighths lose to channel imbalance.
__device__ __forceinline__ void star_block_of(int xcd, unsigned& bx, unsigned& by, unsigned& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if (!xcd) return;
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  const unsigned L = bx + gx * (by + gy * bz), k = L & 7u;
  unsigned v = L;
  if (xcd == 1) { const unsigned q = total >> 3, r = total & 7u; v = k * q + min(k, r) + (L >> 3); }
  else {                                                              // runs of `xcd` consecutive patches per XCD, the runs dealt round-robin
    const unsigned G = (unsigned)xcd, s = L >> 3, full = total / (8u * G) * (8u * G);
    if (L < full) v = (s / G) * 8u * G + k * G + s % G;
  }
  bx = v % gx; by = (v / gx) % gy; bz = v / (gx * gy);
}
template <int N> __device__ __forceinline__ void star3_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// PROBE (timing only, results wrong): the own points of plane z + 2 are requested at step z — at the lead of the halo strips — instead
// of those of plane z + 8: what a sweep that needs a plane's own points and its halo in the SAME step would request (tools/star_lead_probe.py)
template <bool DOT, bool SLAB, bool PROBE = false>
__global__ __launch_bounds__(1024) void spmm_star3_kernel(int nx, int ny, int zs, int ze, int zmin, int zmax, long dlo, long dhi, StarCoef cf,
    const double* __restrict__ diag, const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int ncols,
    int zlo, int zhi, int zlen, int ntx, double* __restrict__ partial, const unsigned char* __restrict__ cleanf, int xcd) {
  constexpr int LPP = STAR3_LPP, TY = STAR3_TY, SIDE = 2 * STAR_R * TY;
  extern __shared__ __align__(16) unsigned char star3_smem[];
  v2d* img = reinterpret_cast<v2d*>(star3_smem);                       // img[(row * 28 + col) * 8 + part]
  v2d* ring = img + STAR3_IMG;                                          // ring[slot][piece][lane]
  const unsigned ring0 = (unsigned)(uintptr_t)ring;                     // LDS byte address of the ring
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = tid % LPP, px = (tid / LPP) & 15, py = tid / (16 * LPP);
  unsigned bx, by, bz;
  star_block_of(xcd, bx, by, bz);
  const int tile_x = bx % ntx, tile_y = bx / ntx;
  const int x0 = tile_x * STAR_T, y0 = tile_y * TY;
  const int z0 = zlo + by * zlen, z1 = min(zhi, z0 + zlen);
  const int c0 = 2 * LPP * bz;
  const long plane_rows = (long)nx * ny;
  auto plane_row0 = [&](int zz) -> long { return plane_rows * zz + (SLAB ? (zz < zs ? dlo : zz >= ze ? dhi : 0L) : 0L); };
  const int gx = x0 + px, gy = y0 + py;
  const bool cvalid = c0 + 2 * part < ncols;
  const int col = cvalid ? c0 + 2 * part : c0;
  const bool inside = gx < nx && gy < ny && cvalid;
  const int own_i = inside ? gx + nx * gy : -1;
#define own ((long)max(own_i, 0))
#define own_m (~(unsigned long long)(long)(own_i >> 31))
  const int slot = ((py + STAR_R) * STAR_PW + (px + STAR_R)) * LPP + part;
  // my three halo units: piece wave + 16 j (the pieces past the 36th repeat earlier ones: every wave issues exactly three, so the
  // hand-counted waits hold for all of them), lane = my lane: unit u = piece * 64 + lane, point u / 8, part u % 8
  unsigned hoff[3], hlds[3]; int hdst[3]; unsigned hvalid = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int piece = wave + 16 * j; if (piece >= STAR3_PIECES) piece -= STAR3_PIECES;
    const int u = piece * 64 + lane, pt = u / LPP, hp = u % LPP;
    int xx, yy;
    if (pt < SIDE) { const int ry = pt / 12, a = pt % 12; yy = ry; xx = a < 6 ? a - 6 : 10 + a; }
    else { const int jj = pt - SIDE, d = jj >> 4; xx = jj & 15; yy = d < 6 ? d - 6 : TY - 6 + d; }
    const int ax = x0 + xx, ay = y0 + yy;
    const bool hc = c0 + 2 * hp < ncols;
    const bool ok = ax >= 0 && ax < nx && ay >= 0 && ay < ny && hc;
    hdst[j] = ((yy + STAR_R) * STAR_PW + (xx + STAR_R)) * LPP + hp;
    const long inplane = ok ? (long)ax + (long)nx * ay : 0;
    hoff[j] = (unsigned)((inplane * (long)ldx + (hc ? c0 + 2 * hp : c0)) * 8);   // byte offset inside a plane of X (< 4 GB: checked on the host)
    hlds[j] = (unsigned)(piece * 1024);                                            // piece base inside a strip buffer (wave-uniform)
    if (ok) hvalid |= 1u << j;
  }
  auto zmask = [&](int zz) -> unsigned long long { return (zz >= zmin && zz < zmax) ? ~0ull : 0ull; };
  auto ld_own = [&](int zz) -> v2d {
    const int zc = min(max(zz, zmin), zmax - 1);
    return star2_and(star_ld(x, ldx, own + plane_row0(zc), col), own_m & zmask(zz));
  };
  auto ld_diag = [&](int zz) -> double {
    const int zc = min(max(zz, z0), max(z1 - 1, z0));
    const unsigned long long m = (zz < z1 ? ~0ull : 0ull) & own_m;
    unsigned long long b = __builtin_bit_cast(unsigned long long, diag[own + plane_rows * zc]);
    b = (b & m) | (0x7ff8000000000000ull & ~m);
    return __builtin_bit_cast(double, b);
  };
  auto ld_clean = [&](int zz) -> int { return DOT ? (int)cleanf[own + plane_rows * min(max(zz, z0), max(z1 - 1, z0))] : 1; };
  // the three pieces of plane zz into strip buffer `sl` (clamped plane: what must be zero is masked at the copy)
  auto dma_plane = [&](int zz, int sl) {
    const int zc = min(max(zz, zmin), zmax - 1);
    const char* base = reinterpret_cast<const char*>(x + (size_t)plane_row0(zc) * ldx);
    const unsigned dst = ring0 + (unsigned)sl * (STAR3_RING * 16);
#pragma unroll
    for (int j = 0; j < 3; ++j) star3_dma(base, hoff[j], dst + __builtin_amdgcn_readfirstlane(hlds[j]));
  };
  // strip buffer `sl` (plane zz) -> plane image
  auto copy_strips = [&](int zz, int sl) {
    const unsigned long long zm = zmask(zz);
    const v2d* src = ring + sl * STAR3_RING;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const v2d v = src[(hlds[j] >> 4) + lane];
      img[hdst[j]] = star2_and(v, zm & (((hvalid >> j) & 1u) ? ~0ull : 0ull));
    }
  };
  // ---- prologue: planes z0 - 6 .. z0 + 7 into slots 0 .. 13 (z0 + 7 is first used at step 1); strips of z0, z0 + 1 by DMA
  v2d qv[STAR3_Q];
#pragma unroll
  for (int t = 0; t < STAR3_Q - 1; ++t) qv[t] = ld_own(z0 - STAR_R + t);
  qv[STAR3_Q - 1] = v2d{0.0, 0.0};
  dma_plane(z0, 0);
  dma_plane(z0 + 1, 1);
  // the diagonal (NaN: none) and the clean flag of my point: three slots, plane z + k in slot (U + k) % 3 — renamed by the
  // unrolling like the queue (a register MOVE of a value requested in this step would make hipcc wait for it in this step)
  double dg3[3]; int cl3[3];
  dg3[0] = ld_diag(z0); dg3[1] = ld_diag(z0 + 1); dg3[2] = 0.0;
  cl3[0] = ld_clean(z0); cl3[1] = ld_clean(z0 + 1); cl3[2] = 0;
  v2d spw = v2d{0.0, 0.0}, sww = v2d{0.0, 0.0};
  star3_vmwait<0>();
  __syncthreads();

  for (int zb = z0; zb < z1; zb += STAR3_Q) {
#define STAR3_STEP(U)                                                                                                       \
    {                                                                                                                       \
      const int z = zb + (U);                                                                                               \
      if (z >= z1) break;                                                                                                   \
      const int sl = (z - z0) & 1;                                                                                          \
      /* the strips of plane z have landed: issued two steps ago; since then at least 2 + 3 + 2 operations (own request,      \
         diagonal [, clean flag], the next plane's three pieces, and again) */                                              \
      if ((U) == 0 && zb == z0) { /* (the prologue waited for everything) */ }                                              \
      else star3_vmwait<DOT ? 9 : 7>();                                                                                     \
      copy_strips(z, sl);                                                                                                   \
      img[slot] = qv[S3SLOT(U, 0)];                                                                                         \
      __syncthreads();                                   /* B: the image of plane z is complete, strip buffer sl is free */ \
      dma_plane(z + 2, sl);                                                                                                 \
      qv[S3SLOT(U, STAR_R + 2)] = ld_own(z + (PROBE ? 2 : STAR_R + 2));      /* first used at step z + 2 */                 \
      dg3[((U) + 2) % 3] = ld_diag(z + 2);                                                                                  \
      cl3[((U) + 2) % 3] = ld_clean(z + 2);                                                                                 \
      const double dg = dg3[(U) % 3];                                                                                       \
      const int cl = cl3[(U) % 3];                                                                                          \
      const double d0 = dg == dg ? dg : 0.0;                                                                                \
      v2d acc = qv[S3SLOT(U, 0)] * d0;                                                                                      \
      const v2d* pl = img + slot;                                                                                           \
      _Pragma("unroll") for (int k = 1; k <= STAR_R; ++k) {                                                                 \
        const v2d zsum = qv[S3SLOT(U, -k)] + qv[S3SLOT(U, k)];                                                              \
        const v2d xsum = pl[-LPP * k] + pl[LPP * k];                                                                        \
        const v2d ysum = pl[-LPP * k * STAR_PW] + pl[LPP * k * STAR_PW];                                                    \
        const v2d s6 = (zsum + xsum) + ysum;                                                                                \
        acc.x = fma(cf.cz[k], s6.x, acc.x); acc.y = fma(cf.cz[k], s6.y, acc.y);                                              \
      }                                                                                                                     \
      if (dg == dg) {                                                                                                       \
        __builtin_nontemporal_store(acc, reinterpret_cast<v2d*>(y + (size_t)(own + plane_rows * z) * ldy + col));           \
        if (DOT && cl != 0) {                                                                                               \
          const v2d xc = qv[S3SLOT(U, 0)];                                                                                  \
          spw.x = fma(xc.x, acc.x, spw.x); spw.y = fma(xc.y, acc.y, spw.y);                                                  \
          sww.x = fma(acc.x, acc.x, sww.x); sww.y = fma(acc.y, acc.y, sww.y);                                                \
        }                                                                                                                   \
      }                                                                                                                     \
      __syncthreads();                                   /* A: everybody has read the image of plane z */                   \
    }
    STAR3_STEP(0) STAR3_STEP(1) STAR3_STEP(2) STAR3_STEP(3) STAR3_STEP(4) STAR3_STEP(5) STAR3_STEP(6) STAR3_STEP(7)
    STAR3_STEP(8) STAR3_STEP(9) STAR3_STEP(10) STAR3_STEP(11) STAR3_STEP(12) STAR3_STEP(13) STAR3_STEP(14)
#undef STAR3_STEP
  }
  star3_vmwait<0>();                                     // no piece may land after the block has released its LDS
  if (DOT) {
    __syncthreads();
    img[tid] = spw; img[1024 + tid] = sww;
    __syncthreads();
    const int pidx = tid / LPP;
    for (int h = 512 / LPP; h > 0; h >>= 1) {
      if (pidx < h) {
        const v2d a = img[tid + LPP * h], b = img[1024 + tid + LPP * h];
        img[tid].x += a.x; img[tid].y += a.y; img[1024 + tid].x += b.x; img[1024 + tid].y += b.y;
      }
      __syncthreads();
    }
    if (pidx == 0 && cvalid) {
      double* out = partial + ((size_t)bx + (size_t)gridDim.x * by) * 2 * ncols;
      out[col] = img[tid].x; out[col + 1] = img[tid].y;
      out[ncols + col] = img[1024 + tid].x; out[ncols + col + 1] = img[1024 + tid].y;
    }
  }
}
#undef S3SLOT
#undef own
#undef own_m

// ---------------------------------------------------------------------------------------------- the third form on a masked grid
// The points of a convex domain inside the box, numbered in scan order (x fastest): every grid LINE (y, z) is one run of rows, so
//     row of point (x, y, z) = base(y, z) + x   for xs(y, z) <= x < xe(y, z),   no row elsewhere
// and a table of (base, xs | xe << 16) per line replaces the point-wise map of the second form.  The waves are laid out so that
// everything a wave touches in one request lies on ONE line — its own 8 points (a wave is half a patch line), and each of its three
// DMA pieces (a side strip of one patch line: 6 points, 48 of the 64 lanes; half a line of the top / bottom strips: 8 points) —
// hence every lookup is wave-uniform: a SCALAR load (s_load_dwordx2, the scalar cache, lgkmcnt), no vector register, no entry in
// the vmcnt queue the hand-counted waits of the third form rely on.  40 pieces per plane instead of 36 (the side strips take a
// piece per line); lanes without a row request row 0 and are masked when the strips are copied into the plane image, as before.
struct StarLine { int base; int xr; };            // xr = xs | xe << 16; an empty line: 0
constexpr int STAR3M_ZLO = STAR_R, STAR3M_ZHI = STAR_R + 3, STAR3M_YLO = STAR_R, STAR3M_YHI = STAR3_TY + STAR_R;   // guard lines of the table
constexpr int STAR3M_PIECES = 40;
constexpr int STAR3M_RING = STAR3M_PIECES * 64;                              // v2d entries of one strip buffer (40 960 B)
constexpr unsigned STAR3M_LDS = (STAR3_IMG + 2 * STAR3M_RING) * 16;          // 153 600 B
__device__ __forceinline__ void star3_dma64(const void* addr, unsigned lds_dst) {   // one piece: lane l -> lds_dst + 16 l, 64-bit lane addresses
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(addr), "s"(lds_dst) : "memory");
}
#define S3SLOT(U, k) (((U) + (k) + STAR_R + 2 * STAR3_Q) % STAR3_Q)
template <bool DOT>
__global__ __launch_bounds__(1024) void spmm_star3m_kernel(int nx, int ny, int nz, StarCoef cf, const double* __restrict__ diag,
    const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int ncols, int zlo, int zhi, int zlen, int ntx,
    double* __restrict__ partial, const unsigned char* __restrict__ cleanf, const unsigned long long* __restrict__ lines,
    const int* __restrict__ prange, const int* __restrict__ order, int nown) {
  constexpr int LPP = STAR3_LPP, TY = STAR3_TY;
  const unsigned ldx32 = (unsigned)ldx;                                   // (< 2^31: checked on the host) row * ldx in one v_mad_u64_u32
  const int nyp = ny + STAR3M_YLO + STAR3M_YHI;
  extern __shared__ __align__(16) unsigned char star3_smem[];
  v2d* img = reinterpret_cast<v2d*>(star3_smem);
  v2d* ring = img + STAR3_IMG;
  const unsigned ring0 = (unsigned)(uintptr_t)ring;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // a wave is half a patch line: lane l is point 8 (wave & 1) + (l >> 3) of line wave >> 1, part l & 7 — what depends on the wave is
  // kept in scalar registers and added where it is used (there is no vector register to spare: 128 of 128)
  const int part = lane & 7, pt = lane >> 3;
  const int py = wave >> 1, pxs = 8 * (wave & 1);
  // workgroup -> patch: the patches that have rows at all, LONGEST z range first (the upload sorts them) — the patches of a ball
  // differ in length by the chord of the sphere, and in launch order the last workgroups to start were among the longest
  unsigned bq, by, bz;
  star_block_of(0, bq, by, bz);
  const int bx = order[bq];
  const int tile_x = bx % ntx, tile_y = bx / ntx;
  const int x0 = tile_x * STAR_T, y0 = tile_y * TY;
  const int z0 = max(zlo + (int)by * zlen, prange[2 * bx]);
  const int z1 = min(min(zhi, zlo + ((int)by + 1) * zlen), prange[2 * bx + 1]);
  const int c0 = 2 * LPP * bz;
  const bool cvalid = c0 + 2 * part < ncols;
  const int col = cvalid ? c0 + 2 * part : c0;
  if (z0 >= z1) {                                                         // nothing of this patch in this z range (uniform, before any barrier)
    if (DOT && tid < LPP && c0 + 2 * tid < ncols) {
      double* out = partial + ((size_t)bq + (size_t)gridDim.x * by) * 2 * ncols;
      out[c0 + 2 * tid] = 0.0; out[c0 + 2 * tid + 1] = 0.0; out[ncols + c0 + 2 * tid] = 0.0; out[ncols + c0 + 2 * tid + 1] = 0.0;
    }
    return;
  }
  const int gxs = x0 + pxs, gy = y0 + py;                                 // my point: (gxs + pt, gy)
  const int gx = gxs + pt;
  const bool inside = gx < nx && cvalid;
  const int sbase = ((py + STAR_R) * STAR_PW + (pxs + STAR_R)) * LPP;     // my slot of the plane image: sbase + lane
  const int slot = sbase + lane;
  // the line (yy, zz) — wave-uniform arguments.  The table carries empty guard lines round the box (STAR3M_ZLO planes below, _ZHI
  // above, _YLO / _YHI lines): whatever a step asks for is one 8-byte scalar load, no clamp, no select
  auto line_at = [&](int zz, int yy) -> StarLine {
    const unsigned long long raw = lines[(size_t)(zz + STAR3M_ZLO) * nyp + (yy + STAR3M_YLO)];
    return StarLine{(int)(unsigned)raw, (int)(unsigned)(raw >> 32)};
  };
  auto on_line = [&](const StarLine& L, int xx) -> bool { return xx >= (L.xr & 0xffff) && xx < (L.xr >> 16); };
  // my three pieces: 0 = a side strip of patch line wave / 2 (left: even waves), 1 and 2 = halves of the lines above / below.
  // Lane l of a piece is point l >> 3 of it, part l & 7: everything else about a piece is wave-uniform (scalar registers) —
  //   hya: its line's y in the grid, hxa: the x of its first point, hu: where its first point sits in the plane image, hld: in a strip buffer
  int hya[3], hxa[3], hu[3]; unsigned hld[3];
  {
    const int q1 = wave, q2 = 16 + (wave & 7);                            // (waves 8 .. 15 repeat pieces 32 .. 39: every wave issues exactly three)
    const int d1 = q1 >> 1, d2 = q2 >> 1;
    const int hy0 = wave >> 1, hy1 = d1 < 6 ? d1 - 6 : d1 + 2, hy2 = d2 < 6 ? d2 - 6 : d2 + 2;
    const int hx0 = (wave & 1) ? STAR_T : -STAR_R, hx1 = 8 * (q1 & 1), hx2 = 8 * (q2 & 1);
    hya[0] = y0 + hy0; hya[1] = y0 + hy1; hya[2] = y0 + hy2;
    hxa[0] = x0 + hx0; hxa[1] = x0 + hx1; hxa[2] = x0 + hx2;
    hu[0] = ((hy0 + STAR_R) * STAR_PW + (hx0 + STAR_R)) * LPP; hu[1] = ((hy1 + STAR_R) * STAR_PW + (hx1 + STAR_R)) * LPP; hu[2] = ((hy2 + STAR_R) * STAR_PW + (hx2 + STAR_R)) * LPP;
    hld[0] = (unsigned)wave * 64u; hld[1] = (unsigned)(16 + q1) * 64u; hld[2] = (unsigned)(16 + q2) * 64u;   // (v2d units)
  }
  auto ld_own = [&](int zz, const StarLine& L) -> v2d {
    const bool ok = inside && on_line(L, gx);
    const int row = ok ? L.base + gx : 0;
    return star2_and(*reinterpret_cast<const v2d*>(x + ((size_t)(unsigned)row * ldx32 + col)), ok ? ~0ull : 0ull);
  };
  // (row < nown: a line of the output plane may belong to the neighbouring slab — read like any halo row, never written)
  auto ld_diag = [&](int zz, const StarLine& L) -> double {
    const bool ok = inside && on_line(L, gx) && zz < z1 && L.base + gx < nown;
    const int row = ok ? L.base + gx : 0;
    unsigned long long b = __builtin_bit_cast(unsigned long long, diag[row]);
    const unsigned long long m = ok ? ~0ull : 0ull;
    b = (b & m) | (0x7ff8000000000000ull & ~m);
    return __builtin_bit_cast(double, b);
  };
  // (the flag is requested and the sums are formed without DOT as well: hipcc's register allocation of the leaner variant spilled
  //  four vector registers, and a scratch access is a vector-memory operation the hand-counted waits do not know)
  auto ld_clean = [&](int zz, const StarLine& L) -> int {
    const bool ok = inside && on_line(L, gx) && zz < z1 && L.base + gx < nown;
    return (int)cleanf[ok ? L.base + gx : 0];
  };
  // the three pieces of a plane into strip buffer `sl`; L: the lines they lie on in that plane
  auto dma_plane = [&](int sl, const StarLine (&L)[3]) {
    const unsigned dst = ring0 + (unsigned)sl * (STAR3M_RING * 16);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ax = hxa[j] + pt;
      const int row = on_line(L[j], ax) ? L[j].base + ax : 0;
      star3_dma64(x + ((size_t)(unsigned)row * ldx32 + col), dst + hld[j] * 16u);
    }
  };
  auto copy_strips = [&](int sl, const StarLine (&L)[3]) {
    const v2d* src = ring + sl * STAR3M_RING + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const v2d v = src[hld[j]];
      const bool ok = cvalid && on_line(L[j], hxa[j] + pt);
      if (j > 0 || pt < STAR_R) img[hu[j] + lane] = star2_and(v, ok ? ~0ull : 0ull);   // (lanes 48 .. 63 of a side piece carry nothing)
    }
  };
  // ---- prologue
  v2d qv[STAR3_Q];
#pragma unroll
  for (int t = 0; t < STAR3_Q - 1; ++t) { const StarLine L = line_at(z0 - STAR_R + t, gy); qv[t] = ld_own(z0 - STAR_R + t, L); }
  qv[STAR3_Q - 1] = v2d{0.0, 0.0};
  { const StarLine La[3] = {line_at(z0, hya[0]), line_at(z0, hya[1]), line_at(z0, hya[2])};
    const StarLine Lb[3] = {line_at(z0 + 1, hya[0]), line_at(z0 + 1, hya[1]), line_at(z0 + 1, hya[2])};
    dma_plane(0, La);
    dma_plane(1, Lb); }
  double dg3[3]; int cl3[3];
  { const StarLine L0 = line_at(z0, gy), L1 = line_at(z0 + 1, gy);
    dg3[0] = ld_diag(z0, L0); dg3[1] = ld_diag(z0 + 1, L1); dg3[2] = 0.0;
    cl3[0] = ld_clean(z0, L0); cl3[1] = ld_clean(z0 + 1, L1); cl3[2] = 0; }
  v2d spw = v2d{0.0, 0.0}, sww = v2d{0.0, 0.0};
  star3_vmwait<0>();
  __syncthreads();

  for (int zb = z0; zb < z1; zb += STAR3_Q) {
#define STAR3M_STEP(U)                                                                                                      \
    {                                                                                                                       \
      const int z = zb + (U);                                                                                               \
      if (z >= z1) break;                                                                                                   \
      const int sl = (z - z0) & 1;                                                                                          \
      /* the step's lookups first (scalar loads: their latency passes behind the wait and the copy): the lines of my pieces in  \
         plane z (masks of the copy) and z + 2 (the next request), of my own point in z (the row of Y), z + 2 and z + 8 */     \
      const StarLine Lc[3] = {line_at(z, hya[0]), line_at(z, hya[1]), line_at(z, hya[2])};                                  \
      const StarLine Ln[3] = {line_at(z + 2, hya[0]), line_at(z + 2, hya[1]), line_at(z + 2, hya[2])};                      \
      const StarLine L0 = line_at(z, gy), L2 = line_at(z + 2, gy), L8 = line_at(z + STAR_R + 2, gy);                        \
      if ((U) == 0 && zb == z0) { /* (the prologue waited for everything) */ }                                              \
      else star3_vmwait<9>();                            /* as in the third form: 3 pieces + own + diagonal + flag per step */ \
      copy_strips(sl, Lc);                                                                                                  \
      img[slot] = qv[S3SLOT(U, 0)];                                                                                         \
      __syncthreads();                                                                                                      \
      dma_plane(sl, Ln);                                                                                                    \
      qv[S3SLOT(U, STAR_R + 2)] = ld_own(z + STAR_R + 2, L8);                                                               \
      dg3[((U) + 2) % 3] = ld_diag(z + 2, L2);                                                                              \
      cl3[((U) + 2) % 3] = ld_clean(z + 2, L2);                                                                             \
      const double dg = dg3[(U) % 3];                                                                                       \
      const int cl = cl3[(U) % 3];                                                                                          \
      const double d0 = dg == dg ? dg : 0.0;                                                                                \
      v2d acc = qv[S3SLOT(U, 0)] * d0;                                                                                      \
      const v2d* pl = img + slot;                                                                                           \
      _Pragma("unroll") for (int k = 1; k <= STAR_R; ++k) {                                                                 \
        const v2d zsum = qv[S3SLOT(U, -k)] + qv[S3SLOT(U, k)];                                                              \
        const v2d xsum = pl[-LPP * k] + pl[LPP * k];                                                                        \
        const v2d ysum = pl[-LPP * k * STAR_PW] + pl[LPP * k * STAR_PW];                                                    \
        const v2d s6 = (zsum + xsum) + ysum;                                                                                \
        acc.x = fma(cf.cz[k], s6.x, acc.x); acc.y = fma(cf.cz[k], s6.y, acc.y);                                              \
      }                                                                                                                     \
      if (dg == dg) {                                                                                                       \
        const size_t yrow = (size_t)(unsigned)(L0.base + gx);                                                               \
        __builtin_nontemporal_store(acc, reinterpret_cast<v2d*>(y + yrow * ldy + col));                                     \
        if (cl != 0) {                        /* (without DOT the sums are formed and dropped: see ld_clean) */              \
          const v2d xc = qv[S3SLOT(U, 0)];                                                                                  \
          spw.x = fma(xc.x, acc.x, spw.x); spw.y = fma(xc.y, acc.y, spw.y);                                                  \
          sww.x = fma(acc.x, acc.x, sww.x); sww.y = fma(acc.y, acc.y, sww.y);                                                \
        }                                                                                                                   \
      }                                                                                                                     \
      __syncthreads();                                                                                                      \
    }
    STAR3M_STEP(0) STAR3M_STEP(1) STAR3M_STEP(2) STAR3M_STEP(3) STAR3M_STEP(4) STAR3M_STEP(5) STAR3M_STEP(6) STAR3M_STEP(7)
    STAR3M_STEP(8) STAR3M_STEP(9) STAR3M_STEP(10) STAR3M_STEP(11) STAR3M_STEP(12) STAR3M_STEP(13) STAR3M_STEP(14)
#undef STAR3M_STEP
  }
  star3_vmwait<0>();                                     // no piece may land after the block has released its LDS
  if (!DOT) asm volatile("" : : "v"(spw.x), "v"(spw.y), "v"(sww.x), "v"(sww.y));
  if (DOT) {
    __syncthreads();
    const int t2 = 64 * wave + lane;                     // (= threadIdx.x, rebuilt: not kept in a register through the sweep)
    img[t2] = spw; img[1024 + t2] = sww;
    __syncthreads();
    const int pidx = t2 / LPP;
    for (int h = 512 / LPP; h > 0; h >>= 1) {
      if (pidx < h) {
        const v2d a = img[t2 + LPP * h], b = img[1024 + t2 + LPP * h];
        img[t2].x += a.x; img[t2].y += a.y; img[1024 + t2].x += b.x; img[1024 + t2].y += b.y;
      }
      __syncthreads();
    }
    if (pidx == 0 && cvalid) {
      double* out = partial + ((size_t)bq + (size_t)gridDim.x * by) * 2 * ncols;
      out[col] = img[t2].x; out[col + 1] = img[t2].y;
      out[ncols + col] = img[1024 + t2].x; out[ncols + col + 1] = img[1024 + t2].y;
    }
  }
}
#undef S3SLOT

// partial[b * 2 m + j] = sum over the block's listed rows of x[r, j] y[r, j]; at + m: of y[r, j]^2 (rows = list[i]); 256 threads =
// 4 row lanes x 64 columns, as coldots2_partial of vec_kernels.hip
__global__ __launch_bounds__(256) void star_coldots2_rows(int nlist, const int* __restrict__ list, const double* __restrict__ x, size_t ldx,
    const double* __restrict__ y, size_t ldy, int m, double* __restrict__ partial, int per_block) {
  __shared__ double red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i0 = blockIdx.x * per_block, i1 = min(nlist, i0 + per_block);
  for (int c0 = 0; c0 < m; c0 += 64) {
    const int j = c0 + tx;
    double s = 0.0, q = 0.0;
    if (j < m)
      for (int i = i0 + ty; i < i1; i += 4) {
        const size_t r = (size_t)list[i];
        const double yv = y[r * ldy + j];
        s = fma(x[r * ldx + j], yv, s); q = fma(yv, yv, q);
      }
    red[0][ty][tx] = s; red[1][ty][tx] = q;
    __syncthreads();
    if (ty == 0 && j < m) {
      partial[(size_t)blockIdx.x * 2 * m + j] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
      partial[(size_t)blockIdx.x * 2 * m + m + j] = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- upload-time analysis
struct StarHost {
  StarGeom g; int R = 0; StarCoef c; long nclean = 0;
  std::vector<double> diag;                                           // NaN: row stays in the remainder
  std::vector<char> clean;                                            // 1: star row
  std::vector<int> inv;                                               // masked domains: grid point -> row (-1: none)
  std::vector<int> own_box;                                           // masked domains whose geometry was recovered here (star_infer_box)
  std::vector<int> rem_rowptr, rem_col; std::vector<double> rem_val;  // every entry of the rows that are not clean (local columns)
};

static inline uint64_t star_bits(double v) { uint64_t b; memcpy(&b, &v, 8); return b; }

// The rows of one slab: local rows [0, nrows) are global rows row_begin + r; local column c < nrows is global column row_begin + c,
// c >= nrows is halo row c - nrows = global column ghost[c - nrows] (ascending).  One rank: row_begin = 0, no halo columns.
// With a geometry (box != NULL; one rank, no halo columns): row r is grid point box[r] = x + bnx (y + bny z) of a bnx x bny x bnz
// box of which only some points are rows (a masked domain: the grid points inside a sphere, say) — "global index" then means
// the box index, and nglobal the size of the box.
struct StarRows {
  int nrows, ncols_local; long row_begin, nglobal; const int* ghost; const int *rowptr, *colidx; const double* val;
  bool global_cols = false;                                           // colidx holds global columns already (partitioners)
  const int* box = nullptr; int bnx = 0, bny = 0, bnz = 0;
  bool box_cols = false;                                              // box[] covers every LOCAL column (a row slab of a masked grid: own rows, then halo rows)
  long gcol(int c) const { return box != nullptr ? (long)box[c] : global_cols ? (long)c : c < nrows ? row_begin + c : (long)ghost[c - nrows]; }
  long gpos(int r) const { return box != nullptr ? (long)box[r] : row_begin + r; }
};

// The offsets (global column - global row) of a star on a lexicographic grid: |o| in {1..Rx} u {sy, 2 sy, .., Ry sy} u {sz, .., Rz sz},
// each carried (as + o or - o: rows next to a face of the grid, or of a slab at the end of the grid, have only one of the two)
// by at least half of the sampled rows.  false: the frequent offsets are not such a set.
static bool star_offsets(const StarRows& M, int* Rx_, int* Ry_, int* Rz_, long* sy_, long* sz_) {
  const int nrows = M.nrows;
  if (nrows < 2048) return false;
  const int nsamp = std::min(nrows, 8192);
  std::unordered_map<long, int> hist;
  std::vector<long> seen;
  for (int t = 0; t < nsamp; ++t) {
    const int r = (int)((long)t * nrows / nsamp);
    seen.clear();
    for (int q = M.rowptr[r]; q < M.rowptr[r + 1]; ++q) { const long o = M.gcol(M.colidx[q]) - M.gpos(r); if (o != 0) seen.push_back(o < 0 ? -o : o); }
    std::sort(seen.begin(), seen.end());
    seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
    for (long o : seen) ++hist[o];
  }
  std::vector<long> offs;
  for (auto& kv : hist) if (kv.second * 2 >= nsamp) offs.push_back(kv.first);
  std::sort(offs.begin(), offs.end());
  if (offs.empty() || offs[0] != 1) return false;
  // runs: 1 .. Rx, then sy, 2 sy, .. Ry sy, then sz, 2 sz, .. Rz sz
  size_t i = 0; int Rx = 0, Ry = 0, Rz = 0; long sy = 0, sz = 0;
  while (i < offs.size() && offs[i] == Rx + 1) { ++Rx; ++i; }
  if (i >= offs.size()) return false;
  sy = offs[i];
  while (i < offs.size() && offs[i] == (long)(Ry + 1) * sy) { ++Ry; ++i; }
  if (i >= offs.size()) return false;
  sz = offs[i];
  while (i < offs.size() && offs[i] == (long)(Rz + 1) * sz) { ++Rz; ++i; }
  if (i != offs.size()) return false;                                  // a frequent offset that is not part of a star
  if (sz % sy != 0 || M.nglobal % sz != 0) return false;
  *Rx_ = Rx; *Ry_ = Ry; *Rz_ = Rz; *sy_ = sy; *sz_ = sz;
  return true;
}

// grid strides and the star's coefficients from the offsets (global column - global row) most rows share; false: no star on a grid here
static bool star_detect(const StarRows& M, StarHost* H) {
  const int nrows = M.nrows; const int* rowptr = M.rowptr; const int* colidx = M.colidx; const double* val = M.val;
  int Rx = 0, Ry = 0, Rz = 0; long sy = 0, sz = 0;
  if (!star_offsets(M, &Rx, &Ry, &Rz, &sy, &sz)) return false;
  const int nsamp = std::min(nrows, 8192);
  const int R = std::max(Rx, std::max(Ry, Rz));
  if (R > STAR_R || sy <= 2L * STAR_R || sz % sy != 0 || sz / sy <= 2L * STAR_R || M.nglobal % sz != 0 || M.nglobal / sz < 2) return false;
  StarGeom& g = H->g;
  g.nx = (int)sy; g.ny = (int)(sz / sy); g.nz = (int)(M.nglobal / sz); H->R = R;
  if (M.box != nullptr) {            // a masked domain: the box is the grid, the rows are found through the map / the line table
    if (g.nx != M.bnx || g.ny != M.bny || g.nz != M.bnz || (M.ncols_local != nrows && !M.box_cols)) return false;
    g.zs = 0; g.ze = g.nz;
    if (M.box_cols) { g.zs = (int)(M.box[0] / sz); g.ze = (int)(M.box[nrows - 1] / sz) + 1; }   // a slab: the planes its own rows touch
  } else {
    // the slab must be whole planes (a partition cut inside a plane keeps the other forms: gcge_amd.dist.partition_by_nnz(align=))
    if (M.row_begin % sz != 0 || (long)nrows % sz != 0) return false;
    g.zs = (int)(M.row_begin / sz); g.ze = g.zs + (int)((long)nrows / sz);
  }
  if (g.ze > g.nz) return false;
  g.zmin = std::max(0, g.zs - R); g.zmax = std::min(g.nz, g.ze + R);
  g.mid_off = -sz * g.zs; g.lo_off = g.hi_off = 0;
  // the planes below and above among the halo rows: whole and contiguous (every point of them is some row's neighbour)
  const int ng = M.ncols_local - nrows;
  auto run = [&](long first, long count, long* off) -> bool {       // halo rows first .. first + count - 1 -> *off + global = row of X
    if (count == 0) return true;
    if (M.ghost == nullptr) return false;
    const int* lo = std::lower_bound(M.ghost, M.ghost + ng, (int)first);
    const long idx = lo - M.ghost;
    if (idx + count > ng) return false;
    if (M.ghost[idx] != first || M.ghost[idx + count - 1] != first + count - 1) return false;   // (ascending and unique: the run is gap-free)
    *off = (long)nrows + idx - first;
    return true;
  };
  if (M.box == nullptr) {             // (a masked grid finds its halo rows through the line table, line by line)
    if (!run((long)g.zmin * sz, (long)(g.zs - g.zmin) * sz, &g.lo_off)) return false;
    if (!run((long)g.ze * sz, (long)(g.zmax - g.ze) * sz, &g.hi_off)) return false;
  }
  // coefficients: the most frequent value of every offset among the sampled rows; the star must be symmetric
  memset(&H->c, 0, sizeof(H->c));
  for (int axis = 0; axis < 3; ++axis) {
    const int Ra = axis == 0 ? Rx : axis == 1 ? Ry : Rz;
    const long stride = axis == 0 ? 1 : axis == 1 ? sy : sz;
    double* dst = axis == 0 ? H->c.cx : axis == 1 ? H->c.cy : H->c.cz;
    for (int k = 1; k <= Ra; ++k) {
      std::unordered_map<uint64_t, int> vals[2];
      for (int t = 0; t < nsamp; ++t) {
        const int r = (int)((long)t * nrows / nsamp);
        for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
          const long o = M.gcol(colidx[q]) - M.gpos(r);
          if (o == k * stride) ++vals[0][star_bits(val[q])];
          else if (o == -k * stride) ++vals[1][star_bits(val[q])];
        }
      }
      uint64_t best[2] = {0, 0}; int bcnt[2] = {-1, -1};
      for (int sgn = 0; sgn < 2; ++sgn) for (auto& kv : vals[sgn]) if (kv.second > bcnt[sgn]) { bcnt[sgn] = kv.second; best[sgn] = kv.first; }
      if (bcnt[0] < 0 && bcnt[1] < 0) return false;
      if (bcnt[0] >= 0 && bcnt[1] >= 0 && best[0] != best[1]) return false;   // (a slab at the end of the grid may see one direction only)
      memcpy(&dst[k], &best[bcnt[0] >= 0 ? 0 : 1], 8);
    }
  }
  return true;
}

// Split A = S + D + R: S the star (the same coefficients in every row, truncated at the faces of the grid), D the diagonal, R the
// remainder.  EVERY row takes S + D from the sweep; a row is "clean" when its R is empty (its off-diagonal entries are exactly the
// star, bit for bit).  In the other rows — inside atom blocks, next to anything irregular — R holds: every entry that is not at a
// star position as it is; at a star position the entry MINUS the star's coefficient (nothing when they are equal; minus the
// coefficient where the row has no entry there), rounded once.  R is added to what the sweep wrote by the kernels that follow it
// (dense blocks + listed rows, spmm_dense.hip).  So the sweep needs no per-row decision and the listed rows carry only what really
// differs from the star: on the SiO2-like matrix two thirds of the non-zeros the listed rows held before were plain star entries.
static bool star_build_host(const StarRows& M, StarHost* H) {
  const int nrows = M.nrows; const int* rowptr = M.rowptr; const int* colidx = M.colidx; const double* val = M.val;
  if (M.ncols_local != nrows && M.ghost == nullptr && !M.box_cols) return false;     // halo columns of unknown origin: the other forms
  if (!star_detect(M, H)) return false;
  const StarGeom& gm = H->g;
  const int nx = gm.nx, ny = gm.ny, nz = gm.nz;
  const long sy = nx, sz = (long)nx * ny;
  H->diag.assign((size_t)nrows, 0.0);
  H->rem_rowptr.assign((size_t)nrows + 1, 0);
  uint64_t cb[3][STAR_R + 1]; double cv[3][STAR_R + 1];
  for (int k = 0; k <= STAR_R; ++k) {
    cv[0][k] = H->c.cx[k]; cv[1][k] = H->c.cy[k]; cv[2][k] = H->c.cz[k];
    for (int a = 0; a < 3; ++a) cb[a][k] = star_bits(cv[a][k]);
  }
  const long stride[3] = {1, sy, sz};
  // masked domains: grid point -> row (-1: not a row)
  std::vector<int>& inv = H->inv;
  inv.clear();
  if (M.box != nullptr) {
    inv.assign((size_t)M.nglobal, -1);
    const int ncl = M.box_cols ? M.ncols_local : nrows;               // (a slab: the halo rows too — ascending among themselves)
    for (int r = 0; r < ncl; ++r) {
      if (M.box[r] < 0 || M.box[r] >= M.nglobal || inv[M.box[r]] != -1 || (r > 0 && r != nrows && M.box[r] <= M.box[r - 1])) return false;   // (scan order, no point twice)
      inv[M.box[r]] = r;
    }
  }
  // local column (= row of X) of grid point gq: own rows, or the halo planes below / above as the sweep addresses them
  auto xcol = [&](long gq) -> long {
    if (M.box != nullptr) return inv[gq];
    const int zz = (int)(gq / sz);
    return (zz < gm.zs ? gm.lo_off : zz >= gm.ze ? gm.hi_off : gm.mid_off) + gq;
  };
  std::vector<char>& clean = H->clean;
  clean.assign((size_t)nrows, 0);
  // rows are independent: contiguous chunks, one thread each, every chunk with its own remainder arrays (joined in order below)
  const int nt = gcge_upload_threads();
  std::vector<std::vector<int>> crc((size_t)nt); std::vector<std::vector<double>> crv((size_t)nt);
  std::vector<long> cclean((size_t)nt, 0); std::vector<int> cfail((size_t)nt, 0);
  std::vector<int>& rcnt = H->rem_rowptr;                             // [r + 1] = entries of row r for now, prefix-summed below
  gcge_parallel_chunks(nrows, nt, [&](int chunk, long rb, long re) {
  std::vector<std::pair<int, double>> rem;                             // remainder of the row being looked at
  std::vector<int>& rc = crc[chunk]; std::vector<double>& rv = crv[chunk];
  long nclean = 0;
  for (int r = (int)rb; r < (int)re; ++r) {
    const long gr = M.gpos(r);
    const int gz = (int)(gr / sz), gy = (int)((gr - (long)gz * sz) / sy), gx = (int)(gr - (long)gz * sz - (long)gy * sy);
    const int g[3] = {gx, gy, gz}, dim[3] = {nx, ny, nz};
    // the neighbour k steps along axis a (sgn 0: down, 1: up) is a point of the domain
    auto exists = [&](int a, int sgn, int k) -> bool {
      if (!(sgn ? g[a] + k < dim[a] : g[a] - k >= 0)) return false;
      return M.box == nullptr || inv[gr + (sgn ? 1 : -1) * k * stride[a]] >= 0;
    };
    bool seen[3][2][STAR_R + 1] = {};
    rem.clear();
    bool have_diag = false; double dv = 0.0;
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
      const long o = M.gcol(colidx[q]) - gr;
      if (o == 0 && !have_diag) { have_diag = true; dv = val[q]; continue; }
      const long ao = o < 0 ? -o : o; const int sgn = o < 0 ? 0 : 1;
      int a = -1, k = 0;
      if (ao >= 1 && ao <= STAR_R) { a = 0; k = (int)ao; }
      else if (ao % sz == 0 && ao / sz >= 1 && ao / sz <= STAR_R) { a = 2; k = (int)(ao / sz); }
      else if (ao % sy == 0 && ao / sy >= 1 && ao / sy <= STAR_R) { a = 1; k = (int)(ao / sy); }
      // a star position: the neighbour k steps along axis a exists in the grid and the star reaches that far
      const bool star_pos = a >= 0 && cb[a][k] != 0 && !seen[a][sgn][k] && exists(a, sgn, k);
      if (!star_pos) { rem.emplace_back(colidx[q], val[q]); continue; }
      seen[a][sgn][k] = true;
      if (star_bits(val[q]) != cb[a][k]) rem.emplace_back(colidx[q], val[q] - cv[a][k]);
    }
    for (int a = 0; a < 3; ++a)
      for (int k = 1; k <= STAR_R; ++k) {
        if (cb[a][k] == 0) continue;
        for (int sgn = 0; sgn < 2; ++sgn) {
          if (!exists(a, sgn, k) || seen[a][sgn][k]) continue;
          // the sweep adds coefficient x neighbour here, the row has no such entry: the remainder takes it back
          const long gq = gr + (sgn ? 1 : -1) * k * stride[a];
          const int zz = (int)(gq / sz);
          if (zz < gm.zmin || zz >= gm.zmax) { cfail[chunk] = 1; return; }   // (cannot happen: zmin / zmax cover the star's reach)
          rem.emplace_back((int)xcol(gq), -cv[a][k]);
        }
      }
    if (dv != dv) { cfail[chunk] = 1; return; }                        // a NaN on the diagonal: not for this form
    H->diag[r] = dv;
    if (rem.empty()) { clean[r] = 1; ++nclean; }
    else {
      std::sort(rem.begin(), rem.end(), [](const std::pair<int, double>& p, const std::pair<int, double>& q) { return p.first < q.first; });
      for (auto& e : rem) { rc.push_back(e.first); rv.push_back(e.second); }
    }
    rcnt[r + 1] = (int)rem.size();
  }
  cclean[chunk] = nclean;
  });
  long nclean = 0;
  for (int c = 0; c < nt; ++c) { if (cfail[c]) return false; nclean += cclean[c]; }
  for (int r = 0; r < nrows; ++r) rcnt[r + 1] += rcnt[r];
  std::vector<int>& rc = H->rem_col; std::vector<double>& rv = H->rem_val;
  rc.resize((size_t)rcnt[nrows]); rv.resize((size_t)rcnt[nrows]);
  {
    size_t off = 0;
    for (int c = 0; c < nt; ++c) {
      if (!crc[c].empty()) { memcpy(rc.data() + off, crc[c].data(), crc[c].size() * sizeof(int)); memcpy(rv.data() + off, crv[c].data(), crv[c].size() * sizeof(double)); }
      off += crc[c].size();
      std::vector<int>().swap(crc[c]); std::vector<double>().swap(crv[c]);
    }
    if (off != rc.size()) return false;
  }
  H->nclean = nclean;
  if (2 * nclean < nrows) return false;
  return true;
}

// Geometry of a MASKED grid recovered from the rows alone (one rank; a matrix that arrives as a file, e.g. Matrix Market, names
// no grid): the rows are the points of a convex domain inside a box in scan order (x fastest), every row couples to its
// neighbours along the three axes (a star of any arm length) and to anything else (atom blocks).  Then
//   * rows r and r + 1 lie on one x LINE exactly when the entry (r, r + 1) is there;
//   * consecutive lines l, l + 1 of one PLANE (y and y + 1) are coupled through the + y neighbours: every star row of l has ONE
//     entry among the rows of l + 1, and all of them agree on the shift between where the two lines begin in x — the shift most
//     rows vote for; lines without such a vote (the last line of a plane and the first of the next one share no neighbour) end a plane;
//   * consecutive planes are coupled through the + z neighbours in the same way: a vote on the (x, y) shift between their frames.
// Votes come from the rows that are no longer than the median row (the star rows), from all rows of a line only when none of
// its rows is that short.  box[r] = x + nx (y + ny z) in the bounding box of what was found.  A wrong guess costs speed, never
// the result: the split of star_build_host is exact for any one-to-one ascending map (the remainder takes every difference),
// and the form is refused when fewer than half of the rows come out clean.  false: the rows do not look like this.
static bool star_infer_box(int nrows, const int* rowptr, const int* colidx, std::vector<int>* box_, int* nx_, int* ny_, int* nz_) {
  if (nrows < 2048) return false;
  // lines
  std::vector<int> lstart;                                             // first row of every line (+ nrows at the end)
  std::vector<int> lineof((size_t)nrows);
  long linked = 0;
  lstart.push_back(0);
  for (int r = 0; r < nrows; ++r) {
    lineof[r] = (int)lstart.size() - 1;
    if (r + 1 == nrows) break;
    bool next = false;
    for (int q = rowptr[r]; q < rowptr[r + 1] && !next; ++q) next = colidx[q] == r + 1;
    if (next) ++linked; else lstart.push_back(r + 1);
  }
  const int nl = (int)lstart.size();
  lstart.push_back(nrows);
  if (2 * linked < nrows || nl < 16) return false;
  // the median row length: rows up to it vote
  int med;
  {
    std::vector<int> len((size_t)nrows);
    for (int r = 0; r < nrows; ++r) len[r] = rowptr[r + 1] - rowptr[r];
    std::nth_element(len.begin(), len.begin() + nrows / 2, len.end());
    med = len[nrows / 2];
  }
  // planes: shift of line l + 1 against line l, or a break
  std::vector<int> xoff((size_t)nl, 0), pstart;                        // x of a line's first row in its plane's frame; first line of every plane
  std::vector<int> votes;
  pstart.push_back(0);
  for (int l = 0; l + 1 < nl; ++l) {
    const int a0 = lstart[l], a1 = lstart[l + 1], b0 = a1, b1 = lstart[l + 2];
    const int span = (a1 - a0) + (b1 - b0);
    const int need = std::min(a1 - a0, b1 - b0);
    int best = 0, bestn = 0;
    // (the + y neighbour of a star row is its only entry in the next line: the vote of a true pair of lines is nearly unanimous;
    //  pass 1 — every row of the line votes — where the short rows alone do not carry it)
    for (int pass = 0; pass < 2 && 2 * bestn < need; ++pass) {
      votes.assign((size_t)span + 1, 0);
      for (int r = a0; r < a1; ++r) {
        if (pass == 0 && rowptr[r + 1] - rowptr[r] > med) continue;
        for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
          const int c = colidx[q];
          if (c >= b0 && c < b1) ++votes[(size_t)((r - a0) - (c - b0) + (b1 - b0))];
        }
      }
      bestn = 0;
      for (int k = 0; k <= span; ++k) if (votes[k] > bestn) { bestn = votes[k]; best = k - (b1 - b0); }
    }
    if (bestn > 0 && 2 * bestn >= need) xoff[l + 1] = xoff[l] + best;
    else { pstart.push_back(l + 1); xoff[l + 1] = 0; }
  }
  const int np = (int)pstart.size();
  pstart.push_back(nl);
  if (getenv("GCGE_STAR_INFER_DEBUG")) fprintf(stderr, "star_infer_box: %d rows, %d lines, %d planes, median row %d\n", nrows, nl, np, med);
  if (np < 2 * STAR_R + 2) return false;
  // frames: (x, y) shift of plane p + 1 against plane p
  std::vector<int> px((size_t)np, 0), py((size_t)np, 0);
  std::unordered_map<long, int> pv;
  for (int p = 0; p + 1 < np; ++p) {
    const int r0 = lstart[pstart[p]], r1 = lstart[pstart[p + 1]], s1 = lstart[pstart[p + 2]];
    const long need = std::min(r1 - r0, s1 - r1);
    int bestn = 0; long bestk = 0;
    for (int pass = 0; pass < 2 && 4L * bestn < need; ++pass) {
      pv.clear();
      for (int r = r0; r < r1; ++r) {
        if (pass == 0 && rowptr[r + 1] - rowptr[r] > med) continue;
        const int l = lineof[r], x = xoff[l] + (r - lstart[l]), y = l - pstart[p];
        for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
          const int c = colidx[q];
          if (c < r1 || c >= s1) continue;
          const int lc = lineof[c], xc = xoff[lc] + (c - lstart[lc]), yc = lc - pstart[p + 1];
          if (x - xc <= -(1 << 20) || x - xc >= (1 << 20) || y - yc <= -(1 << 20) || y - yc >= (1 << 20)) return false;   // (the key packs two 22-bit fields)
          ++pv[(long)(x - xc + (1 << 20)) * (1L << 22) + (y - yc + (1 << 20))];
        }
      }
      bestn = 0;
      for (auto& kv : pv) if (kv.second > bestn) { bestn = kv.second; bestk = kv.first; }
    }
    if (getenv("GCGE_STAR_INFER_DEBUG")) fprintf(stderr, "plane %d rows %d..%d lines %d votes %d need %ld\n", p, r0, r1, pstart[p + 1] - pstart[p], bestn, need);
    if (bestn == 0 || 4L * bestn < std::min(r1 - r0, s1 - r1)) return false;   // two planes that share no column of points: not such a domain
    px[p + 1] = px[p] + (int)(bestk >> 22) - (1 << 20);
    py[p + 1] = py[p] + (int)(bestk & ((1L << 22) - 1)) - (1 << 20);
  }
  // the bounding box
  long xmin = 0, xmax = 0, ymin = 0, ymax = 0; bool first = true;
  for (int p = 0; p < np; ++p)
    for (int l = pstart[p]; l < pstart[p + 1]; ++l) {
      const long x0 = (long)px[p] + xoff[l], x1 = x0 + (lstart[l + 1] - lstart[l]) - 1, y = (long)py[p] + (l - pstart[p]);
      if (first) { xmin = x0; xmax = x1; ymin = ymax = y; first = false; }
      xmin = std::min(xmin, x0); xmax = std::max(xmax, x1); ymin = std::min(ymin, y); ymax = std::max(ymax, y);
    }
  const long nx = xmax - xmin + 1, ny = ymax - ymin + 1, nz = np;
  // a box far larger than the domain: the lines did not line up (a ball fills 52 % of its box, a thin slab of one more: 8 x the rows is
  // ample — and bounds the point-wise map of the box, 4 bytes per box point on the host and on the device, by 32 bytes per row)
  if (nx * ny * nz >= (1L << 31) || nx * ny * nz > 8L * nrows) return false;
  box_->resize((size_t)nrows);
  for (int p = 0; p < np; ++p)
    for (int l = pstart[p]; l < pstart[p + 1]; ++l) {
      const long x0 = (long)px[p] + xoff[l] - xmin, y = (long)py[p] + (l - pstart[p]) - ymin;
      for (int r = lstart[l]; r < lstart[l + 1]; ++r) (*box_)[r] = (int)(x0 + (r - lstart[l]) + nx * (y + ny * p));
    }
  for (int r = 1; r < nrows; ++r) if ((*box_)[r] <= (*box_)[r - 1]) return false;   // (scan order must have come out)
  *nx_ = (int)nx; *ny_ = (int)ny; *nz_ = (int)nz;
  return true;
}

}  // namespace gcge

using namespace gcge;

static int g_star_mode = 0;   // 0 automatic, -1 never
static int g_star_infer = 1;  // 1: a matrix without a lexicographic grid is tried as a masked grid in scan order (star_infer_box)
extern "C" void gcge_hip_spmm_star_infer(int on) { g_star_infer = on != 0; }
static int g_star_form = 3;   // 2: second form of the sweep (registers stage the halo strips), 3: third form (LDS-DMA strips, 16-column passes)
extern "C" void gcge_hip_spmm_star_form(int form) { g_star_form = form == 3 ? 3 : 2; }
static int g_star_masked_zchunks = 0;  // masked third form: z ranges per patch (0: the rule of the other forms)
extern "C" void gcge_hip_spmm_star_masked_zchunks(int n) { g_star_masked_zchunks = n; }
static int g_star_masked_third = 1;   // masked grids with a line table take the third form (0: the second form with its point-wise map, round 4)
extern "C" void gcge_hip_spmm_star_masked_third(int on) { g_star_masked_third = on != 0; }
static int g_star_lpp = 4;    // second form: 8 = 16-column passes on 16 x 8 patches (128-byte pieces of the rows), 4 = 8 columns on 16 x 16 (64-byte pieces)
extern "C" void gcge_hip_spmm_star_lanes(int lpp) { g_star_lpp = lpp == 4 ? 4 : 8; }
static int g_star_xcd = 22;   // 0: workgroups in launch order; 1 / G >= 2: XCD-aware orders (star_block_of): runs of 22 patches (two patch rows of the 171^2 plane) per XCD are 2 % faster than launch order (2.78 against 2.84 ms, profiles/r04_star/16, gpurun_out/r5/09)
extern "C" void gcge_hip_spmm_star_xcd(int on) { g_star_xcd = on < 0 ? 0 : on; }
static int g_star_dbg = 0;    // measurement only: 1 no halo loads, 2 no LDS arm reads, 4 no stores, 8 no own-plane loads (results are wrong then)
extern "C" void gcge_hip_spmm_star_dbg(int bits) { g_star_dbg = bits; }
extern "C" void gcge_hip_spmm_star_mode(int mode) { g_star_mode = mode; }
extern "C" int gcge_hip_spmm_star_mode_get(void) { return g_star_mode; }

// Grid of a matrix whose rows are (mostly) star stencils, from a slab of its rows with GLOBAL column indices (host only; what a
// partitioner needs to put its cuts on plane boundaries: gcge_amd.dist.partition_by_nnz(align = nx * ny)).  out[0..3] = nx, ny,
// nz, arm length.  1: found, 0: no such grid.
extern "C" int gcge_hip_star_grid(int nrows, long row_begin, long nglobal, const int* rowptr, const int* colidx_global, const double* val, long* out) {
  StarRows M = {nrows, nrows, row_begin, nglobal, nullptr, rowptr, colidx_global, val};
  M.global_cols = true;
  int Rx = 0, Ry = 0, Rz = 0; long sy = 0, sz = 0;
  if (!star_offsets(M, &Rx, &Ry, &Rz, &sy, &sz)) return 0;
  if (out) { out[0] = sy; out[1] = sz / sy; out[2] = nglobal / sz; out[3] = std::max(Rx, std::max(Ry, Rz)); }
  return 1;
}

// Structural self-check of the split (host only; tests): every clean row is rebuilt from the star, its diagonal and the grid and
// compared with the CSR row, bit for bit; every other row must sit in the remainder unchanged.  0: identical; > 0: differences;
// -1: the matrix does not take this form.  out[0..4] = nx, ny, nz, arm length, clean rows; out[5..10] (slabs) = first / last + 1
// plane of the slab, first / last + 1 plane the sweep may load, rows of X where the planes below / above begin (-1: none).
static long star_selfcheck(const StarRows& M, long* out) {
  StarHost H;
  const int nrows = M.nrows; const long row_begin = M.row_begin; const int* rowptr = M.rowptr; const int* colidx = M.colidx; const double* val = M.val;
  if (!star_build_host(M, &H)) return -1;
  const StarGeom& g = H.g;
  const long sy = g.nx, sz = (long)g.nx * g.ny;
  if (out) {
    out[0] = g.nx; out[1] = g.ny; out[2] = g.nz; out[3] = H.R; out[4] = H.nclean;
    out[5] = g.zs; out[6] = g.ze; out[7] = g.zmin; out[8] = g.zmax;
    out[9] = g.zmin < g.zs ? g.lo_off + sz * g.zmin : -1; out[10] = g.ze < g.zmax ? g.hi_off + sz * g.ze : -1;
  }
  if (out) out[11] = H.rem_rowptr[nrows];
  long bad = 0;
  // where the sweep finds grid point (global row gq): the row of X, as the kernel computes it
  auto xrow = [&](long gq) -> long {
    if (M.box != nullptr) return H.inv[gq];
    const int zz = (int)(gq / sz); return (zz < g.zs ? g.lo_off : zz >= g.ze ? g.hi_off : g.mid_off) + gq;
  };
  auto there = [&](long gq) -> bool { return M.box == nullptr || H.inv[gq] >= 0; };   // a point of the domain
  // every row rebuilt as star + diagonal + remainder (per column, in the order the kernels add them) against the CSR row: clean
  // rows bit for bit; the others to the one rounding of "entry minus coefficient" at the star positions; a position the row
  // has no entry at must cancel exactly
  std::vector<std::pair<long, double>> want, got;
  for (int r = 0; r < nrows; ++r) {
    want.clear(); got.clear();
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) want.emplace_back((long)colidx[q], val[q]);   // local columns = rows of X
    const long gr = M.gpos(r);
    const int gz = (int)(gr / sz), gy = (int)((gr - (long)gz * sz) / sy), gx = (int)(gr - (long)gz * sz - (long)gy * sy);
    const bool is_clean = H.clean[r] != 0;
    if (is_clean != (H.rem_rowptr[r + 1] == H.rem_rowptr[r])) ++bad;
    bool stored_diag = false;
    for (auto& w : want) stored_diag |= w.first == r;
    if (stored_diag || H.diag[r] != 0.0) got.emplace_back((long)r, H.diag[r]);
    for (int k = 1; k <= STAR_R; ++k) {
      if (star_bits(H.c.cx[k])) { if (gx - k >= 0 && there(gr - k)) got.emplace_back(xrow(gr - k), H.c.cx[k]); if (gx + k < g.nx && there(gr + k)) got.emplace_back(xrow(gr + k), H.c.cx[k]); }
      if (star_bits(H.c.cy[k])) { if (gy - k >= 0 && there(gr - k * sy)) got.emplace_back(xrow(gr - k * sy), H.c.cy[k]); if (gy + k < g.ny && there(gr + k * sy)) got.emplace_back(xrow(gr + k * sy), H.c.cy[k]); }
      if (star_bits(H.c.cz[k])) {
        if (gz - k >= 0 && there(gr - k * sz)) { if (gz - k < g.zmin) ++bad; got.emplace_back(xrow(gr - k * sz), H.c.cz[k]); }
        if (gz + k < g.nz && there(gr + k * sz)) { if (gz + k >= g.zmax) ++bad; got.emplace_back(xrow(gr + k * sz), H.c.cz[k]); }
      }
    }
    const size_t nstar = got.size();
    for (int q = H.rem_rowptr[r]; q < H.rem_rowptr[r + 1]; ++q) got.emplace_back((long)H.rem_col[q], H.rem_val[q]);
    auto less = [](const std::pair<long, double>& p, const std::pair<long, double>& q) { return p.first < q.first; };
    std::stable_sort(want.begin(), want.end(), less); std::stable_sort(got.begin(), got.end(), less);   // (stable: star part before remainder)
    (void)nstar;
    size_t i = 0, j = 0;
    bool ok = true;
    while (i < want.size() || j < got.size()) {
      const long c = (j >= got.size() || (i < want.size() && want[i].first <= got[j].first)) ? want[i].first : got[j].first;
      double wv = 0.0, gv = 0.0, mag = 0.0; int nw = 0, ng = 0;
      while (i < want.size() && want[i].first == c) { wv += want[i].second; ++i; ++nw; }
      while (j < got.size() && got[j].first == c) { gv += got[j].second; mag = std::max(mag, fabs(got[j].second)); ++j; ++ng; }
      if (is_clean || nw == 0 || ng <= 1) { if (star_bits(wv + 0.0) != star_bits(gv + 0.0) && !(wv == 0.0 && gv == 0.0)) ok = false; }
      else if (fabs(wv - gv) > 4.0 * 2.220446049250313e-16 * std::max(mag, fabs(wv))) ok = false;
    }
    if (!ok) ++bad;
  }
  return bad;
}
extern "C" long gcge_hip_star_selfcheck_slab(int nrows, int ncols_local, long row_begin, long nglobal, const int* ghost, const int* rowptr,
                                             const int* colidx, const double* val, long* out) {
  const StarRows M = {nrows, ncols_local, row_begin, nglobal, ghost, rowptr, colidx, val};
  return star_selfcheck(M, out);
}
// the same for a matrix on a MASKED grid: row r is grid point box_of_row[r] = x + nx (y + ny z) (ascending: scan order); out as above
extern "C" long gcge_hip_star_selfcheck_grid(int nrows, const int* rowptr, const int* colidx, const double* val, int nx, int ny, int nz,
                                             const int* box_of_row, long* out) {
  StarRows M = {nrows, nrows, 0, (long)nx * ny * nz, nullptr, rowptr, colidx, val};
  M.box = box_of_row; M.bnx = nx; M.bny = ny; M.bnz = nz;
  return star_selfcheck(M, out);
}
// The geometry star_infer_box recovers (host only; tests, tools): dims[0..2] = nx, ny, nz of the bounding box, box_of_row[r] = x + nx
// (y + ny z).  1: found, 0: the rows are not the points of a masked grid in scan order.
extern "C" int gcge_hip_star_infer_grid(int nrows, const int* rowptr, const int* colidx, int* dims, int* box_of_row) {
  std::vector<int> box; int nx = 0, ny = 0, nz = 0;
  if (!star_infer_box(nrows, rowptr, colidx, &box, &nx, &ny, &nz)) return 0;
  dims[0] = nx; dims[1] = ny; dims[2] = nz;
  memcpy(box_of_row, box.data(), (size_t)nrows * sizeof(int));
  return 1;
}
extern "C" long gcge_hip_star_selfcheck(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, long* out) {
  if (ncols_local != nrows) return -1;
  long o[12];
  const long bad = gcge_hip_star_selfcheck_slab(nrows, ncols_local, 0, nrows, nullptr, rowptr, colidx, val, o);
  if (out && bad >= 0) for (int i = 0; i < 5; ++i) out[i] = o[i];
  return bad;
}

extern "C" void gcge_hip_star_free(void* sm) {
  StarMat* S = (StarMat*)sm;
  if (!S) return;
  hipFree(S->d_diag); hipFree(S->d_clean);
  if (S->d_map) { hipFree(S->d_map); hipFree(S->d_prange); }
  if (S->d_lines) { hipFree(S->d_lines); hipFree(S->d_prange3); hipFree(S->d_order3); }
  delete S;
}

// NULL: the matrix keeps the other forms.  Otherwise the device object, and through rem_* the CSR arrays of the remainder (rows
// that are not clean keep all their entries, clean rows are empty), owned by the object until gcge_hip_star_release_remainder.
// ghost: the global rows behind the halo columns nrows .. ncols_local - 1 (ascending; NULL on one rank).
static StarHost* g_star_last = nullptr;
// geometry of the NEXT matrix handed to gcge_hip_star_build (gcge_hip_mat_create_grid sets it, the build consumes it)
static struct { int nrows, nx, ny, nz; const int* box; bool cols; } g_star_geom = {0, 0, 0, 0, nullptr, false};
extern "C" void gcge_hip_star_next_geometry(int nrows, int nx, int ny, int nz, const int* box_of_row) {
  g_star_geom.nrows = nrows; g_star_geom.nx = nx; g_star_geom.ny = ny; g_star_geom.nz = nz; g_star_geom.box = box_of_row; g_star_geom.cols = false;
}
// a ROW SLAB of a masked grid: the box index of every local column — the own rows, then the halo rows (ascending by global row)
extern "C" void gcge_hip_star_next_geometry_cols(int ncols_local, int nx, int ny, int nz, const int* box_of_local_col) {
  g_star_geom.nrows = ncols_local; g_star_geom.nx = nx; g_star_geom.ny = ny; g_star_geom.nz = nz; g_star_geom.box = box_of_local_col; g_star_geom.cols = true;
}
extern "C" void* gcge_hip_star_build(int nrows, int ncols_local, long row_begin, long nglobal, const int* ghost, const int* rowptr,
                                     const int* colidx, const double* val, const int** rem_rowptr, const int** rem_col, const double** rem_val) {
  if (g_star_mode < 0 || nrows <= 0) return nullptr;
  StarHost* H = new StarHost();
  StarRows M = {nrows, ncols_local, row_begin, nglobal, ghost, rowptr, colidx, val};
  if (g_star_geom.box != nullptr) {
    if (!g_star_geom.cols && g_star_geom.nrows == nrows && ncols_local == nrows && row_begin == 0) {
      M.box = g_star_geom.box; M.bnx = g_star_geom.nx; M.bny = g_star_geom.ny; M.bnz = g_star_geom.nz;
      M.nglobal = (long)M.bnx * M.bny * M.bnz;
    } else if (g_star_geom.cols && g_star_geom.nrows == ncols_local) {
      M.box = g_star_geom.box; M.bnx = g_star_geom.nx; M.bny = g_star_geom.ny; M.bnz = g_star_geom.nz; M.box_cols = true;
      M.nglobal = (long)M.bnx * M.bny * M.bnz;
    }
    g_star_geom.box = nullptr;
  }
  bool ok = star_build_host(M, H);
  if (!ok && M.box == nullptr && g_star_infer && ncols_local == nrows && row_begin == 0 && nglobal == nrows) {
    // no lexicographic grid: the points of a masked one in scan order?  (the geometry recovered from the rows)
    delete H; H = new StarHost();
    int bx = 0, by = 0, bz = 0;
    if (star_infer_box(nrows, rowptr, colidx, &H->own_box, &bx, &by, &bz)) {
      M.box = H->own_box.data(); M.bnx = bx; M.bny = by; M.bnz = bz; M.nglobal = (long)bx * by * bz;
      ok = star_build_host(M, H);
    }
  }
  if (!ok) { delete H; return nullptr; }
  StarMat* S = new StarMat();
  S->d_map = nullptr; S->d_prange = nullptr; S->d_lines = nullptr; S->d_prange3 = nullptr; S->d_order3 = nullptr; S->npatch3 = 0;
  if (M.box != nullptr) {
    GCGE_HIP_CHECK(hipMalloc(&S->d_map, H->inv.size() * sizeof(int)));
    GCGE_HIP_CHECK(hipMemcpy(S->d_map, H->inv.data(), H->inv.size() * sizeof(int), hipMemcpyHostToDevice));
    const int nx = H->g.nx, ny = H->g.ny, nz = H->g.nz, ntx = (nx + STAR_T - 1) / STAR_T, nty = (ny + STAR_T - 1) / STAR_T;
    std::vector<int> pr((size_t)2 * ntx * nty);
    for (int p = 0; p < ntx * nty; ++p) { pr[2 * p] = nz; pr[2 * p + 1] = 0; }
    for (int r = 0; r < nrows; ++r) {
      const long b = M.box[r];
      const int z = (int)(b / ((long)nx * ny)), y = (int)((b / nx) % ny), x = (int)(b % nx), p = (y / STAR_T) * ntx + x / STAR_T;
      pr[2 * p] = std::min(pr[2 * p], z); pr[2 * p + 1] = std::max(pr[2 * p + 1], z + 1);
    }
    GCGE_HIP_CHECK(hipMalloc(&S->d_prange, pr.size() * sizeof(int)));
    GCGE_HIP_CHECK(hipMemcpy(S->d_prange, pr.data(), pr.size() * sizeof(int), hipMemcpyHostToDevice));
    // the line table of the third form: every grid line one run of rows (scan order: row = base + x), box edges below 65536
    const int nyp = ny + STAR3M_YLO + STAR3M_YHI, nzp = nz + STAR3M_ZLO + STAR3M_ZHI;
    std::vector<StarLine> ln((size_t)nyp * nzp, StarLine{0, 0});
    bool runs = nx < 32768 && nrows > 0;
    const int nty8 = (ny + STAR3_TY - 1) / STAR3_TY;
    std::vector<int> pr3((size_t)2 * ntx * nty8);
    for (int p = 0; p < ntx * nty8; ++p) { pr3[2 * p] = nz; pr3[2 * p + 1] = 0; }
    const int ncl = M.box_cols ? M.ncols_local : nrows;               // (a slab: the halo rows have lines as well; a line is own OR halo)
    for (int r = 0; r < ncl && runs; ++r) {
      const long b = M.box[r];
      const int z = (int)(b / ((long)nx * ny)), y = (int)((b / nx) % ny), x = (int)(b % nx);
      StarLine& L = ln[(size_t)(z + STAR3M_ZLO) * nyp + (y + STAR3M_YLO)];
      if (L.xr == 0) { L.base = r - x; L.xr = x | (x + 1) << 16; }
      else if (r - x == L.base && x == (L.xr >> 16) && r != nrows) L.xr = (L.xr & 0xffff) | (x + 1) << 16;
      else runs = false;                                                 // a hole in the line (or a line cut by the partition): the point-wise map of the second form
      if (r >= nrows) continue;
      const int p = (y / STAR3_TY) * ntx + x / STAR_T;
      pr3[2 * p] = std::min(pr3[2 * p], z); pr3[2 * p + 1] = std::max(pr3[2 * p + 1], z + 1);
    }
    if (runs) {
      GCGE_HIP_CHECK(hipMalloc(&S->d_lines, ln.size() * sizeof(StarLine)));
      GCGE_HIP_CHECK(hipMemcpy(S->d_lines, ln.data(), ln.size() * sizeof(StarLine), hipMemcpyHostToDevice));
      GCGE_HIP_CHECK(hipMalloc(&S->d_prange3, pr3.size() * sizeof(int)));
      GCGE_HIP_CHECK(hipMemcpy(S->d_prange3, pr3.data(), pr3.size() * sizeof(int), hipMemcpyHostToDevice));
      std::vector<int> ord;
      for (int p = 0; p < ntx * nty8; ++p) if (pr3[2 * p + 1] > pr3[2 * p]) ord.push_back(p);
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return pr3[2 * a + 1] - pr3[2 * a] > pr3[2 * b + 1] - pr3[2 * b]; });
      S->npatch3 = (int)ord.size();
      GCGE_HIP_CHECK(hipMalloc(&S->d_order3, std::max<size_t>(ord.size(), 1) * sizeof(int)));
      GCGE_HIP_CHECK(hipMemcpy(S->d_order3, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice));
    }
  }
  S->masked_slab = M.box != nullptr && M.box_cols && ncols_local != nrows;
  S->shared_lo = S->shared_hi = 0;
  if (S->masked_slab) {
    const long pl = (long)H->g.nx * H->g.ny;
    for (int c = nrows; c < ncols_local; ++c) {
      const int z = (int)(M.box[c] / pl);
      if (z == H->g.zs) S->shared_lo = 1;
      if (z == H->g.ze - 1) S->shared_hi = 1;
    }
  }
  S->g = H->g; S->R = H->R; S->nclean = H->nclean; S->nrows = nrows; S->c = H->c;
  S->iso = memcmp(H->c.cx, H->c.cy, sizeof(H->c.cx)) == 0 && memcmp(H->c.cx, H->c.cz, sizeof(H->c.cx)) == 0;
  if (S->masked_slab && (S->d_lines == nullptr || !S->iso)) {         // a slab of a masked grid: the line table or nothing (the row map of the second form knows no halo rows)
    hipFree(S->d_map); hipFree(S->d_prange); if (S->d_lines) { hipFree(S->d_lines); hipFree(S->d_prange3); hipFree(S->d_order3); } delete S; delete H; return nullptr;
  }
  if (S->d_map != nullptr && !S->iso) { hipFree(S->d_map); hipFree(S->d_prange); if (S->d_lines) { hipFree(S->d_lines); hipFree(S->d_prange3); hipFree(S->d_order3); } delete S; delete H; return nullptr; }   // (masked grids: the one-coefficient-set kernel only)
  GCGE_HIP_CHECK(hipMalloc(&S->d_diag, (size_t)nrows * sizeof(double)));
  GCGE_HIP_CHECK(hipMemcpy(S->d_diag, H->diag.data(), (size_t)nrows * sizeof(double), hipMemcpyHostToDevice));
  std::vector<double>().swap(H->diag);
  GCGE_HIP_CHECK(hipMalloc(&S->d_clean, (size_t)nrows));
  GCGE_HIP_CHECK(hipMemcpy(S->d_clean, H->clean.data(), (size_t)nrows, hipMemcpyHostToDevice));
  *rem_rowptr = H->rem_rowptr.data(); *rem_col = H->rem_col.data(); *rem_val = H->rem_val.data();
  if (g_star_last) delete g_star_last;
  g_star_last = H;
  return S;
}
extern "C" void gcge_hip_star_release_remainder(void) { if (g_star_last) { delete g_star_last; g_star_last = nullptr; } }

extern "C" const unsigned char* gcge_hip_star_host_mask(void) {   // 1: a star row (valid until gcge_hip_star_release_remainder)
  return g_star_last ? (const unsigned char*)g_star_last->clean.data() : nullptr;
}
// 0: every grid point is a row; 2 / 3: a masked grid swept by the second (point-wise row map) / third form (line table, LDS-DMA strips)
extern "C" int gcge_hip_star_masked_form(const void* sm) {
  const StarMat* S = (const StarMat*)sm;
  if (S->d_map == nullptr) return 0;
  return (S->iso && S->d_lines != nullptr && (S->masked_slab || (g_star_form == 3 && g_star_masked_third))) ? 3 : 2;
}
extern "C" void gcge_hip_star_stats(const void* sm, long* out) {   // nx, ny, nz, arm length, clean rows, rows, first / last + 1 plane of the slab
  const StarMat* S = (const StarMat*)sm;
  out[0] = S->g.nx; out[1] = S->g.ny; out[2] = S->g.nz; out[3] = S->R; out[4] = S->nclean; out[5] = S->nrows; out[6] = S->g.zs; out[7] = S->g.ze;
}

// The output planes that need no halo row — [*ilo, *ihi) — so that they can be swept while the halo is in flight (the sweep loads
// STAR_R planes on either side of an output plane whatever the arm length).  Returns 0 when the slab has no halo or no such plane.
extern "C" int gcge_hip_star_interior(const void* sm, int* ilo, int* ihi) {
  const StarGeom& g = ((const StarMat*)sm)->g;
  if (((const StarMat*)sm)->masked_slab) {
    // a plane shared with the neighbour holds halo lines itself: every output plane within R of it waits for the exchange
    const StarMat* S = (const StarMat*)sm;
    const int lo = g.zmin < g.zs || S->shared_lo ? g.zs + STAR_R + S->shared_lo : g.zs;
    const int hi = g.ze < g.zmax || S->shared_hi ? g.ze - STAR_R - S->shared_hi : g.ze;
    if (ilo) *ilo = lo;
    if (ihi) *ihi = hi;
    return hi > lo;
  }
  const int lo = g.zmin < g.zs ? g.zs + STAR_R : g.zs, hi = g.ze < g.zmax ? g.ze - STAR_R : g.ze;
  if (ilo) *ilo = lo;
  if (ihi) *ihi = hi;
  return (g.zmin < g.zs || g.ze < g.zmax) && hi > lo;
}

extern "C" double* gcge_hip_partial_ws(size_t len);
extern "C" void gcge_hip_reduce_partials(const double* d_partial, int nblocks, int len, double* d_out, void* stream);

// one launch over the output planes [zlo, zhi); returns the number of partial rows it writes behind `part` (DOT) or 0
static int star_launch(const StarMat* S, const double* d_x, long ldx, double* d_y, long ldy, int ncols, int zlo, int zhi, double* part, bool count_only,
                       hipStream_t stream) {
  const int nzl = zhi - zlo;
  if (nzl <= 0) return 0;
  const StarGeom& g = S->g;
  const bool slab = g.zmin < g.zs || g.ze < g.zmax, iso = S->iso;
  // 8 lanes per point = 16-column passes on 16 x 8 patches, 4 = 8-column passes on 16 x 16 patches (see the kernel)
  const bool mapped = S->d_map != nullptr;
  // third form (LDS-DMA strips, two planes of prefetch): one coefficient set, every grid point a row, lane offsets of a plane in 32 bits
  const bool third = g_star_form == 3 && iso && !mapped && (double)g.nx * g.ny * (double)ldx * 8.0 < 4.0e9;
  // ... and on a masked grid with a line table (spmm_star3m_kernel: 64-bit lane addresses, no size limit)
  const bool third_m = iso && mapped && S->d_lines != nullptr && ldx < (1L << 31) && (S->masked_slab || (g_star_form == 3 && g_star_masked_third));
  const int lpp = (third || third_m) ? 8 : (!iso || mapped || g_star_lpp == 4) ? 4 : 8;   // (per-axis coefficients / masked grids: the 8-column form, which needs fewer registers)
  const int ty = 64 / lpp;
  const int ntx = (g.nx + STAR_T - 1) / STAR_T, nty = (g.ny + ty - 1) / ty, npass = (ncols + 2 * lpp - 1) / (2 * lpp);
  // z ranges: ONE where the patches x passes already give every CU two workgroups' worth of work (each range re-reads 12 planes
  // of warm-up: 171^3, 64 columns: 1 / 2 / 3 / 4 ranges = 3.93 / 4.26 / 4.35 / 4.47 ms for the whole product), otherwise enough
  // ranges of at least 24 planes to get there
  int zchunks = (int)std::max(1L, std::min((long)nzl / 24, (2L * 256 + (long)ntx * nty * npass - 1) / ((long)ntx * nty * npass)));
  if (third_m && g_star_masked_zchunks > 0) zchunks = std::max(1, std::min(g_star_masked_zchunks, nzl / 24));
  const int zlen = (nzl + zchunks - 1) / zchunks;
  zchunks = (nzl + zlen - 1) / zlen;
  const int npatch = third_m ? S->npatch3 : ntx * nty;                 // (masked third form: only the patches that have rows)
  const int nb = npatch * zchunks;
  if (count_only || nb == 0) return nb;
  const dim3 grid((unsigned)npatch, (unsigned)zchunks, (unsigned)npass);
  // operands in global plane numbering (see the kernel): the slab's first plane is plane zs of the grid
  const long shift = g.mid_off;                                       // = - plane_rows * zs
  const double* xv = (const double*)((uintptr_t)d_x + (uintptr_t)(shift * ldx * (long)sizeof(double)));
  double* yv = (double*)((uintptr_t)d_y + (uintptr_t)(shift * ldy * (long)sizeof(double)));
  const double* dv = (const double*)((uintptr_t)S->d_diag + (uintptr_t)(shift * (long)sizeof(double)));
  const unsigned char* cv = (const unsigned char*)((uintptr_t)S->d_clean + (uintptr_t)shift);
  const long dlo = g.lo_off - g.mid_off, dhi = g.hi_off - g.mid_off;
#define STAR_ARGS g.nx, g.ny, g.zs, g.ze, g.zmin, g.zmax, dlo, dhi, S->c, dv, xv, (size_t)ldx, yv, (size_t)ldy, ncols, zlo, zhi, zlen, ntx, part, cv, (const int*)S->d_map, (const int*)S->d_prange
#define STAR_LAUNCH2(DOT, ISO, SLAB, LPP) hipLaunchKernelGGL((spmm_star2_kernel<DOT, ISO, SLAB, LPP>), grid, dim3(1024), 0, stream, STAR_ARGS)
#define STAR_DBG(B) case B: hipLaunchKernelGGL((spmm_star2_kernel<false, true, false, 8, B>), grid, dim3(1024), 0, stream, STAR_ARGS); return nb;
  if (g_star_dbg != 0 && part == nullptr && iso && lpp == 8 && !slab)
    switch (g_star_dbg) { STAR_DBG(1) STAR_DBG(2) STAR_DBG(8) STAR_DBG(3) STAR_DBG(9) STAR_DBG(16) default: break; }
  const bool dot = part != nullptr;
  if (third) {
    static bool attr_dev[64] = {};                                    // per device: a process may drive several
    int dev_ = 0; (void)hipGetDevice(&dev_);
    bool& attr_set = attr_dev[dev_ & 63];
    if (!attr_set) {
      bool ok = true;
      ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_star3_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STAR3_LDS) == hipSuccess;
      ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_star3_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STAR3_LDS) == hipSuccess;
      ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_star3_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STAR3_LDS) == hipSuccess;
      ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_star3_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STAR3_LDS) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); fprintf(stderr, "gcge_hip: %u bytes of LDS refused for the third form of the sweep\n", STAR3_LDS); abort(); }
      attr_set = true;
    }
#define STAR_LAUNCH3(DOT, SLAB) hipLaunchKernelGGL((spmm_star3_kernel<DOT, SLAB>), grid, dim3(1024), STAR3_LDS, stream, g.nx, g.ny, g.zs, g.ze, g.zmin, g.zmax, \
                                                   dlo, dhi, S->c, dv, xv, (size_t)ldx, yv, (size_t)ldy, ncols, zlo, zhi, zlen, ntx, part, cv, g_star_xcd)
    if (g_star_dbg == 32 && !dot && !slab) {
      static bool probe_dev[64] = {};
      bool& probe_attr = probe_dev[dev_ & 63];
      if (!probe_attr) { GCGE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_star3_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STAR3_LDS)); probe_attr = true; }
      hipLaunchKernelGGL((spmm_star3_kernel<false, false, true>), grid, dim3(1024), STAR3_LDS, stream, g.nx, g.ny, g.zs, g.ze, g.zmin, g.zmax,
                         dlo, dhi, S->c, dv, xv, (size_t)ldx, yv, (size_t)ldy, ncols, zlo, zhi, zlen, ntx, part, cv, g_star_xcd);
      return nb;
    }
    if (dot) { if (slab) STAR_LAUNCH3(true, true); else STAR_LAUNCH3(true, false); }
    else     { if (slab) STAR_LAUNCH3(false, true); else STAR_LAUNCH3(false, false); }
#undef STAR_LAUNCH3
    return nb;
  }
  if (third_m) {
    static bool attr_mdev[64] = {};
    int dev_ = 0; (void)hipGetDevice(&dev_);
    bool& attr_m = attr_mdev[dev_ & 63];
    if (!attr_m) {
      bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_star3m_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STAR3M_LDS) == hipSuccess;
      ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_star3m_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)STAR3M_LDS) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); fprintf(stderr, "gcge_hip: %u bytes of LDS refused for the third form of the sweep (masked grids)\n", STAR3M_LDS); abort(); }
      attr_m = true;
    }
    if (dot) hipLaunchKernelGGL((spmm_star3m_kernel<true>), grid, dim3(1024), STAR3M_LDS, stream, g.nx, g.ny, g.nz, S->c, (const double*)S->d_diag, d_x, (size_t)ldx, d_y, (size_t)ldy,
                                ncols, zlo, zhi, zlen, ntx, part, (const unsigned char*)S->d_clean, (const unsigned long long*)S->d_lines, (const int*)S->d_prange3, (const int*)S->d_order3, (int)S->nrows);
    else     hipLaunchKernelGGL((spmm_star3m_kernel<false>), grid, dim3(1024), STAR3M_LDS, stream, g.nx, g.ny, g.nz, S->c, (const double*)S->d_diag, d_x, (size_t)ldx, d_y, (size_t)ldy,
                                ncols, zlo, zhi, zlen, ntx, part, (const unsigned char*)S->d_clean, (const unsigned long long*)S->d_lines, (const int*)S->d_prange3, (const int*)S->d_order3, (int)S->nrows);
    return nb;
  }
  if (mapped) {          // masked grid: one rank, one coefficient set (checked at upload)
    if (S->masked_slab) { fprintf(stderr, "gcge_hip: a row slab of a masked grid has the third form of the sweep only (leading dimension %ld)\n", ldx); abort(); }
    if (dot) hipLaunchKernelGGL((spmm_star2_kernel<true, true, false, 4, 0, true>), grid, dim3(1024), 0, stream, STAR_ARGS);
    else     hipLaunchKernelGGL((spmm_star2_kernel<false, true, false, 4, 0, true>), grid, dim3(1024), 0, stream, STAR_ARGS);
    return nb;
  }
#define STAR_PICK(ISO, LPP)                                                                                          \
  do { if (dot) { if (slab) STAR_LAUNCH2(true, ISO, true, LPP); else STAR_LAUNCH2(true, ISO, false, LPP); }          \
       else     { if (slab) STAR_LAUNCH2(false, ISO, true, LPP); else STAR_LAUNCH2(false, ISO, false, LPP); } } while (0)
  if (iso) { if (lpp == 8) STAR_PICK(true, 8); else STAR_PICK(true, 4); }
  else STAR_PICK(false, 4);
#undef STAR_PICK
#undef STAR_DBG
#undef STAR_LAUNCH2
#undef STAR_ARGS
  return nb;
}

// Y[clean rows, 0:ncols) = (star + diagonal) X; the other rows of Y are left as they are.  -1: operands this kernel does not take.
// d_dots != NULL: d_dots[0:ncols) = sum over the star rows of x y, d_dots[ncols:2 ncols) = of y^2 (device, 2 ncols doubles).
// part: 0 = every plane of the slab in one go; 1 = the planes that need no halo row (gcge_hip_star_interior); 2 = the others, after
// part 1 on the same stream — with d_dots the sums of part 1 stay in the partial workspace and part 2 reduces both (fixed order).
static int g_star_part1_rows = 0;
extern "C" int gcge_hip_star_spmm_part(const void* sm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, double* d_dots, void* stream, int part) {
  const StarMat* S = (const StarMat*)sm;
  if (ncols <= 0) return 0;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15) || d_x == d_y) return -1;
  const StarGeom& g = S->g;
  int ilo = g.zs, ihi = g.ze;
  if (part != 0 && !gcge_hip_star_interior(sm, &ilo, &ihi)) return -1;
  hipStream_t st = (hipStream_t)stream;
  int zr[3][2]; int nr = 0;
  if (part == 0) { zr[0][0] = g.zs; zr[0][1] = g.ze; nr = 1; }
  else if (part == 1) { zr[0][0] = ilo; zr[0][1] = ihi; nr = 1; }
  else { zr[0][0] = g.zs; zr[0][1] = ilo; zr[1][0] = ihi; zr[1][1] = g.ze; nr = 2; }
  if (d_dots == nullptr) {
    for (int i = 0; i < nr; ++i) star_launch(S, d_x, ldx, d_y, ldy, ncols, zr[i][0], zr[i][1], nullptr, false, st);
    return (int)hipGetLastError();
  }
  // with the column sums of the star rows: one partial row per (patch, z range), summed in fixed order; the workspace is laid
  // out for the whole product (interior first, then the two strips) so that part 1 and part 2 share it
  const int n_int = part == 0 ? 0 : star_launch(S, d_x, ldx, d_y, ldy, ncols, ilo, ihi, nullptr, true, st);
  int n_all = n_int;
  if (part == 0) n_all = star_launch(S, d_x, ldx, d_y, ldy, ncols, g.zs, g.ze, nullptr, true, st);
  else n_all += star_launch(S, d_x, ldx, d_y, ldy, ncols, g.zs, ilo, nullptr, true, st) + star_launch(S, d_x, ldx, d_y, ldy, ncols, ihi, g.ze, nullptr, true, st);
  double* ws = gcge_hip_partial_ws((size_t)n_all * 2 * ncols);
  if (part == 1) { star_launch(S, d_x, ldx, d_y, ldy, ncols, ilo, ihi, ws, false, st); g_star_part1_rows = n_int; return (int)hipGetLastError(); }
  int off = part == 2 ? n_int : 0;
  if (part == 2 && g_star_part1_rows != n_int) return -2;            // (part 1 of the same product must have run just before)
  for (int i = 0; i < nr; ++i) off += star_launch(S, d_x, ldx, d_y, ldy, ncols, zr[i][0], zr[i][1], ws + (size_t)off * 2 * ncols, false, st);
  g_star_part1_rows = 0;
  gcge_hip_reduce_partials(ws, n_all, 2 * ncols, d_dots, stream);
  return (int)hipGetLastError();
}
extern "C" int gcge_hip_star_spmm_dots(const void* sm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, double* d_dots, void* stream) {
  return gcge_hip_star_spmm_part(sm, d_x, ldx, d_y, ldy, ncols, d_dots, stream, 0);
}
extern "C" int gcge_hip_star_spmm(const void* sm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream) {
  return gcge_hip_star_spmm_part(sm, d_x, ldx, d_y, ldy, ncols, nullptr, stream, 0);
}
// d_out[0:m) = sum over the LISTED rows of x[r, j] y[r, j], d_out[m:2m) = of y[r, j]^2 (the rows the sweep does not multiply)
extern "C" int gcge_hip_star_coldots2_rows(int nlist, const int* d_list, const double* d_x, long ldx, const double* d_y, long ldy, int m,
                                           double* d_out, void* stream) {
  if (m <= 0) return 0;
  if (nlist <= 0) return (int)hipMemsetAsync(d_out, 0, 2 * (size_t)m * sizeof(double), (hipStream_t)stream);
  int nb = (nlist + 511) / 512; if (nb > 2048) nb = 2048;
  const int per = ((nlist + nb - 1) / nb + 3) / 4 * 4;
  nb = (nlist + per - 1) / per;
  double* part = gcge_hip_partial_ws((size_t)nb * 2 * m);
  hipLaunchKernelGGL(star_coldots2_rows, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, nlist, d_list, d_x, (size_t)ldx, d_y, (size_t)ldy, m, part, per);
  gcge_hip_reduce_partials(part, nb, 2 * m, d_out, stream);
  return (int)hipGetLastError();
}
