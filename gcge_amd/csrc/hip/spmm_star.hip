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
// What is not clean — rows inside atom blocks, rows with any other entry — stays a CSR matrix (the remainder, every entry of
// those rows) and takes the block / tile / pad-8 forms; the remainder is multiplied first (its pad-8 part is a LIST of the other rows,
// gcge_hip_dense_build_rows: the clean rows are neither read nor written there), this kernel then writes the clean ones.  One result per row either way: bit-reproducible.
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
  StarGeom g; int R; bool iso; long nclean, nrows; StarCoef c; double* d_diag;   // d_diag[local row]: the row's diagonal entry, NaN: not a clean row
};

// staging plan of a thread: unit u = tid + 1024 q, point u >> 2, 16-byte part u & 3
struct StarUnit { int src; int dst; };   // src: in-plane row offset (x + nx y) or -1 (outside the grid: zero); dst: LDS index (v2d units), -1: no unit

__device__ __forceinline__ v2d star_ld(const double* __restrict__ x, size_t ldx, long row, int col) {
  return *reinterpret_cast<const v2d*>(x + (size_t)row * ldx + col);
}

// One plane step with queue phase U (compile-time): see the file header.  `z` is the output plane.
#define STAR_SLOT(U, k) (((U) + 6 + (k) + 2 * STAR_Q) % STAR_Q)

// DOT: additionally partial[(workgroup of this patch and z range) * 2 ncols + j] = sum over the star rows the workgroup wrote of
// x[r, j] y[r, j], and at + ncols the same of y[r, j]^2 (columns of this pass only) — the p.w and w.w of a CG step, for free.
// ISO: the three axes share one set of coefficients, bit for bit (a uniform grid spacing — the usual case): 12 scalar registers
// of coefficients instead of 36.  SLAB: a row slab with halo planes (dlo / dhi != 0); without it the selects below are compiled out.
// The scalar registers are what this loop is short of: with everything a slab needs on top of three coefficient sets the compiler
// re-fetched kernel arguments inside the loop (s_load + lgkmcnt(0) per step: 2.42 -> 2.74 ms on the 171^3 matrix).  Hence also:
// diagv / xv / yv are the operands shifted by the host to GLOBAL plane numbering (row of grid point (i, zz) = plane_rows zz + i
// for the planes of the slab), so no "minus first plane" is left in here; dlo / dhi = what to add for the planes below / above.
template <bool DOT, bool ISO, bool SLAB>
__global__ __launch_bounds__(1024) void spmm_star_kernel(int nx, int ny, int zs, int ze, int zmin, int zmax, long dlo, long dhi, StarCoef cf,
    const double* __restrict__ diag, const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy, int ncols,
    int zlo, int zhi, int zlen, int ntx, double* __restrict__ partial) {
  __shared__ v2d plane[4 * STAR_NP];            // part-major: plane[part * NP + slot]
  __shared__ v2d corein[4 * 256];
  __shared__ v2d outt[4 * 256];
  const int tid = threadIdx.x;
  const int p = tid & 255, cp = tid >> 8, px = p & 15, py = p >> 4;
  const int tile_x = blockIdx.x % ntx, tile_y = blockIdx.x / ntx;
  const int x0 = tile_x * STAR_T, y0 = tile_y * STAR_T;
  const int z0 = zlo + blockIdx.y * zlen, z1 = min(zhi, z0 + zlen);     // output planes of this workgroup (global plane numbers)
  const int c0 = 8 * blockIdx.z;
  const long plane_rows = (long)nx * ny;
  auto plane_row0 = [&](int zz) -> long { return plane_rows * zz + (SLAB ? (zz < zs ? dlo : zz >= ze ? dhi : 0L) : 0L); };   // row of X of its point 0
  constexpr long loc = 0;
  // ---- compute lane: point (x0 + px, y0 + py), columns c0 + 2 cp, + 1
  const int gx = x0 + px, gy = y0 + py;
  const bool inside = gx < nx && gy < ny;
  const long own = inside ? (long)gx + (long)nx * gy : 0;
  const int ccol = c0 + 2 * cp;
  const bool cvalid = ccol < ncols;
  const int slot = (py + STAR_R) * STAR_PW + (px + STAR_R);
  // ---- staging units
  StarUnit su[3];
  const int si = tid & 3;
  const int scol = c0 + 2 * si;
  const bool svalid = scol < ncols;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int pt = (tid + 1024 * q) >> 2;
    int xx, yy, dst;
    if (pt < 256) { xx = pt & 15; yy = pt >> 4; dst = -2 - (si * 256 + pt); }                  // core of plane z + 6 -> corein (encoded below)
    else if (pt < 448) { const int j = pt - 256, ry = j / 12, a = j % 12; yy = ry; xx = a < 6 ? a - 6 : 10 + a;
                         dst = si * STAR_NP + (yy + STAR_R) * STAR_PW + (xx + STAR_R); }
    else if (pt < 640) { const int j = pt - 448, d = j >> 4; xx = j & 15; yy = d < 6 ? d - 6 : 10 + d;
                         dst = si * STAR_NP + (yy + STAR_R) * STAR_PW + (xx + STAR_R); }
    else { xx = 0; yy = 0; dst = -1; }
    const int ax = x0 + xx, ay = y0 + yy;
    su[q].dst = dst;
    su[q].src = (dst != -1 && ax >= 0 && ax < nx && ay >= 0 && ay < ny && svalid) ? ax + nx * ay : -1;
  }
  // out-tile flush: point tid >> 2, part si
  const int opt = tid >> 2, ox = x0 + (opt & 15), oy = y0 + (opt >> 4);
  const bool oinside = ox < nx && oy < ny && svalid;
  const long orow = oinside ? (long)ox + (long)nx * oy : 0;

  // ---- queue: planes z0 - 6 .. z0 + 5 into slots 0 .. 11
  v2d qv[STAR_Q];
#pragma unroll
  for (int t = 0; t < STAR_Q - 1; ++t) {
    const int zz = z0 - STAR_R + t;
    qv[t] = (inside && cvalid && zz >= zmin && zz < zmax) ? star_ld(x, ldx, own + plane_row0(zz), ccol) : v2d{0.0, 0.0};
  }
  qv[STAR_Q - 1] = v2d{0.0, 0.0};
  // staged loads of the first step: core of plane z0 + 6, arms of plane z0
  v2d st[3], stn[3];                                          // this step's staged rows / the next step's, in flight for a whole step
  auto stage_load = [&](int z) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int zz = su[q].dst < -1 ? z + STAR_R : z;          // core units fetch plane z + 6, arm units plane z
      stn[q] = (su[q].src >= 0 && zz >= zmin && zz < zmax) ? star_ld(x, ldx, (long)su[q].src + plane_row0(zz), scol) : v2d{0.0, 0.0};
    }
  };
  stage_load(z0);
  double dg = (inside && z0 < z1) ? diag[own + loc + plane_rows * z0] : NAN;     // diagonal of my point in the output plane (NaN: not mine to write)
  double dgo = NAN;                                                       // the same for the point whose result I flush
  long flush_plane = -1;
  v2d spw = v2d{0.0, 0.0}, sww = v2d{0.0, 0.0};                           // DOT: sums over my point's star rows

  for (int zb = z0; zb < z1; zb += STAR_Q) {
#define STAR_STEP(U)                                                                                                        \
    {                                                                                                                       \
      const int z = zb + (U);                                                                                               \
      if (z >= z1) break;                                                                                                   \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) st[q] = stn[q];                                                          \
      stage_load(z + 1);                                 /* requested a whole step before they are written to LDS */         \
      __syncthreads();                                   /* A: last step's LDS reads are done, its results are in outt */    \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                                       \
        if (su[q].dst >= 0) plane[su[q].dst] = st[q];                                                                       \
        else if (su[q].dst < -1) corein[-2 - su[q].dst] = st[q];                                                            \
      }                                                                                                                     \
      plane[cp * STAR_NP + slot] = qv[STAR_SLOT(U, 0)];                                                                        \
      if (flush_plane >= 0 && oinside && dgo == dgo)                                                                        \
        __builtin_nontemporal_store(outt[si * 256 + opt], reinterpret_cast<v2d*>(y + (size_t)(orow + loc + plane_rows * flush_plane) * ldy + scol)); \
      __syncthreads();                                   /* B */                                                            \
      qv[STAR_SLOT(U, STAR_R)] = corein[cp * 256 + p];      /* plane z + 6 */                                                   \
      dgo = oinside ? diag[orow + loc + plane_rows * z] : NAN;                                                              \
      flush_plane = z;                                                                                                      \
      const double dnext = (inside && z + 1 < z1) ? diag[own + loc + plane_rows * (z + 1)] : NAN;                            \
      const double d0 = dg == dg ? dg : 0.0;                                                                                \
      v2d acc = qv[STAR_SLOT(U, 0)] * d0;                                                                                      \
      _Pragma("unroll") for (int k = 1; k <= STAR_R; ++k) {                                                                 \
        const v2d zs = qv[STAR_SLOT(U, -k)] + qv[STAR_SLOT(U, k)];                                                                 \
        acc.x = fma(cf.cz[k], zs.x, acc.x); acc.y = fma(cf.cz[k], zs.y, acc.y);                                              \
      }                                                                                                                     \
      const v2d* pl = plane + cp * STAR_NP + slot;                                                                          \
      _Pragma("unroll") for (int k = 1; k <= STAR_R; ++k) {                                                                 \
        const v2d xs = pl[-k] + pl[k];                                                                                      \
        acc.x = fma(ISO ? cf.cz[k] : cf.cx[k], xs.x, acc.x); acc.y = fma(ISO ? cf.cz[k] : cf.cx[k], xs.y, acc.y);            \
        const v2d ys = pl[-k * STAR_PW] + pl[k * STAR_PW];                                                                   \
        acc.x = fma(ISO ? cf.cz[k] : cf.cy[k], ys.x, acc.x); acc.y = fma(ISO ? cf.cz[k] : cf.cy[k], ys.y, acc.y);            \
      }                                                                                                                     \
      outt[cp * 256 + p] = acc;                                                                                             \
      if (DOT && dg == dg) {                                                                                                \
        const v2d xc = qv[STAR_SLOT(U, 0)];                                                                                 \
        spw.x = fma(xc.x, acc.x, spw.x); spw.y = fma(xc.y, acc.y, spw.y);                                                    \
        sww.x = fma(acc.x, acc.x, sww.x); sww.y = fma(acc.y, acc.y, sww.y);                                                  \
      }                                                                                                                     \
      dg = dnext;                                                                                                           \
    }
    STAR_STEP(0) STAR_STEP(1) STAR_STEP(2) STAR_STEP(3) STAR_STEP(4) STAR_STEP(5) STAR_STEP(6)
    STAR_STEP(7) STAR_STEP(8) STAR_STEP(9) STAR_STEP(10) STAR_STEP(11) STAR_STEP(12)
#undef STAR_STEP
  }
  __syncthreads();
  if (flush_plane >= 0 && oinside && dgo == dgo)
    __builtin_nontemporal_store(outt[si * 256 + opt], reinterpret_cast<v2d*>(y + (size_t)(orow + loc + plane_rows * flush_plane) * ldy + scol));
  if (DOT) {
    // fixed-order reduction over the 256 points of every column pair: through the plane image (4 x 784 >= 2 x 1024 v2d)
    __syncthreads();
    plane[tid] = spw; plane[1024 + tid] = sww;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if (p < h) {
        const v2d a = plane[tid + h], b = plane[1024 + tid + h];
        plane[tid].x += a.x; plane[tid].y += a.y; plane[1024 + tid].x += b.x; plane[1024 + tid].y += b.y;
      }
      __syncthreads();
    }
    if (p == 0 && cvalid) {
      double* out = partial + ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 2 * ncols;
      out[ccol] = plane[tid].x; out[ccol + 1] = plane[tid].y;
      out[ncols + ccol] = plane[1024 + tid].x; out[ncols + ccol + 1] = plane[1024 + tid].y;
    }
  }
}
#undef STAR_SLOT

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
  std::vector<int> rem_rowptr, rem_col; std::vector<double> rem_val;  // every entry of the rows that are not clean (local columns)
};

static inline uint64_t star_bits(double v) { uint64_t b; memcpy(&b, &v, 8); return b; }

// The rows of one slab: local rows [0, nrows) are global rows row_begin + r; local column c < nrows is global column row_begin + c,
// c >= nrows is halo row c - nrows = global column ghost[c - nrows] (ascending).  One rank: row_begin = 0, no halo columns.
struct StarRows {
  int nrows, ncols_local; long row_begin, nglobal; const int* ghost; const int *rowptr, *colidx; const double* val;
  bool global_cols = false;                                           // colidx holds global columns already (partitioners)
  long gcol(int c) const { return global_cols ? (long)c : c < nrows ? row_begin + c : (long)ghost[c - nrows]; }
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
    for (int q = M.rowptr[r]; q < M.rowptr[r + 1]; ++q) { const long o = M.gcol(M.colidx[q]) - (M.row_begin + r); if (o != 0) seen.push_back(o < 0 ? -o : o); }
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
  // the slab must be whole planes (a partition cut inside a plane keeps the other forms: gcge_amd.dist.partition_by_nnz(align=))
  if (M.row_begin % sz != 0 || (long)nrows % sz != 0) return false;
  StarGeom& g = H->g;
  g.nx = (int)sy; g.ny = (int)(sz / sy); g.nz = (int)(M.nglobal / sz); H->R = R;
  g.zs = (int)(M.row_begin / sz); g.ze = g.zs + (int)((long)nrows / sz);
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
  if (!run((long)g.zmin * sz, (long)(g.zs - g.zmin) * sz, &g.lo_off)) return false;
  if (!run((long)g.ze * sz, (long)(g.zmax - g.ze) * sz, &g.hi_off)) return false;
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
          const long o = M.gcol(colidx[q]) - (M.row_begin + r);
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

// rows whose off-diagonal entries are exactly the star (truncated at the faces) -> diag[]; everything else -> remainder CSR
static bool star_build_host(const StarRows& M, StarHost* H) {
  const int nrows = M.nrows; const int* rowptr = M.rowptr; const int* colidx = M.colidx; const double* val = M.val;
  if (M.ncols_local != nrows && M.ghost == nullptr) return false;     // halo columns of unknown origin: the other forms
  if (!star_detect(M, H)) return false;
  const int nx = H->g.nx, ny = H->g.ny, nz = H->g.nz;
  const long sy = nx, sz = (long)nx * ny;
  H->diag.assign((size_t)nrows, NAN);
  H->rem_rowptr.assign((size_t)nrows + 1, 0);
  uint64_t cb[3][STAR_R + 1];
  for (int k = 0; k <= STAR_R; ++k) { cb[0][k] = star_bits(H->c.cx[k]); cb[1][k] = star_bits(H->c.cy[k]); cb[2][k] = star_bits(H->c.cz[k]); }
  long nclean = 0;
  std::vector<char>& clean = H->clean;
  clean.assign((size_t)nrows, 0);
  for (int r = 0; r < nrows; ++r) {
    const long gr = M.row_begin + r;
    const int gz = (int)(gr / sz), gy = (int)((gr - (long)gz * sz) / sy), gx = (int)(gr - (long)gz * sz - (long)gy * sy);
    const int g[3] = {gx, gy, gz}, dim[3] = {nx, ny, nz};
    int expect = 0;
    for (int a = 0; a < 3; ++a)
      for (int k = 1; k <= STAR_R; ++k) {
        if (cb[a][k] == 0) continue;                                  // (+0.0: no such neighbour in the star)
        expect += (g[a] - k >= 0) + (g[a] + k < dim[a]);
      }
    int matched = 0; bool ok = true, have_diag = false; double dv = 0.0;
    for (int q = rowptr[r]; q < rowptr[r + 1] && ok; ++q) {
      const long o = M.gcol(colidx[q]) - gr;
      if (o == 0) { if (have_diag) ok = false; have_diag = true; dv = val[q]; continue; }
      const long ao = o < 0 ? -o : o; const int sgn = o < 0 ? -1 : 1;
      int a, k;
      if (ao <= STAR_R) { a = 0; k = (int)ao; }
      else if (ao % sz == 0 && ao / sz <= STAR_R) { a = 2; k = (int)(ao / sz); }
      else if (ao % sy == 0 && ao / sy <= STAR_R) { a = 1; k = (int)(ao / sy); }
      else { ok = false; break; }
      const int nb = g[a] + sgn * k;
      if (nb < 0 || nb >= dim[a] || cb[a][k] == 0 || star_bits(val[q]) != cb[a][k]) { ok = false; break; }
      ++matched;
    }
    if (ok && matched == expect && !(dv != dv)) { clean[r] = 1; H->diag[r] = dv; ++nclean; }
  }
  H->nclean = nclean;
  if (2 * nclean < nrows) return false;
  for (int r = 0; r < nrows; ++r) H->rem_rowptr[r + 1] = H->rem_rowptr[r] + (clean[r] ? 0 : rowptr[r + 1] - rowptr[r]);
  H->rem_col.resize((size_t)H->rem_rowptr[nrows]); H->rem_val.resize((size_t)H->rem_rowptr[nrows]);
  for (int r = 0; r < nrows; ++r)
    if (!clean[r]) {
      memcpy(H->rem_col.data() + H->rem_rowptr[r], colidx + rowptr[r], (size_t)(rowptr[r + 1] - rowptr[r]) * sizeof(int));
      memcpy(H->rem_val.data() + H->rem_rowptr[r], val + rowptr[r], (size_t)(rowptr[r + 1] - rowptr[r]) * sizeof(double));
    }
  return true;
}

}  // namespace gcge

using namespace gcge;

static int g_star_mode = 0;   // 0 automatic, -1 never
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
extern "C" long gcge_hip_star_selfcheck_slab(int nrows, int ncols_local, long row_begin, long nglobal, const int* ghost, const int* rowptr,
                                             const int* colidx, const double* val, long* out) {
  StarHost H;
  const StarRows M = {nrows, ncols_local, row_begin, nglobal, ghost, rowptr, colidx, val};
  if (!star_build_host(M, &H)) return -1;
  const StarGeom& g = H.g;
  const long sy = g.nx, sz = (long)g.nx * g.ny;
  if (out) {
    out[0] = g.nx; out[1] = g.ny; out[2] = g.nz; out[3] = H.R; out[4] = H.nclean;
    out[5] = g.zs; out[6] = g.ze; out[7] = g.zmin; out[8] = g.zmax;
    out[9] = g.zmin < g.zs ? g.lo_off + sz * g.zmin : -1; out[10] = g.ze < g.zmax ? g.hi_off + sz * g.ze : -1;
  }
  long bad = 0;
  // where the sweep finds grid point (global row gq): the row of X, as the kernel computes it
  auto xrow = [&](long gq) -> long { const int zz = (int)(gq / sz); return (zz < g.zs ? g.lo_off : zz >= g.ze ? g.hi_off : g.mid_off) + gq; };
  std::vector<std::pair<long, uint64_t>> want, got;
  for (int r = 0; r < nrows; ++r) {
    want.clear(); got.clear();
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) want.emplace_back((long)colidx[q], star_bits(val[q]));   // local columns = rows of X
    if (H.diag[r] == H.diag[r]) {
      if (H.rem_rowptr[r + 1] != H.rem_rowptr[r]) ++bad;
      const long gr = row_begin + r;
      const int gz = (int)(gr / sz), gy = (int)((gr - (long)gz * sz) / sy), gx = (int)(gr - (long)gz * sz - (long)gy * sy);
      bool stored_diag = false;
      for (auto& w : want) stored_diag |= w.first == r;
      if (stored_diag || H.diag[r] != 0.0) got.emplace_back((long)r, star_bits(H.diag[r]));
      for (int k = 1; k <= STAR_R; ++k) {
        if (star_bits(H.c.cx[k])) { if (gx - k >= 0) got.emplace_back(xrow(gr - k), star_bits(H.c.cx[k])); if (gx + k < g.nx) got.emplace_back(xrow(gr + k), star_bits(H.c.cx[k])); }
        if (star_bits(H.c.cy[k])) { if (gy - k >= 0) got.emplace_back(xrow(gr - k * sy), star_bits(H.c.cy[k])); if (gy + k < g.ny) got.emplace_back(xrow(gr + k * sy), star_bits(H.c.cy[k])); }
        if (star_bits(H.c.cz[k])) {
          if (gz - k >= 0) { if (gz - k < g.zmin) ++bad; got.emplace_back(xrow(gr - k * sz), star_bits(H.c.cz[k])); }
          if (gz + k < g.nz) { if (gz + k >= g.zmax) ++bad; got.emplace_back(xrow(gr + k * sz), star_bits(H.c.cz[k])); }
        }
      }
    } else {
      for (int q = H.rem_rowptr[r]; q < H.rem_rowptr[r + 1]; ++q) got.emplace_back((long)H.rem_col[q], star_bits(H.rem_val[q]));
    }
    std::sort(want.begin(), want.end()); std::sort(got.begin(), got.end());
    if (want != got) ++bad;
  }
  return bad;
}
extern "C" long gcge_hip_star_selfcheck(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, long* out) {
  if (ncols_local != nrows) return -1;
  long o[11];
  const long bad = gcge_hip_star_selfcheck_slab(nrows, ncols_local, 0, nrows, nullptr, rowptr, colidx, val, o);
  if (out && bad >= 0) for (int i = 0; i < 5; ++i) out[i] = o[i];
  return bad;
}

extern "C" void gcge_hip_star_free(void* sm) {
  StarMat* S = (StarMat*)sm;
  if (!S) return;
  hipFree(S->d_diag);
  delete S;
}

// NULL: the matrix keeps the other forms.  Otherwise the device object, and through rem_* the CSR arrays of the remainder (rows
// that are not clean keep all their entries, clean rows are empty), owned by the object until gcge_hip_star_release_remainder.
// ghost: the global rows behind the halo columns nrows .. ncols_local - 1 (ascending; NULL on one rank).
static StarHost* g_star_last = nullptr;
extern "C" void* gcge_hip_star_build(int nrows, int ncols_local, long row_begin, long nglobal, const int* ghost, const int* rowptr,
                                     const int* colidx, const double* val, const int** rem_rowptr, const int** rem_col, const double** rem_val) {
  if (g_star_mode < 0 || nrows <= 0) return nullptr;
  StarHost* H = new StarHost();
  const StarRows M = {nrows, ncols_local, row_begin, nglobal, ghost, rowptr, colidx, val};
  if (!star_build_host(M, H)) { delete H; return nullptr; }
  StarMat* S = new StarMat();
  S->g = H->g; S->R = H->R; S->nclean = H->nclean; S->nrows = nrows; S->c = H->c;
  S->iso = memcmp(H->c.cx, H->c.cy, sizeof(H->c.cx)) == 0 && memcmp(H->c.cx, H->c.cz, sizeof(H->c.cx)) == 0;
  GCGE_HIP_CHECK(hipMalloc(&S->d_diag, (size_t)nrows * sizeof(double)));
  GCGE_HIP_CHECK(hipMemcpy(S->d_diag, H->diag.data(), (size_t)nrows * sizeof(double), hipMemcpyHostToDevice));
  std::vector<double>().swap(H->diag);
  *rem_rowptr = H->rem_rowptr.data(); *rem_col = H->rem_col.data(); *rem_val = H->rem_val.data();
  if (g_star_last) delete g_star_last;
  g_star_last = H;
  return S;
}
extern "C" void gcge_hip_star_release_remainder(void) { if (g_star_last) { delete g_star_last; g_star_last = nullptr; } }

extern "C" const unsigned char* gcge_hip_star_host_mask(void) {   // 1: a star row (valid until gcge_hip_star_release_remainder)
  return g_star_last ? (const unsigned char*)g_star_last->clean.data() : nullptr;
}
extern "C" void gcge_hip_star_stats(const void* sm, long* out) {   // nx, ny, nz, arm length, clean rows, rows, first / last + 1 plane of the slab
  const StarMat* S = (const StarMat*)sm;
  out[0] = S->g.nx; out[1] = S->g.ny; out[2] = S->g.nz; out[3] = S->R; out[4] = S->nclean; out[5] = S->nrows; out[6] = S->g.zs; out[7] = S->g.ze;
}

// The output planes that need no halo row — [*ilo, *ihi) — so that they can be swept while the halo is in flight (the sweep loads
// STAR_R planes on either side of an output plane whatever the arm length).  Returns 0 when the slab has no halo or no such plane.
extern "C" int gcge_hip_star_interior(const void* sm, int* ilo, int* ihi) {
  const StarGeom& g = ((const StarMat*)sm)->g;
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
  const int ntx = (S->g.nx + STAR_T - 1) / STAR_T, nty = (S->g.ny + STAR_T - 1) / STAR_T, npass = (ncols + 7) / 8;
  // z ranges: ONE where the patches x passes already give every CU two workgroups' worth of work (each range re-reads 12 planes
  // of warm-up: 171^3, 64 columns: 1 / 2 / 3 / 4 ranges = 3.93 / 4.26 / 4.35 / 4.47 ms for the whole product), otherwise enough
  // ranges of at least 24 planes to get there
  int zchunks = (int)std::max(1L, std::min((long)nzl / 24, (2L * 256 + (long)ntx * nty * npass - 1) / ((long)ntx * nty * npass)));
  const int zlen = (nzl + zchunks - 1) / zchunks;
  zchunks = (nzl + zlen - 1) / zlen;
  const int nb = ntx * nty * zchunks;
  if (count_only) return nb;
  const dim3 grid((unsigned)(ntx * nty), (unsigned)zchunks, (unsigned)npass);
  // operands in global plane numbering (see the kernel): the slab's first plane is plane zs of the grid
  const StarGeom& g = S->g;
  const long shift = g.mid_off;                                       // = - plane_rows * zs
  const double* xv = (const double*)((uintptr_t)d_x + (uintptr_t)(shift * ldx * (long)sizeof(double)));
  double* yv = (double*)((uintptr_t)d_y + (uintptr_t)(shift * ldy * (long)sizeof(double)));
  const double* dv = (const double*)((uintptr_t)S->d_diag + (uintptr_t)(shift * (long)sizeof(double)));
  const long dlo = g.lo_off - g.mid_off, dhi = g.hi_off - g.mid_off;
  const bool slab = g.zmin < g.zs || g.ze < g.zmax, iso = S->iso;
#define STAR_LAUNCH(DOT, ISO, SLAB)                                                                                                     \
  hipLaunchKernelGGL((spmm_star_kernel<DOT, ISO, SLAB>), grid, dim3(1024), 0, stream, g.nx, g.ny, g.zs, g.ze, g.zmin, g.zmax, dlo, dhi, S->c, dv, xv, \
                     (size_t)ldx, yv, (size_t)ldy, ncols, zlo, zhi, zlen, ntx, part)
  if (part == nullptr) {
    if (iso) { if (slab) STAR_LAUNCH(false, true, true); else STAR_LAUNCH(false, true, false); }
    else     { if (slab) STAR_LAUNCH(false, false, true); else STAR_LAUNCH(false, false, false); }
  } else {
    if (iso) { if (slab) STAR_LAUNCH(true, true, true); else STAR_LAUNCH(true, true, false); }
    else     { if (slab) STAR_LAUNCH(true, false, true); else STAR_LAUNCH(true, false, false); }
  }
#undef STAR_LAUNCH
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
