// K1 (tile path) — CSR SpMM with LDS-staged X tiles for matrices WITHOUT a row-pattern form.
//
// Same contract as spmm.hip / spmm_pad8.hip (Y[:, 0:m) = A X[:, 0:m), reference app/app_ccs.c:50-139; the 4-column
// CSR precedent of the reference is app/app_phg.c:172-217).  The generic kernels gather one 128-byte X-row segment
// per non-zero through L1/L2 — 36-70 gathers per row on the matrices of BASELINE config 5 (12th-order stencil + dense
// "atom" blocks), ten times the algorithmic bytes, bound by the texture-address / L1 path at 13 % of the HBM roofline
// (profiles/r02_spmm_generic).  Here the rows are cut into TILES whose rows share most of their columns:
//   * on matrices whose offsets (column - row) show a lexicographic grid (strides 1, sy, sz read off the offset
//     histogram) a tile is a brick of bx x by x bz grid points; otherwise a run of consecutive rows;
//   * at upload the union of the columns a tile's rows reference becomes the tile's X list, every non-zero keeps a
//     16-bit POSITION in that list instead of a 32-bit column (10 B per non-zero instead of 12); a tile whose union
//     does not fit the LDS tile is split in two (fewer rows, smaller union) until it does; the entries a row has beyond
//     40 go to a small overflow list that a second launch adds (rows that long are the business of spmm_dense.hip);
//   * one workgroup per tile, whose (value, position) entries it keeps in REGISTERS for all passes (the matrix is read
//     once whatever the block width); per 8-column pass the tile's X rows are staged in LDS ONCE (coalesced 64-byte
//     segments, swizzled so that neighbouring positions fall on different banks; double-buffered: the next pass is
//     in flight while this one is multiplied), then every non-zero is one LDS read instead of one L1/L2 gather.  Rows are laid out 32 to a wave (two lanes per row, 4 columns each,
//     ELL inside a 32-row slice), so the (value, position) stream is read with coalesced loads and nothing is reduced
//     across lanes: the dense atom blocks broadcast (all rows of a slice read the same position), stencil rows read
//     consecutive positions — both conflict-free under the swizzle.
// Padding entries carry value 0 and position 0.  Sums run in a fixed order: bit-reproducible from run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <thread>
#include <unordered_map>
#include <vector>
#include "gcge_hip_internal.h"

namespace gcge {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TILE_ROWS = 128;    // rows per tile: 4 slices of 32
constexpr int TILE_CAP = 1088;    // X rows per tile in LDS (2 buffers x 68 KB at 8 columns); 8x4x4 brick of a +-6 star stencil: exactly 1088
constexpr int STEP_DOUBLES = 40;  // one ELL step of a slice: 32 values + 32 16-bit positions = 320 B
constexpr int TILE_MAXW = 40;     // ELL steps a slice may have (two waves share them: 20 steps each stay in registers); longer rows overflow

struct TileHdr { int row_off, nrows, ucol_off, nu; int step_off[4]; int width[4]; };

struct TileMat {
  int ntiles; long nsteps, nnz, nucols; int nrows, bx, by, bz; long sy, sz;   // sy == 0: runs of consecutive rows
  TileHdr* d_th; int* d_rows; int* d_ucols; double* d_steps;
  int nov; long ov_nnz; int* d_ov_rows; int* d_ov_ptr; int* d_ov_col; double* d_ov_val;   // overflow: entries beyond TILE_MAXW per row
};

// One workgroup (8 waves) per tile; wave w: slice w & 3, half w >> 2 of the slice's ELL steps.
//   * the (value, position) steps of a wave — at most MAXH = TILE_MAXW / 2 — are loaded ONCE and stay in registers for
//     all passes: the matrix is read once per product whatever the block width;
//   * a pass covers 8 columns; the X rows of pass p + 1 are requested into registers before pass p is computed from one
//     LDS buffer and written into the other buffer afterwards: the staging latency is hidden behind the arithmetic
//     with ONE workgroup per CU (the two X buffers and the Y tile fill the LDS);
//   * LDS image: position p, column pair c (0..3) at p*4 + (c ^ ((p >> 2) & 1)) (16-byte units): the 16 lanes a
//     ds_read_b128 serves together hold 8 rows x 2 halves and hit 16 different bank quads when the rows read
//     neighbouring positions (stencil rows) or the same one.
template <int MAXH>
__global__ __launch_bounds__(512) void spmm_tile_kernel(
    const TileHdr* __restrict__ th, const int* __restrict__ rows, const int* __restrict__ ucols,
    const double* __restrict__ steps, const double* __restrict__ x, size_t ldx, double* __restrict__ y, size_t ldy,
    int ncols, int ntiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  v2d* xt = reinterpret_cast<v2d*>(smem_raw);        // 2 buffers x TILE_CAP positions x 4 column pairs
  v2d* yt = xt + (size_t)2 * TILE_CAP * 4;           // TILE_ROWS rows x 4 column pairs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row32 = lane >> 1, h = lane & 1, slice = wave & 3, kh = wave >> 2;
  // blocks are dealt round-robin to the 8 XCDs: every XCD walks one contiguous eighth of the tiles, so bricks that
  // share halo rows run on the same L2 at about the same time (tile = blockIdx, all XCDs in one z-region for the
  // Infinity Cache: 5.88 against 5.80 ms, profiles/r03_spmm_generic/34_...)
  const int per = (ntiles + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const TileHdr* __restrict__ T = th + tile;
  const int nu = T->nu, nr = T->nrows;
  // ---- my share of the slice's steps, into registers
  double sval[MAXH]; unsigned spos[MAXH];
  const int W = T->width[slice];
  const int cnt = kh ? (W >> 1) : ((W + 1) >> 1), k0 = kh ? ((W + 1) >> 1) : 0;
  {
    const double* __restrict__ sb = steps + (size_t)T->step_off[slice] * STEP_DOUBLES;
    const int last = max(W - 1, 0);      // W == 0: step_off still points into the array (one spare step at its end)
#pragma unroll
    for (int i = 0; i < MAXH; ++i) {
      const double* sp = sb + (size_t)min(k0 + i, last) * STEP_DOUBLES;
      const double v = sp[row32];
      const unsigned p = reinterpret_cast<const unsigned short*>(sp + 32)[row32];
      sval[i] = i < cnt ? v : 0.0;
      spos[i] = i < cnt ? p : 0u;
    }
  }
  // ---- staging: position su + 128 q, 16-byte part si
  const int su = tid >> 2, si = tid & 3;
  constexpr int NQ = (TILE_CAP + 127) / 128;
  int col[NQ]; v2d sv[NQ];
  {
    const int* __restrict__ uc = ucols + T->ucol_off;
#pragma unroll
    for (int q = 0; q < NQ; ++q) col[q] = uc[min(su + 128 * q, nu - 1)];
  }
  auto load_x = [&](int pass) {
    const int c = 8 * pass + 2 * si;
    const double* __restrict__ xs = x + (c < ncols ? c : 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) sv[q] = *reinterpret_cast<const v2d*>(xs + (size_t)col[q] * ldx);
  };
  auto store_x = [&](int buf) {
    v2d* dst = xt + (size_t)buf * TILE_CAP * 4;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int u = su + 128 * q;
      if (u < nu) dst[u * 4 + (si ^ ((u >> 2) & 1))] = sv[q];
    }
  };
  load_x(0);
  store_x(0);
  __syncthreads();
  const int npass = (ncols + 7) >> 3;
  const int r = slice * 32 + row32;
  v2d* yr = yt + (size_t)r * 4 + 2 * h;
  for (int pass = 0; pass < npass; ++pass) {
    const bool more = pass + 1 < npass;
    if (more) load_x(pass + 1);
    const v2d* __restrict__ xb = xt + (size_t)(pass & 1) * TILE_CAP * 4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int i0 = 0; i0 < MAXH; i0 += 4) {
      if (i0 < cnt) {                     // wave-uniform: groups of four steps beyond my share are skipped
#pragma unroll
        for (int i = i0; i < i0 + 4 && i < MAXH; ++i) {
          const v2d* xr = xb + spos[i] * 4;
          const unsigned sw = (spos[i] >> 2) & 1;
          const v2d x0 = xr[(2 * h) ^ sw], x1 = xr[(2 * h + 1) ^ sw];
          a0 = fma(sval[i], x0.x, a0); a1 = fma(sval[i], x0.y, a1);
          a2 = fma(sval[i], x1.x, a2); a3 = fma(sval[i], x1.y, a3);
        }
      }
    }
    if (more) store_x((pass + 1) & 1);    // that buffer was last read in pass - 1, which every wave has left
    // the two halves of a slice meet in the Y tile (second half stored, first half added), then the tile's rows leave as
    // 64-byte segments
    if (kh == 1) { yr[0] = v2d{a0, a1}; yr[1] = v2d{a2, a3}; }
    __syncthreads();
    if (kh == 0) { const v2d p0 = yr[0], p1 = yr[1]; yr[0] = v2d{a0 + p0.x, a1 + p0.y}; yr[1] = v2d{a2 + p1.x, a3 + p1.y}; }
    __syncthreads();
    {
      const int c = 8 * pass + 2 * si;
      if (su < nr && c < ncols)
        __builtin_nontemporal_store(yt[su * 4 + si], reinterpret_cast<v2d*>(y + (size_t)rows[T->row_off + su] * ldy + c));
    }
    __syncthreads();
  }
}

// Y[row] += (entries a row has beyond TILE_MAXW steps) X: one wave per such row, a lane per column
__global__ __launch_bounds__(256) void spmm_tile_overflow_kernel(int nov, const int* __restrict__ ov_rows, const int* __restrict__ ov_ptr,
    const int* __restrict__ ov_col, const double* __restrict__ ov_val, const double* __restrict__ x, size_t ldx,
    double* __restrict__ y, size_t ldy, int ncols) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= nov) return;
  const int row = ov_rows[i], p0 = ov_ptr[i], p1 = ov_ptr[i + 1];
  for (int c0 = 0; c0 < ncols; c0 += 64) {
    const int c = c0 + lane < ncols ? c0 + lane : 0;
    double acc = 0.0;
    for (int p = p0; p < p1; ++p) acc = fma(ov_val[p], x[(size_t)ov_col[p] * ldx + c], acc);
    if (c0 + lane < ncols) y[(size_t)row * ldy + c0 + lane] += acc;
  }
}

// ---------------------------------------------------------------------------------------------- upload-time builder
struct TileOut {   // what one builder thread produced for its range of tiles
  std::vector<TileHdr> th; std::vector<int> rows, ucols; std::vector<double> steps;
  std::vector<int> ov_rows, ov_ptr, ov_col; std::vector<double> ov_val;
};

// strides of a lexicographic grid read off the offsets (column - row) that most rows share: the positive frequent offsets fall into
// clusters (ratio > 3 between neighbours starts a new one); the smallest member of the second / third cluster is the
// row stride along y / z.  Returns false when the first cluster does not start at 1 or there is no second cluster.
static bool detect_grid(int nrows, const int* rowptr, const int* colidx, long* sy, long* sz) {
  std::unordered_map<long, int> hist;
  const int nsamp = std::min(nrows, 4096);
  for (int t = 0; t < nsamp; ++t) {
    const int r = (int)((long)t * nrows / nsamp);
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) { const long o = (long)colidx[q] - r; if (o > 0) ++hist[o]; }
  }
  std::vector<long> offs;
  for (auto& kv : hist) if (kv.second * 4 >= nsamp) offs.push_back(kv.first);
  if (offs.empty()) return false;
  std::sort(offs.begin(), offs.end());
  if (offs[0] != 1) return false;
  std::vector<long> heads = {offs[0]};
  for (size_t i = 1; i < offs.size(); ++i) if (offs[i] > 3 * offs[i - 1]) heads.push_back(offs[i]);
  if (heads.size() < 2 || heads[1] < 8) return false;
  *sy = heads[1];
  *sz = heads.size() >= 3 ? heads[2] : 0;
  if (*sz != 0 && (*sz % *sy != 0 || *sz / *sy < 4)) *sz = 0;
  return true;
}

struct TileScratch { std::vector<int> stamp, lid, ulist, order; };

// one tile from the rows rw[0, nr): split in two when its union of columns does not fit the LDS tile
static void build_tile(const int* rw, int nr, TileScratch& S, int& stamp_id, const int* rowptr, const int* colidx, const double* val,
                       TileOut* out) {
  const int id = ++stamp_id;
  S.ulist.clear();
  for (int i = 0; i < nr; ++i)
    for (int q = rowptr[rw[i]]; q < rowptr[rw[i] + 1]; ++q) {
      const int c = colidx[q];
      if (S.stamp[c] != id) { S.stamp[c] = id; S.ulist.push_back(c); }
    }
  const int nu = (int)S.ulist.size();
  if (nu > TILE_CAP && nr > 1) {
    build_tile(rw, nr / 2, S, stamp_id, rowptr, colidx, val, out);
    build_tile(rw + nr / 2, nr - nr / 2, S, stamp_id, rowptr, colidx, val, out);
    return;
  }
  // (a single row with more than TILE_CAP entries: the builder refuses such matrices before it gets here)
  std::sort(S.ulist.begin(), S.ulist.end());                       // ascending columns: neighbouring grid points, neighbouring positions
  for (int i = 0; i < nu; ++i) S.lid[S.ulist[i]] = i;
  // rows longest first (stable: equal rows keep their natural order)
  std::vector<int> rs(rw, rw + nr);
  std::stable_sort(rs.begin(), rs.end(), [&](int a, int b) { return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b]; });
  TileHdr T; T.row_off = (int)out->rows.size(); T.nrows = nr; T.ucol_off = (int)out->ucols.size(); T.nu = nu;
  for (int r : rs) out->rows.push_back(r);
  for (int i = 0; i < nu; ++i) out->ucols.push_back(S.ulist[i]);
  for (int s = 0; s < 4; ++s) {
    int W = 0;
    for (int rr = 32 * s; rr < std::min(nr, 32 * s + 32); ++rr) W = std::max(W, std::min(TILE_MAXW, rowptr[rs[rr] + 1] - rowptr[rs[rr]]));
    T.width[s] = W; T.step_off[s] = (int)(out->steps.size() / STEP_DOUBLES);
    const size_t base = out->steps.size();
    out->steps.resize(base + (size_t)W * STEP_DOUBLES, 0.0);
    for (int rr = 32 * s; rr < std::min(nr, 32 * s + 32); ++rr) {
      const int r = rs[rr]; int k = 0;
      bool ov = false;
      for (int q = rowptr[r]; q < rowptr[r + 1]; ++q, ++k) {
        if (k < TILE_MAXW) {
          double* sp = out->steps.data() + base + (size_t)k * STEP_DOUBLES;
          sp[rr - 32 * s] = val[q];
          reinterpret_cast<unsigned short*>(sp + 32)[rr - 32 * s] = (unsigned short)S.lid[colidx[q]];
        } else {
          if (!ov) { ov = true; out->ov_rows.push_back(r); out->ov_ptr.push_back((int)out->ov_col.size()); }
          out->ov_col.push_back(colidx[q]); out->ov_val.push_back(val[q]);
        }
      }
    }
  }
  out->th.push_back(T);
}

static void build_range(int t0, int t1, const std::vector<int>& trow_off, const std::vector<int>& trows, int ncols_local,
                        const int* rowptr, const int* colidx, const double* val, TileOut* out) {
  TileScratch S;
  S.stamp.assign((size_t)ncols_local, 0); S.lid.assign((size_t)ncols_local, 0);
  int stamp_id = 0;
  for (int t = t0; t < t1; ++t)
    build_tile(trows.data() + trow_off[t], trow_off[t + 1] - trow_off[t], S, stamp_id, rowptr, colidx, val, out);
}

}  // namespace gcge

using namespace gcge;

static int g_tile_mode = 0;   // 0 automatic (remainders of the block form that qualify), 1 every matrix without a pattern form, 2 every matrix (tests: next to a pattern form), -1 never
extern "C" void gcge_hip_spmm_tile_mode(int mode) { g_tile_mode = mode; }
extern "C" int gcge_hip_spmm_tile_mode_get(void) { return g_tile_mode; }

extern "C" void gcge_hip_tile_free(void* tm) {
  TileMat* T = (TileMat*)tm;
  if (!T) return;
  hipFree(T->d_th); hipFree(T->d_rows); hipFree(T->d_ucols); hipFree(T->d_steps);
  hipFree(T->d_ov_rows); hipFree(T->d_ov_ptr); hipFree(T->d_ov_col); hipFree(T->d_ov_val);
  delete T;
}

// host side of the upload: tiles and ELL steps of rows [0, nrows) with LOCAL column indices in [0, ncols_local)
struct TileHost {
  std::vector<TileHdr> th; std::vector<int> rows, ucols; std::vector<double> steps;
  std::vector<int> ov_rows, ov_ptr, ov_col; std::vector<double> ov_val;
  int bx = 0, by = 0, bz = 0; long sy = 0, sz = 0;
};
static bool tile_build_host(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, TileHost* H) {
  for (int r = 0; r < nrows; ++r) if (rowptr[r + 1] - rowptr[r] > TILE_CAP) return false;   // a row must fit the LDS tile on its own
  long sy = 0, sz = 0;
  int bx = TILE_ROWS, by = 1, bz = 1;
  const bool grid = detect_grid(nrows, rowptr, colidx, &sy, &sz);
  if (grid) { if (sz > 0) { bx = 8; by = 4; bz = 4; } else { bx = 16; by = 8; bz = 1; sz = (long)nrows + sy; } }
  // tiles: bricks in grid order (x fastest), rows of a brick in natural order
  std::vector<int> trow_off, trows;
  trows.reserve((size_t)nrows);
  if (!grid) {
    for (int r0 = 0; r0 < nrows; r0 += TILE_ROWS) { trow_off.push_back((int)trows.size()); for (int r = r0; r < std::min(nrows, r0 + TILE_ROWS); ++r) trows.push_back(r); }
  } else {
    const long nx = sy, ny = sz / sy, nz = ((long)nrows + sz - 1) / sz;
    // brick order: x fastest (y- or z-fastest orders measured the same: 5.86 / 5.83 / 5.75 ms on the SiO2-like matrix —
    // the staged rows come over the fabric either way, profiles/r03_spmm_generic)
    const long nbx = (nx + bx - 1) / bx, nby = (ny + by - 1) / by, nbz = (nz + bz - 1) / bz;
    for (long t = 0; t < nbx * nby * nbz; ++t) {
      const long ix = t % nbx, iy = (t / nbx) % nby, iz = t / (nbx * nby);
      const long x0 = ix * bx, y0 = iy * by, z0 = iz * bz;
      const size_t before = trows.size();
      for (long z = z0; z < std::min(nz, z0 + bz); ++z) for (long yy = y0; yy < std::min(ny, y0 + by); ++yy)
        for (long xx = x0; xx < std::min(nx, x0 + bx); ++xx) { const long r = xx + sy * yy + sz * z; if (r < nrows) trows.push_back((int)r); }
      if (trows.size() > before) trow_off.push_back((int)before);
    }
  }
  const int nbricks = (int)trow_off.size();
  trow_off.push_back((int)trows.size());
  if ((long)trows.size() != nrows) return false;
  // bricks are independent: a contiguous range per thread, concatenated afterwards
  unsigned hw = std::thread::hardware_concurrency();
  int nth = (int)std::min<unsigned>(hw ? hw : 4, 16);
  if (const char* e = getenv("OMP_NUM_THREADS")) nth = std::max(1, std::min(nth, atoi(e)));
  nth = std::max(1, std::min(nth, nbricks / 64 + 1));
  std::vector<TileOut> outs((size_t)nth);
  {
    std::vector<std::thread> pool;
    for (int i = 0; i < nth; ++i) {
      const int t0 = (int)((long)nbricks * i / nth), t1 = (int)((long)nbricks * (i + 1) / nth);
      pool.emplace_back(build_range, t0, t1, std::cref(trow_off), std::cref(trows), ncols_local, rowptr, colidx, val, &outs[i]);
    }
    for (auto& p : pool) p.join();
  }
  size_t nst = 0, nuc = 0, nov = 0;
  for (auto& o : outs) { nst += o.steps.size(); nuc += o.ucols.size(); nov += o.ov_col.size(); }
  if (nst / STEP_DOUBLES + 1 > 2147483647UL || nuc > 2147483647UL || nov > 2147483647UL) return false;
  for (auto& o : outs) {
    const int o_rows = (int)H->rows.size(), o_uc = (int)H->ucols.size(), o_st = (int)(H->steps.size() / STEP_DOUBLES), o_ov = (int)H->ov_col.size();
    for (auto& t : o.th) { t.row_off += o_rows; t.ucol_off += o_uc; for (int s = 0; s < 4; ++s) t.step_off[s] += o_st; H->th.push_back(t); }
    H->rows.insert(H->rows.end(), o.rows.begin(), o.rows.end());
    H->ucols.insert(H->ucols.end(), o.ucols.begin(), o.ucols.end());
    H->steps.insert(H->steps.end(), o.steps.begin(), o.steps.end());
    H->ov_rows.insert(H->ov_rows.end(), o.ov_rows.begin(), o.ov_rows.end());
    for (int q : o.ov_ptr) H->ov_ptr.push_back(q + o_ov);
    H->ov_col.insert(H->ov_col.end(), o.ov_col.begin(), o.ov_col.end());
    H->ov_val.insert(H->ov_val.end(), o.ov_val.begin(), o.ov_val.end());
    TileOut().steps.swap(o.steps);
  }
  H->ov_ptr.push_back((int)H->ov_col.size());
  H->steps.resize(H->steps.size() + STEP_DOUBLES, 0.0);   // one spare step: empty slices point at it
  H->bx = bx; H->by = by; H->bz = bz; H->sy = grid ? sy : 0; H->sz = grid ? sz : 0;
  return true;
}

// Structural self-check of the upload (host only, no device needed; tests): expands tiles -> ELL steps (+ overflow) back
// into (row, column, value) triples and compares them with the CSR arrays, bit for bit.  0: identical; > 0: number of
// differences (missing, extra or altered entries, rows not covered exactly once, positions outside a tile's list, a
// tile over the LDS capacity); -1: the matrix does not take this form.
extern "C" long gcge_hip_tile_selfcheck(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val,
                                        double* xrows_per_row, double* ell_per_nnz, long* strides) {
  TileHost H;
  if (!tile_build_host(nrows, ncols_local, rowptr, colidx, val, &H)) return -1;
  if (xrows_per_row) *xrows_per_row = (double)H.ucols.size() / nrows;
  if (ell_per_nnz) *ell_per_nnz = 32.0 * (double)(H.steps.size() / STEP_DOUBLES - 1) / (double)rowptr[nrows];
  if (strides) { strides[0] = H.sy; strides[1] = H.sz; }
  long bad = 0;
  std::vector<int> seen((size_t)nrows, 0), ov_of((size_t)nrows, -1);
  for (size_t i = 0; i < H.ov_rows.size(); ++i) { if (ov_of[H.ov_rows[i]] != -1) ++bad; ov_of[H.ov_rows[i]] = (int)i; }
  std::vector<std::pair<int, double>> got, want;
  auto less = [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; };
  for (const TileHdr& T : H.th) {
    if (T.nrows > TILE_ROWS || T.nu > TILE_CAP) ++bad;
    for (int s = 0; s < 4; ++s) if (T.width[s] > TILE_MAXW) ++bad;
    for (int rr = 0; rr < T.nrows; ++rr) {
      const int r = H.rows[(size_t)T.row_off + rr];
      if (r < 0 || r >= nrows) { ++bad; continue; }
      ++seen[r];
      got.clear();
      const int s = rr / 32, q = rr % 32;
      for (int k = 0; k < T.width[s]; ++k) {
        const double* sp = H.steps.data() + ((size_t)T.step_off[s] + k) * STEP_DOUBLES;
        const int pos = reinterpret_cast<const unsigned short*>(sp + 32)[q];
        if (pos >= T.nu) { ++bad; continue; }
        uint64_t bits; memcpy(&bits, &sp[q], 8);
        if (bits == 0 && pos == 0) continue;      // padding (an explicit +0.0 entry at position 0 is indistinguishable and harmless)
        got.emplace_back(H.ucols[(size_t)T.ucol_off + pos], sp[q]);
      }
      if (ov_of[r] >= 0) for (int p = H.ov_ptr[ov_of[r]]; p < H.ov_ptr[ov_of[r] + 1]; ++p) got.emplace_back(H.ov_col[p], H.ov_val[p]);
      want.clear();
      for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) want.emplace_back(colidx[p], val[p]);
      std::sort(got.begin(), got.end(), less); std::sort(want.begin(), want.end(), less);
      size_t g = 0;
      for (const auto& w : want) {
        uint64_t vb; memcpy(&vb, &w.second, 8);
        if (vb == 0) { if (g < got.size() && got[g].first == w.first && got[g].second == 0.0) ++g; continue; }
        if (g >= got.size() || got[g].first != w.first || memcmp(&got[g].second, &w.second, 8) != 0) { ++bad; continue; }
        ++g;
      }
      if (g != got.size()) ++bad;
    }
  }
  for (int r = 0; r < nrows; ++r) if (seen[r] != 1) ++bad;
  return bad;
}

template <class T>
static T* tile_to_device(const std::vector<T>& v) {
  T* d = nullptr;
  GCGE_HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) GCGE_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

// NULL: the matrix keeps the generic kernels (path not applicable, a row longer than the LDS tile, more than 2^31 ELL steps).
// Automatic rule (mode 0; profiles/r03_spmm_generic): the tile form pays on matrices of SHORT rows on a detected grid whose
// bricks stage few X rows per matrix row — the remainder of the SiO2-like matrix once its long rows sit in dense blocks:
// 5.0 against 6.7 ms for the pad-8 kernel — and not on matrices with long rows (the matrix stream then outweighs the
// gathers saved) or without grid structure (runs of rows share too little).  `remainder`: the caller is spmm_dense.hip.
extern "C" void* gcge_hip_tile_build_for(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, int remainder) {
  if (g_tile_mode < 0 || nrows <= 0) return nullptr;
  if (g_tile_mode == 0 && (!remainder || nrows < 4096 || rowptr[nrows] > 64L * nrows)) return nullptr;
  TileHost H;
  if (!tile_build_host(nrows, ncols_local, rowptr, colidx, val, &H)) return nullptr;
  if (g_tile_mode == 0) {
    const double xrows = (double)H.ucols.size() / nrows, ov = (double)H.ov_col.size() / (double)std::max<long>(1, rowptr[nrows]);
    if (H.sy == 0 || xrows > 10.0 || ov > 0.05) return nullptr;
  }
  TileMat* T = new TileMat();
  T->ntiles = (int)H.th.size(); T->nsteps = (long)(H.steps.size() / STEP_DOUBLES) - 1; T->nnz = rowptr[nrows];
  T->nucols = (long)H.ucols.size(); T->nrows = nrows;
  T->bx = H.bx; T->by = H.by; T->bz = H.bz; T->sy = H.sy; T->sz = H.sz;
  T->d_th = tile_to_device(H.th); T->d_rows = tile_to_device(H.rows); T->d_ucols = tile_to_device(H.ucols);
  T->d_steps = tile_to_device(H.steps);
  T->nov = (int)H.ov_rows.size(); T->ov_nnz = (long)H.ov_col.size();
  T->d_ov_rows = tile_to_device(H.ov_rows); T->d_ov_ptr = tile_to_device(H.ov_ptr);
  T->d_ov_col = tile_to_device(H.ov_col); T->d_ov_val = tile_to_device(H.ov_val);
  return T;
}
extern "C" void* gcge_hip_tile_build(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val) {
  if (g_tile_mode <= 0) return nullptr;   // whole matrices: on request only (tests, measurements)
  return gcge_hip_tile_build_for(nrows, ncols_local, rowptr, colidx, val, 0);
}

// what the upload found: tiles, X rows staged per matrix row (the request factor), ELL entries per non-zero (padding),
// entries in overflow rows
extern "C" void gcge_hip_tile_stats(const void* tm, long* ntiles, long* ov_nnz, double* xrows_per_row, double* ell_per_nnz,
                                    int* brick, long* strides) {
  const TileMat* T = (const TileMat*)tm;
  if (ntiles) *ntiles = T->ntiles;
  if (ov_nnz) *ov_nnz = T->ov_nnz;
  if (xrows_per_row) *xrows_per_row = (double)T->nucols / T->nrows;
  if (ell_per_nnz) *ell_per_nnz = 32.0 * (double)T->nsteps / (double)T->nnz;
  if (brick) { brick[0] = T->bx; brick[1] = T->by; brick[2] = T->bz; }
  if (strides) { strides[0] = T->sy; strides[1] = T->sz; }
}

// Y[:, 0:ncols) = A X[:, 0:ncols).  -1: alignment contract not met (the caller keeps the generic kernels).
extern "C" int gcge_hip_tile_spmm(const void* tm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream) {
  const TileMat* T = (const TileMat*)tm;
  if (ncols <= 0 || T->ntiles == 0) return 0;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15) || d_x == d_y) return -1;
  static bool attr_set = false;
  const size_t lds = (size_t)(2 * TILE_CAP + TILE_ROWS) * 4 * sizeof(v2d);
  if (!attr_set) {
    GCGE_HIP_CHECK(hipFuncSetAttribute((const void*)spmm_tile_kernel<TILE_MAXW / 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int per = (T->ntiles + 7) / 8;
  hipLaunchKernelGGL((spmm_tile_kernel<TILE_MAXW / 2>), dim3((unsigned)(8 * per)), dim3(512), lds, (hipStream_t)stream, T->d_th, T->d_rows,
                     T->d_ucols, T->d_steps, d_x, (size_t)ldx, d_y, (size_t)ldy, ncols, T->ntiles);
  if (T->nov > 0)
    hipLaunchKernelGGL(spmm_tile_overflow_kernel, dim3((unsigned)((T->nov + 3) / 4)), dim3(256), 0, (hipStream_t)stream, T->nov, T->d_ov_rows,
                       T->d_ov_ptr, T->d_ov_col, T->d_ov_val, d_x, (size_t)ldx, d_y, (size_t)ldy, ncols);
  return (int)hipGetLastError();
}
