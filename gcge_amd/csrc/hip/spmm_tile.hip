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
//     16-bit POSITION in that list instead of a 32-bit column (10 B per non-zero instead of 12); unions longer than the
//     LDS tile are cut into chunks, most frequently used columns first, and the rows' partial sums stay in registers
//     from chunk to chunk — no remainder matrix, no second kernel;
//   * one workgroup per tile; per 8-column pass the chunk's X rows are staged in LDS ONCE (coalesced 64-byte
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
constexpr int TILE_CAP = 1088;    // X rows per chunk in LDS (2 buffers x 68 KB at 8 columns); 8x4x4 brick of a +-6 star stencil: exactly 1088
constexpr int STEP_DOUBLES = 40;  // one ELL step of a slice: 32 values + 32 16-bit positions = 320 B

struct TileHdr { int row_off, nrows, chunk_off, nchunks; };
struct ChunkHdr { int ucol_off, nu; int step_off[4]; int width[4]; };

struct TileMat {
  int ntiles; long nsteps, nnz, nucols; int nrows, bx, by, bz; long sy, sz;   // sy == 0: runs of consecutive rows
  TileHdr* d_th; ChunkHdr* d_ch; int* d_rows; int* d_ucols; double* d_steps;
  long nchunks;
};

// One workgroup (16 waves) per tile; wave w: slice w & 3, quarter w >> 2 of the slice's ELL steps.
// A tile is a sequence of STAGES (pass of 8 columns) x (chunk of the union); the X rows of stage s + 1 are requested
// into registers before stage s is computed from one LDS buffer and written into the other buffer afterwards, and the
// (value, position) steps are requested one group of four ahead — so neither the HBM latency of the staging nor that of
// the matrix stream is exposed, with ONE workgroup per CU (the two X buffers and the Y tile fill the LDS).
// LDS image: position p, column pair c (0..3) at p*4 + (c ^ ((p >> 2) & 1)) (16-byte units): the 16 lanes a
// ds_read_b128 serves together hold 8 rows x 2 halves and hit 16 different bank quads when the rows read neighbouring
// positions (stencil rows) or the same one (atom rows).
__global__ __launch_bounds__(1024) void spmm_tile_kernel(
    const TileHdr* __restrict__ th, const ChunkHdr* __restrict__ ch, const int* __restrict__ rows,
    const int* __restrict__ ucols, const double* __restrict__ steps, const double* __restrict__ x, size_t ldx,
    double* __restrict__ y, size_t ldy, int ncols, int ntiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  v2d* xt = reinterpret_cast<v2d*>(smem_raw);        // 2 buffers x TILE_CAP positions x 4 column pairs
  v2d* yt = xt + (size_t)2 * TILE_CAP * 4;           // 2 regions x TILE_ROWS rows x 4 column pairs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row32 = lane >> 1, h = lane & 1, slice = wave & 3, kq = wave >> 2;
  // blocks are dealt round-robin to the 8 XCDs: every XCD walks one contiguous eighth of the tiles, so bricks that
  // share halo rows run on the same L2 at about the same time
  const int per = (ntiles + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const TileHdr T = th[tile];
  const int su = tid >> 2, si = tid & 3;             // staging: position su + 256 q, 16-byte part si
  const int nch = T.nchunks, npass = (ncols + 7) >> 3, nst = npass * nch;
  const ChunkHdr* __restrict__ CH = ch + T.chunk_off;
  int col[5]; v2d sv[5];
  auto load_cols = [&](int chunk) {
    const int nu = CH[chunk].nu;
    const int* __restrict__ uc = ucols + CH[chunk].ucol_off;
#pragma unroll
    for (int q = 0; q < 5; ++q) col[q] = uc[min(su + 256 * q, nu - 1)];
  };
  auto load_x = [&](int pass) {
    const int c = 8 * pass + 2 * si;
    const double* __restrict__ xs = x + (c < ncols ? c : 0);
#pragma unroll
    for (int q = 0; q < 5; ++q) sv[q] = *reinterpret_cast<const v2d*>(xs + (size_t)col[q] * ldx);
  };
  auto store_x = [&](int chunk, int buf) {
    const int nu = CH[chunk].nu;
    v2d* dst = xt + (size_t)buf * TILE_CAP * 4;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int u = su + 256 * q;
      if (u < nu) dst[u * 4 + (si ^ ((u >> 2) & 1))] = sv[q];
    }
  };
  load_cols(0);
  load_x(0);
  store_x(0, 0);
  __syncthreads();
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  int pass = 0, chunk = 0;
  for (int s = 0; s < nst; ++s) {
    int npass_ = pass, nchunk = chunk + 1;
    if (nchunk == nch) { nchunk = 0; ++npass_; }
    const bool more = s + 1 < nst;
    if (more) {                       // block-uniform
      if (nch > 1) load_cols(nchunk);   // single-chunk tiles keep their column list in registers from pass to pass
      load_x(npass_);
    }
    // ---- compute stage s from buffer s & 1
    {
      const ChunkHdr* __restrict__ C = CH + chunk;
      const v2d* __restrict__ xb = xt + (size_t)(s & 1) * TILE_CAP * 4;
      const int W = C->width[slice];
      const int k0 = (W * kq) >> 2, k1 = (W * (kq + 1)) >> 2;
      if (k0 < k1) {                  // wave-uniform
        const double* __restrict__ sb = steps + (size_t)C->step_off[slice] * STEP_DOUBLES;
        double cv[4], nv[4]; unsigned cp[4], np[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double* sp = sb + (size_t)min(k0 + q, k1 - 1) * STEP_DOUBLES;
          cv[q] = sp[row32]; cp[q] = reinterpret_cast<const unsigned short*>(sp + 32)[row32];
        }
        for (int k = k0; k < k1; k += 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {   // the next group of four steps (clamped: the last group re-reads the last step)
            const double* sp = sb + (size_t)min(k + 4 + q, k1 - 1) * STEP_DOUBLES;
            nv[q] = sp[row32]; np[q] = reinterpret_cast<const unsigned short*>(sp + 32)[row32];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double v = (k + q < k1) ? cv[q] : 0.0;
            const v2d* xr = xb + cp[q] * 4;
            const unsigned sw = (cp[q] >> 2) & 1;
            const v2d x0 = xr[(2 * h) ^ sw], x1 = xr[(2 * h + 1) ^ sw];
            acc[0] = fma(v, x0.x, acc[0]); acc[1] = fma(v, x0.y, acc[1]);
            acc[2] = fma(v, x1.x, acc[2]); acc[3] = fma(v, x1.y, acc[3]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) { cv[q] = nv[q]; cp[q] = np[q]; }
        }
      }
    }
    if (more) store_x(nchunk, (s + 1) & 1);   // that buffer was last read in stage s - 1, which every wave has left
    if (chunk == nch - 1) {
      // end of a pass: the four quarters of a slice meet in the Y tile in a fixed order ((q0 + q2) + (q1 + q3)), then the
      // tile's rows leave as 64-byte segments
      const int r = slice * 32 + row32;
      v2d* y0 = yt + (size_t)r * 4 + 2 * h;
      v2d* y1 = y0 + (size_t)TILE_ROWS * 4;
      if (kq >= 2) { v2d* d = kq == 2 ? y0 : y1; d[0] = v2d{acc[0], acc[1]}; d[1] = v2d{acc[2], acc[3]}; }
      __syncthreads();
      if (kq < 2) {
        v2d* d = kq == 0 ? y0 : y1;
        const v2d a = d[0], b = d[1];
        acc[0] += a.x; acc[1] += a.y; acc[2] += b.x; acc[3] += b.y;
        if (kq == 1) { d[0] = v2d{acc[0], acc[1]}; d[1] = v2d{acc[2], acc[3]}; }
      }
      __syncthreads();
      if (kq == 0) {
        const v2d a = y1[0], b = y1[1];
        y0[0] = v2d{acc[0] + a.x, acc[1] + a.y}; y0[1] = v2d{acc[2] + b.x, acc[3] + b.y};
      }
      __syncthreads();
      if (tid < 512) {
        const int rr = tid >> 2, c = 8 * pass + 2 * si;
        if (rr < T.nrows && c < ncols)
          __builtin_nontemporal_store(yt[rr * 4 + si], reinterpret_cast<v2d*>(y + (size_t)rows[T.row_off + rr] * ldy + c));
      }
      acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
    }
    __syncthreads();
    pass = npass_; chunk = nchunk;
  }
}

// ---------------------------------------------------------------------------------------------- upload-time builder
struct TileOut {   // what one builder thread produced for its range of tiles
  std::vector<TileHdr> th; std::vector<ChunkHdr> ch; std::vector<int> rows, ucols; std::vector<double> steps;
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

static void build_range(int t0, int t1, const std::vector<int>& trow_off, const std::vector<int>& trows, int ncols_local,
                        const int* rowptr, const int* colidx, const double* val, TileOut* out) {
  std::vector<int> stamp((size_t)ncols_local, -1), lid((size_t)ncols_local, 0);
  std::vector<int> ulist, freq, order, chunk_of, pos_of;
  for (int t = t0; t < t1; ++t) {
    const int nr = trow_off[t + 1] - trow_off[t];
    // rows of the tile, longest first (stable: equal rows keep their natural order)
    std::vector<int> rw(trows.begin() + trow_off[t], trows.begin() + trow_off[t + 1]);
    std::stable_sort(rw.begin(), rw.end(), [&](int a, int b) { return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b]; });
    ulist.clear(); freq.clear();
    for (int r : rw)
      for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
        const int c = colidx[q];
        if (stamp[c] != t) { stamp[c] = t; lid[c] = (int)ulist.size(); ulist.push_back(c); freq.push_back(0); }
        ++freq[lid[c]];
      }
    const int nu = (int)ulist.size();
    const int nch = std::max(1, (nu + TILE_CAP - 1) / TILE_CAP);
    order.resize(nu);
    for (int i = 0; i < nu; ++i) order[i] = i;
    if (nch > 1) std::sort(order.begin(), order.end(), [&](int a, int b) { return freq[a] != freq[b] ? freq[a] > freq[b] : ulist[a] < ulist[b]; });
    chunk_of.assign(nu, 0); pos_of.assign(nu, 0);
    TileHdr T = {(int)out->rows.size(), nr, (int)out->ch.size(), nch};
    for (int r : rw) out->rows.push_back(r);
    for (int j = 0; j < nch; ++j) {   // chunk and position of every column of the union, before any row is written
      const int b = j * TILE_CAP, e = std::min(nu, b + TILE_CAP);
      std::sort(order.begin() + b, order.begin() + e, [&](int a, int c) { return ulist[a] < ulist[c]; });   // ascending columns inside a chunk
      for (int i = b; i < e; ++i) { chunk_of[order[i]] = j; pos_of[order[i]] = i - b; }
    }
    for (int j = 0; j < nch; ++j) {
      const int b = j * TILE_CAP, e = std::min(nu, b + TILE_CAP);
      ChunkHdr C; C.ucol_off = (int)out->ucols.size(); C.nu = e - b;
      for (int i = b; i < e; ++i) out->ucols.push_back(ulist[order[i]]);
      for (int s = 0; s < 4; ++s) {
        int W = 0;
        for (int rr = 32 * s; rr < std::min(nr, 32 * s + 32); ++rr) {
          int cnt = 0; const int r = rw[rr];
          for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) cnt += chunk_of[lid[colidx[q]]] == j;
          W = std::max(W, cnt);
        }
        C.width[s] = W; C.step_off[s] = (int)(out->steps.size() / STEP_DOUBLES);
        const size_t base = out->steps.size();
        out->steps.resize(base + (size_t)W * STEP_DOUBLES, 0.0);
        for (int rr = 32 * s; rr < std::min(nr, 32 * s + 32); ++rr) {
          const int r = rw[rr]; int k = 0;
          for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
            const int l = lid[colidx[q]];
            if (chunk_of[l] != j) continue;
            double* sp = out->steps.data() + base + (size_t)k * STEP_DOUBLES;
            sp[rr - 32 * s] = val[q];
            reinterpret_cast<unsigned short*>(sp + 32)[rr - 32 * s] = (unsigned short)pos_of[l];
            ++k;
          }
        }
      }
      out->ch.push_back(C);
    }
    out->th.push_back(T);
  }
}

}  // namespace gcge

using namespace gcge;

static int g_tile_mode = 0;   // 0 off, 1 every matrix without a pattern form, 2 every matrix (tests: next to a pattern form), -1 never
extern "C" void gcge_hip_spmm_tile_mode(int mode) { g_tile_mode = mode; }
extern "C" int gcge_hip_spmm_tile_mode_get(void) { return g_tile_mode; }

extern "C" void gcge_hip_tile_free(void* tm) {
  TileMat* T = (TileMat*)tm;
  if (!T) return;
  hipFree(T->d_th); hipFree(T->d_ch); hipFree(T->d_rows); hipFree(T->d_ucols); hipFree(T->d_steps);
  delete T;
}

// host side of the upload: tiles, chunks and ELL steps of rows [0, nrows) with LOCAL column indices in [0, ncols_local)
struct TileHost { std::vector<TileOut> outs; int ntiles = 0, bx = 0, by = 0, bz = 0; long sy = 0, sz = 0; };
static bool tile_build_host(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, TileHost* H) {
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
    for (long z0 = 0; z0 < nz; z0 += bz) for (long y0 = 0; y0 < ny; y0 += by) for (long x0 = 0; x0 < nx; x0 += bx) {
      const size_t before = trows.size();
      for (long z = z0; z < std::min(nz, z0 + bz); ++z) for (long yy = y0; yy < std::min(ny, y0 + by); ++yy)
        for (long xx = x0; xx < std::min(nx, x0 + bx); ++xx) { const long r = xx + sy * yy + sz * z; if (r < nrows) trows.push_back((int)r); }
      if (trows.size() > before) trow_off.push_back((int)before);
    }
  }
  const int ntiles = (int)trow_off.size();
  trow_off.push_back((int)trows.size());
  if ((long)trows.size() != nrows) return false;
  // tiles are independent: a contiguous range per thread, concatenated afterwards
  unsigned hw = std::thread::hardware_concurrency();
  int nth = (int)std::min<unsigned>(hw ? hw : 4, 16);
  if (const char* e = getenv("OMP_NUM_THREADS")) nth = std::max(1, std::min(nth, atoi(e)));
  nth = std::max(1, std::min(nth, ntiles / 64 + 1));
  H->outs.resize((size_t)nth);
  {
    std::vector<std::thread> pool;
    for (int i = 0; i < nth; ++i) {
      const int t0 = (int)((long)ntiles * i / nth), t1 = (int)((long)ntiles * (i + 1) / nth);
      pool.emplace_back(build_range, t0, t1, std::cref(trow_off), std::cref(trows), ncols_local, rowptr, colidx, val, &H->outs[i]);
    }
    for (auto& p : pool) p.join();
  }
  // offsets relative to a thread's own arrays -> global ones
  size_t o_ch = 0, o_rows = 0, o_uc = 0, o_st = 0;
  for (auto& o : H->outs) {
    for (auto& t : o.th) { t.row_off += (int)o_rows; t.chunk_off += (int)o_ch; }
    for (auto& c : o.ch) { c.ucol_off += (int)o_uc; for (int s = 0; s < 4; ++s) c.step_off[s] += (int)(o_st / STEP_DOUBLES); }
    o_ch += o.ch.size(); o_rows += o.rows.size(); o_uc += o.ucols.size(); o_st += o.steps.size();
  }
  if (o_st / STEP_DOUBLES > 2147483647UL || o_uc > 2147483647UL) return false;
  H->ntiles = ntiles; H->bx = bx; H->by = by; H->bz = bz; H->sy = grid ? sy : 0; H->sz = grid ? sz : 0;
  return true;
}

// Structural self-check of the upload (host only, no device needed; tests): expands tiles -> chunks -> ELL steps back into
// (row, column, value) triples and compares them with the CSR arrays, bit for bit.  0: identical; > 0: number of
// differences (missing, extra or altered entries, rows not covered exactly once, positions outside a chunk).
extern "C" long gcge_hip_tile_selfcheck(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val,
                                        double* xrows_per_row, double* ell_per_nnz, long* strides) {
  TileHost H;
  if (!tile_build_host(nrows, ncols_local, rowptr, colidx, val, &H)) return -1;
  std::vector<TileHdr> th; std::vector<ChunkHdr> ch; std::vector<int> rows, ucols; std::vector<double> steps;
  for (auto& o : H.outs) {
    th.insert(th.end(), o.th.begin(), o.th.end()); ch.insert(ch.end(), o.ch.begin(), o.ch.end());
    rows.insert(rows.end(), o.rows.begin(), o.rows.end()); ucols.insert(ucols.end(), o.ucols.begin(), o.ucols.end());
    steps.insert(steps.end(), o.steps.begin(), o.steps.end());
  }
  if (xrows_per_row) *xrows_per_row = (double)ucols.size() / nrows;
  if (ell_per_nnz) *ell_per_nnz = 32.0 * (double)(steps.size() / STEP_DOUBLES) / (double)rowptr[nrows];
  if (strides) { strides[0] = H.sy; strides[1] = H.sz; }
  long bad = 0;
  std::vector<int> seen((size_t)nrows, 0);
  std::vector<std::pair<int, double>> got;
  for (const TileHdr& T : th) {
    if (T.nrows > TILE_ROWS) ++bad;
    for (int rr = 0; rr < T.nrows; ++rr) {
      const int r = rows[(size_t)T.row_off + rr];
      if (r < 0 || r >= nrows) { ++bad; continue; }
      ++seen[r];
      got.clear();
      for (int j = 0; j < T.nchunks; ++j) {
        const ChunkHdr& C = ch[(size_t)T.chunk_off + j];
        if (C.nu > TILE_CAP) ++bad;
        const int s = rr / 32, q = rr % 32;
        for (int k = 0; k < C.width[s]; ++k) {
          const double* sp = steps.data() + ((size_t)C.step_off[s] + k) * STEP_DOUBLES;
          const int pos = reinterpret_cast<const unsigned short*>(sp + 32)[q];
          if (pos >= C.nu) { ++bad; continue; }
          uint64_t bits; memcpy(&bits, &sp[q], 8);
          if (bits == 0 && pos == 0) continue;      // padding (an explicit +0.0 entry at position 0 is indistinguishable and harmless)
          got.emplace_back(ucols[(size_t)C.ucol_off + pos], sp[q]);
        }
      }
      std::sort(got.begin(), got.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      std::vector<std::pair<int, double>> want;      // the row as stored (slabs: halo columns need not be ascending)
      for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) want.emplace_back(colidx[p], val[p]);
      std::sort(want.begin(), want.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      size_t g = 0;
      for (const auto& w : want) {
        uint64_t vb; memcpy(&vb, &w.second, 8);
        if (vb == 0) { if (g < got.size() && got[g].first == w.first && got[g].second == 0.0) ++g; continue; }
        if (g >= got.size() || got[g].first != w.first || memcmp(&got[g].second, &w.second, 8) != 0) { ++bad; continue; }
        ++g;
      }
      if (g != got.size()) ++bad;
    }
    // rows of a slice beyond the tile's last row must be padding
  }
  for (int r = 0; r < nrows; ++r) if (seen[r] != 1) ++bad;
  return bad;
}

// NULL: the matrix keeps the generic kernels (short rows, tiny matrices, more than 2^31 ELL steps).
extern "C" void* gcge_hip_tile_build(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val) {
  if (g_tile_mode <= 0 || nrows <= 0) return nullptr;   // measured slower than the pad-8 kernel so far (profiles/r03_spmm_generic): only on request
  const long nnz = rowptr[nrows];
  TileHost H;
  if (!tile_build_host(nrows, ncols_local, rowptr, colidx, val, &H)) return nullptr;
  size_t nth_ = 0, nch = 0, nuc = 0, nst = 0;
  for (auto& o : H.outs) { nth_ += o.th.size(); nch += o.ch.size(); nuc += o.ucols.size(); nst += o.steps.size(); }
  const int ntiles = H.ntiles;
  TileMat* T = new TileMat();
  T->ntiles = ntiles; T->nsteps = (long)(nst / STEP_DOUBLES); T->nnz = nnz; T->nucols = (long)nuc; T->nrows = nrows;
  T->bx = H.bx; T->by = H.by; T->bz = H.bz; T->sy = H.sy; T->sz = H.sz; T->nchunks = (long)nch;
  GCGE_HIP_CHECK(hipMalloc(&T->d_th, (size_t)ntiles * sizeof(TileHdr)));
  GCGE_HIP_CHECK(hipMalloc(&T->d_ch, nch * sizeof(ChunkHdr)));
  GCGE_HIP_CHECK(hipMalloc(&T->d_rows, (size_t)nrows * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&T->d_ucols, std::max<size_t>(nuc, 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&T->d_steps, std::max<size_t>(nst, 1) * sizeof(double)));
  size_t o_th = 0, o_ch = 0, o_rows = 0, o_uc = 0, o_st = 0;
  for (auto& o : H.outs) {
    if (!o.th.empty()) GCGE_HIP_CHECK(hipMemcpy(T->d_th + o_th, o.th.data(), o.th.size() * sizeof(TileHdr), hipMemcpyHostToDevice));
    if (!o.ch.empty()) GCGE_HIP_CHECK(hipMemcpy(T->d_ch + o_ch, o.ch.data(), o.ch.size() * sizeof(ChunkHdr), hipMemcpyHostToDevice));
    if (!o.rows.empty()) GCGE_HIP_CHECK(hipMemcpy(T->d_rows + o_rows, o.rows.data(), o.rows.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!o.ucols.empty()) GCGE_HIP_CHECK(hipMemcpy(T->d_ucols + o_uc, o.ucols.data(), o.ucols.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!o.steps.empty()) GCGE_HIP_CHECK(hipMemcpy(T->d_steps + o_st, o.steps.data(), o.steps.size() * sizeof(double), hipMemcpyHostToDevice));
    o_th += o.th.size(); o_ch += o.ch.size(); o_rows += o.rows.size(); o_uc += o.ucols.size(); o_st += o.steps.size();
    std::vector<double>().swap(o.steps);
  }
  return T;
}

// what the upload found: tiles, chunks, staged X rows per matrix row (the request factor), ELL entries per non-zero (padding)
extern "C" void gcge_hip_tile_stats(const void* tm, long* ntiles, long* nchunks, double* xrows_per_row, double* ell_per_nnz,
                                    int* brick, long* strides) {
  const TileMat* T = (const TileMat*)tm;
  if (ntiles) *ntiles = T->ntiles;
  if (nchunks) *nchunks = T->nchunks;
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
  const size_t lds = (size_t)(2 * TILE_CAP + 2 * TILE_ROWS) * 4 * sizeof(v2d);
  if (!attr_set) {
    GCGE_HIP_CHECK(hipFuncSetAttribute((const void*)spmm_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int per = (T->ntiles + 7) / 8;
  hipLaunchKernelGGL(spmm_tile_kernel, dim3((unsigned)(8 * per)), dim3(1024), lds, (hipStream_t)stream, T->d_th, T->d_ch, T->d_rows,
                     T->d_ucols, T->d_steps, d_x, (size_t)ldx, d_y, (size_t)ldy, ncols, T->ntiles);
  return (int)hipGetLastError();
}
