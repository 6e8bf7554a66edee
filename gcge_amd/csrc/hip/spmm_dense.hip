// K1 (supernode path) — dense row blocks of a sparse matrix on FP64 MFMA, the rest through the generic CSR kernels.
//
// Same contract as spmm.hip / spmm_pad8.hip (Y[:, 0:m) = A X[:, 0:m), reference app/app_ccs.c:50-139).  Matrices like
// BASELINE config 5's (real-space DFT Hamiltonians: a high-order stencil plus one dense block per atom — non-local
// projectors — test_eig_sol_SiO2_MAT.c of the reference) keep HALF of their non-zeros in 8 % of their rows: rows of
// 500-2000 entries that share their column set with a few hundred other rows.  The generic kernels gather one X row
// per non-zero for them (13 % of the HBM roofline, ten times the algorithmic bytes through L1).  Here such
// SUPERNODES — row sets R whose rows all reference (most of) one column set C — are found at upload and stored as
// dense |R| x |C| blocks in MFMA fragment order (8 B per entry, no indices); the product Y[R] += D X[C] then is a
// small dense contraction on v_mfma_f64_16x16x4_f64: one wave per 32 rows of a block holds 32 x 64 results in
// registers, an X row fetched for a k-step feeds 2 row fragments (32 rows), so the block's X rows cross L1 |R| / 32
// times instead of |R| times, and the values stream through once for all (up to) 64 columns of a pass.
// What does not belong to a block (the stencil part, rows near block borders) stays a CSR matrix — the REMAINDER —
// and takes the pad-8 kernel first (Y = A_rem X; under the plane sweep of spmm_star.hip: Y += over a LIST of the rows that have a
// remainder at all); no row lies in two blocks of one launch, so the launches that follow add into Y without atomics and the result is
// bit-reproducible.  Rows inside two overlapping atom balls get the second ball's entries from a second LAYER of blocks (a second
// launch, found by searching the remainder again): see gcge_hip_dense_build_rows.
//
// Detection (host, O(nnz of the long rows)): seeds = rows of >= min_len entries, longest first; the seed's columns
// are the candidate set C0; structural symmetry makes the rows whose INDEX lies in C0 the candidate rows; a
// candidate joins when at least half of its own entries and a quarter of C0 are shared; columns used by fewer
// than a quarter of the joined rows are dropped again; blocks below 16 rows, 32 columns or 35 % fill are not formed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <vector>
#include "gcge_hip_internal.h"

extern "C" int gcge_hip_pad8_spmm(int nrows, const int* d_orp, const int* d_pcol, const double* d_pval, const double* d_x,
                                  long ldx, double* d_y, long ldy, int ncols, void* stream);
extern "C" void gcge_hip_spmm_pad8_auto(double avg_octets_per_row);
extern "C" void gcge_hip_spmm_pad8_row_map(const int* d_map);
extern "C" void gcge_hip_spmm_pad8_row_map_add(const int* d_map);
extern "C" void* gcge_hip_tile_build_for(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, int remainder);
extern "C" void gcge_hip_tile_free(void* tm);
extern "C" int gcge_hip_tile_spmm(const void* tm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream);
extern "C" int gcge_hip_spmm_tile_mode_get(void);

namespace gcge {

typedef double v4d __attribute__((ext_vector_type(4)));

struct DenseSn { int row_off, nrows, col_off, ng; long val_off; };   // ng: groups of 4 columns (even), rows padded to 32 with -1
struct DenseItem { int sn, block; };

struct DenseMat {
  int nsn, nitems, nrows; long dense_entries, dense_nnz, rem_nnz;
  DenseSn* d_sn; DenseItem* d_items; int* d_rows; int* d_cols; double* d_vals;
  int* d_orp; int* d_pcol; double* d_pval; long noct;   // remainder, pad-8 form
  int* d_rowmap; int nlisted;                            // remainder given as a list of rows (gcge_hip_dense_build_rows); NULL: all rows
  int* d_padlist; int npad;
  int nlayers; int layer_item[9];                        // blocks of layer l: items [layer_item[l], layer_item[l + 1]); one launch per layer, in order                              // ... of them the rows whose remainder is NOT empty: what the pad-8 kernel walks (ADDING to Y)
  void* rem_tile;                                        // remainder in tile form (spmm_tile.hip) when that path is switched on
};

struct DenseHost {
  std::vector<DenseSn> sn; std::vector<DenseItem> items; std::vector<int> rows, cols; std::vector<double> vals;
  std::vector<int> rem_rowptr, rem_col; std::vector<double> rem_val;
  long dense_nnz = 0;
};

// One wave per 32 rows of a block and pass of up to 64 columns.  (Two consecutive 32-row blocks per wave — 16 MFMAs per X row
// fetched instead of 8 — was measured: 0.82 against 0.77 ms on the SiO2-like matrix; 216 VGPRs halve the waves per SIMD.)
// Every load is 16 bytes per lane: the texture-address path charges an 8-byte load like a 16-byte one (spmm_pad8.hip), and with
// 8-byte loads (two for the values, four for the X rows per column group: the round-4 form) eight waves asked a CU's L1 for 96 B per
// clock — the kernel ran at the address path's rate, 47 TF, not at the MFMA's.  So the fragments are laid out for PAIRS:
//   A fragment (values): lane l holds D[row 16 f + (l & 15)][column 4 g + (l >> 4)], f = 0, 1 — stored as that pair, one load
//   B fragment (X rows): lane l holds X[C[4 g + (l >> 4)]][c0 + 4 (l & 15) + cf], cf = 0 .. 3 — 32 contiguous bytes, two loads;
//                        the 16 lanes of a row cover its 512 bytes (MFMA column (l & 15) of fragment cf IS result column 4 (l & 15) + cf)
//   accumulators:        lane l, register t of tile (f, cf): row 16 f + 4 t + (l >> 4), column c0 + 4 (l & 15) + cf — 32 contiguous bytes again
//   FULL: m is a multiple of 64 (no column tests).
typedef double v2dd __attribute__((ext_vector_type(2)));
template <bool FULL>
__global__ __launch_bounds__(256, FULL ? 3 : 2) void spmm_dense_kernel(
    const DenseItem* __restrict__ items, int nitems, const DenseSn* __restrict__ sns, const int* __restrict__ rows,
    const int* __restrict__ cols, const double* __restrict__ vals, const double* __restrict__ x, size_t ldx,
    double* __restrict__ y, size_t ldy, int m) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x * 4 + wave;
  if (item >= nitems) return;
  const DenseItem it = items[item];
  const DenseSn S = sns[it.sn];
  const int li = lane & 15, kk = lane >> 4;
  const int ng = S.ng;
  const v2dd* __restrict__ vp = reinterpret_cast<const v2dd*>(vals + S.val_off + (size_t)it.block * ng * 128) + lane;
  const int* __restrict__ cp = cols + S.col_off + kk;
  const int* __restrict__ rp = rows + S.row_off + 32 * it.block;
  for (int c0 = 0; c0 < m; c0 += 64) {
    // my four columns c0 + 4 li .. + 3 as two pairs; a pair past the end (m is even: a pair is in or out) re-reads column 0, never stored
    const bool in0 = FULL || c0 + 4 * li < m, in1 = FULL || c0 + 4 * li + 2 < m;
    const double* __restrict__ xb0 = x + (in0 ? c0 + 4 * li : 0);
    const double* __restrict__ xb1 = x + (in1 ? c0 + 4 * li + 2 : 0);
    v4d acc[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int cf = 0; cf < 4; ++cf) acc[f][cf] = v4d{0.0, 0.0, 0.0, 0.0};
    // FOUR register sets, the loads three column groups ahead of the MFMAs that use them: the block values stream from HBM
    // (read once, 1.7 GB on the SiO2-like matrix) and only two waves fit a SIMD (64 accumulator registers).  ng is even (padded at
    // upload); groups past the end are clamped re-reads whose MFMAs are skipped.
    v2dd a[4], b[4][2];
    auto load = [&](v2dd& av, v2dd (&bv)[2], int g, int col) {
      av = vp[(size_t)g * 64];
      bv[0] = *reinterpret_cast<const v2dd*>(xb0 + (size_t)col * ldx);
      bv[1] = *reinterpret_cast<const v2dd*>(xb1 + (size_t)col * ldx);
    };
    auto mfma = [&](const v2dd& av, const v2dd (&bv)[2]) {
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv[0].x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv[0].y, acc[0][1], 0, 0, 0);
      acc[0][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv[1].x, acc[0][2], 0, 0, 0);
      acc[0][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv[1].y, acc[0][3], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv[0].x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv[0].y, acc[1][1], 0, 0, 0);
      acc[1][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv[1].x, acc[1][2], 0, 0, 0);
      acc[1][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv[1].y, acc[1][3], 0, 0, 0);
    };
    auto colof = [&](int g) { return cp[4 * min(g, ng - 1)]; };
    // column indices a whole round (four groups) ahead of the gathers that use them
    load(a[0], b[0], 0, cp[0]);
    load(a[1], b[1], min(1, ng - 1), colof(1));
    load(a[2], b[2], min(2, ng - 1), colof(2));
    int cA = colof(3), cB = colof(4), cC = colof(5), cD = colof(6);
    for (int g = 0; g < ng; g += 4) {
      const int nA = colof(g + 7), nB = colof(g + 8), nC = colof(g + 9), nD = colof(g + 10);
      load(a[3], b[3], min(g + 3, ng - 1), cA);
      __builtin_amdgcn_sched_barrier(0);
      mfma(a[0], b[0]);
      __builtin_amdgcn_sched_barrier(0);
      load(a[0], b[0], min(g + 4, ng - 1), cB);
      __builtin_amdgcn_sched_barrier(0);
      mfma(a[1], b[1]);
      __builtin_amdgcn_sched_barrier(0);
      load(a[1], b[1], min(g + 5, ng - 1), cC);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 2 < ng) mfma(a[2], b[2]);
      __builtin_amdgcn_sched_barrier(0);
      load(a[2], b[2], min(g + 6, ng - 1), cD);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 2 < ng) mfma(a[3], b[3]);
      __builtin_amdgcn_sched_barrier(0);
      cA = nA; cB = nB; cC = nC; cD = nD;
    }
    // Y[R] += block: every row belongs to one block of this launch only and what wrote Y before has finished (stream order)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = rp[16 * f + 4 * t + kk];
        if (r < 0) continue;
        double* q = y + (size_t)r * ldy + c0 + 4 * li;
        if (in0) { v2dd* q0 = reinterpret_cast<v2dd*>(q); v2dd o = *q0; o.x += acc[f][0][t]; o.y += acc[f][1][t]; *q0 = o; }
        if (in1) { v2dd* q1 = reinterpret_cast<v2dd*>(q + 2); v2dd o = *q1; o.x += acc[f][2][t]; o.y += acc[f][3][t]; *q1 = o; }
      }
  }
}

// ------------------------------------------------------------------------------------------------ detection (host)
static bool dense_build_host(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, int min_len,
                             DenseHost* H, int min_rows = 16) {
  std::vector<int> seeds;
  for (int r = 0; r < nrows; ++r) if (rowptr[r + 1] - rowptr[r] >= min_len) seeds.push_back(r);
  if (seeds.empty()) return false;
  std::stable_sort(seeds.begin(), seeds.end(), [&](int a, int b) { return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b]; });
  std::vector<int> assigned((size_t)nrows, -1), stamp((size_t)ncols_local, -1), pos((size_t)ncols_local, 0), cnt;
  std::vector<int> cand, R, C;
  int id = 0;
  std::vector<int> C0;
  for (int r0 : seeds) {
    if (assigned[r0] != -1) continue;
    // the candidate column set: the seed's columns AND its own index — the rows a sweep leaves come without their diagonal
    // (spmm_star.hip takes it), so column r0 is missing from row r0 alone; without it every other row of the block kept ONE entry
    // outside (a listed row per block row: 3.8e5 rows of 2 entries on the SiO2-like matrix)
    C0.assign(colidx + rowptr[r0], colidx + rowptr[r0 + 1]);
    // (a linear search: the columns of a slab's rows are ascending by GLOBAL index, not by the local one)
    if (r0 < ncols_local && std::find(C0.begin(), C0.end(), r0) == C0.end()) C0.push_back(r0);
    const int n0 = (int)C0.size();
    ++id;
    for (int c : C0) stamp[c] = id;
    // candidate rows: the rows whose index is a column of the seed (structural symmetry), not yet in a block
    R.clear();
    for (int r : C0) {
      if (r >= nrows || assigned[r] != -1) continue;
      const int len = rowptr[r + 1] - rowptr[r];
      if (2 * len < min_len) continue;
      int ov = 0;
      for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) ov += stamp[colidx[p]] == id;
      if (2 * ov >= len && 4 * ov >= n0) R.push_back(r);
    }
    if ((int)R.size() < min_rows) { assigned[r0] = -2; continue; }
    // columns of the seed that at least a quarter of the joined rows use
    C.clear(); cnt.assign((size_t)n0, 0);
    for (int q = 0; q < n0; ++q) pos[C0[q]] = q;
    for (int r : R) for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) if (stamp[colidx[p]] == id) ++cnt[pos[colidx[p]]];
    long filled = 0;
    for (int q = 0; q < n0; ++q)
      if (4L * cnt[q] >= (long)R.size()) { C.push_back(C0[q]); filled += cnt[q]; }
      else stamp[C0[q]] = -1;   // dropped
    if ((int)C.size() < 32 || (double)filled < 0.35 * (double)R.size() * (double)C.size()) { assigned[r0] = -2; continue; }
    std::sort(R.begin(), R.end()); std::sort(C.begin(), C.end());
    const int sn = (int)H->sn.size();
    const int nb = ((int)R.size() + 31) / 32;
    int ng = ((int)C.size() + 3) / 4; ng += ng & 1;
    DenseSn S = {(int)H->rows.size(), (int)R.size(), (int)H->cols.size(), ng, (long)H->vals.size()};
    for (int i = 0; i < 32 * nb; ++i) H->rows.push_back(i < (int)R.size() ? R[i] : -1);
    for (int k = 0; k < 4 * ng; ++k) H->cols.push_back(k < (int)C.size() ? C[k] : C[0]);   // padding: a valid column, zero values
    for (size_t k = 0; k < C.size(); ++k) pos[C[k]] = (int)k;
    H->vals.resize(H->vals.size() + (size_t)nb * ng * 128, 0.0);
    double* D = H->vals.data() + S.val_off;
    for (size_t ir = 0; ir < R.size(); ++ir) {
      const int r = R[ir]; assigned[r] = sn;
      const size_t b = ir / 32, f = (ir % 32) / 16, i = ir % 16;
      for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) {
        const int c = colidx[p];
        if (stamp[c] != id) continue;
        const size_t k = (size_t)pos[c], g = k / 4, kq = k % 4;
        D[((b * ng + g) * 64 + kq * 16 + i) * 2 + f] = val[p];
        ++H->dense_nnz;
      }
    }
    // the remainder of these rows is written below, against the same stamps: keep them alive under a per-block id
    H->sn.push_back(S);
    for (int b = 0; b < nb; ++b) H->items.push_back(DenseItem{sn, b});
  }
  if (H->sn.empty()) return false;
  // remainder: every entry that is not inside its row's block
  H->rem_rowptr.assign((size_t)nrows + 1, 0);
  std::fill(stamp.begin(), stamp.end(), -1);
  std::vector<std::vector<int>> members(H->sn.size());
  for (int r = 0; r < nrows; ++r) if (assigned[r] >= 0) members[assigned[r]].push_back(r);
  std::vector<char> in_block((size_t)rowptr[nrows], 0);
  for (size_t s = 0; s < H->sn.size(); ++s) {
    const DenseSn& S = H->sn[s];
    for (int k = 0; k < 4 * S.ng; ++k) stamp[H->cols[(size_t)S.col_off + k]] = (int)s;
    for (int r : members[s]) for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) if (stamp[colidx[p]] == (int)s) in_block[p] = 1;
  }
  for (int r = 0; r < nrows; ++r) {
    int c = 0;
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) c += !in_block[p];
    H->rem_rowptr[r + 1] = H->rem_rowptr[r] + c;
  }
  H->rem_col.resize((size_t)H->rem_rowptr[nrows]); H->rem_val.resize((size_t)H->rem_rowptr[nrows]);
  for (int r = 0; r < nrows; ++r) {
    int o = H->rem_rowptr[r];
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) if (!in_block[p]) { H->rem_col[o] = colidx[p]; H->rem_val[o] = val[p]; ++o; }
  }
  return true;
}

}  // namespace gcge

using namespace gcge;

static int g_dense_mode = 0;     // 0 automatic, 1 rows of >= 24 entries may seed a block (tests), -1 never
static int g_dense_layers = 4;      // launches of the block kernel at most (1: the round-4 form, every row in one block only)
static int g_dense_layer_len = 32;  // rows of at least this many entries seed a block of a later layer
extern "C" void gcge_hip_spmm_dense_layers(int layers, int seed_len) { g_dense_layers = layers >= 1 ? layers : 4; g_dense_layer_len = seed_len >= 24 ? seed_len : 32; }
static int g_dense_min_len = 96;
extern "C" void gcge_hip_spmm_dense_min_len(int len) { g_dense_min_len = len >= 24 ? len : 96; }   // rows of at least this many entries may seed a block
extern "C" void gcge_hip_spmm_dense_mode(int mode) { g_dense_mode = mode; }
extern "C" int gcge_hip_spmm_dense_mode_get(void) { return g_dense_mode; }

// Host-only structural self-check of the split (tests; no device needed): blocks + remainder, expanded back into (row,
// column, value) triples, equal the CSR arrays bit for bit and no row lies in two blocks.  Returns the number of
// differences (0 = identical), -1 when no block was found; also the number of blocks, the share of the non-zeros they
// hold and their fill.
extern "C" long gcge_hip_dense_selfcheck(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val,
                                         int min_len, long* nblocks, double* share, double* fill) {
  DenseHost H;
  if (!dense_build_host(nrows, ncols_local, rowptr, colidx, val, min_len, &H)) return -1;
  long bad = 0, entries = 0;
  std::vector<std::vector<std::pair<int, double>>> got((size_t)nrows);
  std::vector<int> owner((size_t)nrows, -1);
  for (size_t s = 0; s < H.sn.size(); ++s) {
    const DenseSn& S = H.sn[s];
    const int nb = (S.nrows + 31) / 32;
    entries += (long)S.nrows * 4 * S.ng;
    for (int ir = 0; ir < 32 * nb; ++ir) {
      const int r = H.rows[(size_t)S.row_off + ir];
      if (ir >= S.nrows) { if (r != -1) ++bad; continue; }
      if (r < 0 || r >= nrows) { ++bad; continue; }
      if (owner[r] != -1) ++bad;
      owner[r] = (int)s;
      const size_t b = ir / 32, f = (ir % 32) / 16, i = ir % 16;
      for (int k = 0; k < 4 * S.ng; ++k) {
        const double v = H.vals[(size_t)S.val_off + ((b * S.ng + k / 4) * 64 + (k % 4) * 16 + i) * 2 + f];
        uint64_t bits; memcpy(&bits, &v, 8);
        if (bits != 0) got[r].emplace_back(H.cols[(size_t)S.col_off + k], v);
      }
    }
  }
  for (int r = 0; r < nrows; ++r) {
    for (int p = H.rem_rowptr[r]; p < H.rem_rowptr[r + 1]; ++p) got[r].emplace_back(H.rem_col[p], H.rem_val[p]);
    std::vector<std::pair<int, double>> want;
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) want.emplace_back(colidx[p], val[p]);
    auto less = [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; };
    std::sort(got[r].begin(), got[r].end(), less); std::sort(want.begin(), want.end(), less);
    // explicit zeros of the CSR arrays that fell inside a block are indistinguishable from the block's padding: skip them
    size_t g = 0;
    for (const auto& w : want) {
      if (g < got[r].size() && got[r][g].first == w.first) { if (memcmp(&got[r][g].second, &w.second, 8) != 0) ++bad; ++g; }
      else if (w.second != 0.0) ++bad;
    }
    if (g != got[r].size()) ++bad;
  }
  if (nblocks) *nblocks = (long)H.sn.size();
  if (share) *share = (double)H.dense_nnz / (double)rowptr[nrows];
  if (fill) *fill = entries ? (double)H.dense_nnz / (double)entries : 0.0;
  return bad;
}

extern "C" void gcge_hip_dense_free(void* dm) {
  DenseMat* D = (DenseMat*)dm;
  if (!D) return;
  hipFree(D->d_sn); hipFree(D->d_items); hipFree(D->d_rows); hipFree(D->d_cols); hipFree(D->d_vals);
  hipFree(D->d_orp); hipFree(D->d_pcol); hipFree(D->d_pval);
  if (D->d_rowmap) hipFree(D->d_rowmap);
  if (D->d_padlist) hipFree(D->d_padlist);
  if (D->rem_tile) gcge_hip_tile_free(D->rem_tile);
  delete D;
}

template <class T>
static T* to_device(const std::vector<T>& v) {
  T* d = nullptr;
  GCGE_HIP_CHECK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) GCGE_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

// NULL: no block worth forming (the matrix keeps the generic kernels alone)
extern "C" void* gcge_hip_dense_build_rows(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, const unsigned char* not_listed);
extern "C" void* gcge_hip_dense_build(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val) {
  return gcge_hip_dense_build_rows(nrows, ncols_local, rowptr, colidx, val, nullptr);
}
// not_listed != NULL: rows with not_listed[r] != 0 are EMPTY in these arrays and belong to another kernel (spmm_star.hip): the
// pad-8 part of the remainder then walks a list of the other rows only and leaves those rows of Y alone
extern "C" void* gcge_hip_dense_build_rows(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, const unsigned char* not_listed) {
  if (g_dense_mode < 0 || nrows <= 0) return nullptr;
  if (const char* e = getenv("GCGE_DENSE_LAYERS")) g_dense_layers = atoi(e) >= 1 ? atoi(e) : g_dense_layers;   // (measurements / bisection)
  const int min_len = g_dense_mode == 1 ? 24 : g_dense_min_len;
  DenseHost H;
  const long nnz = rowptr[nrows];
  bool blocks = dense_build_host(nrows, ncols_local, rowptr, colidx, val, min_len, &H);
  if (blocks && not_listed == nullptr && g_dense_mode == 0 && 10 * H.dense_nnz < nnz) return nullptr;   // blocks hold less than a tenth of the matrix: not worth a second launch
  if (!blocks) {
    if (not_listed == nullptr) return nullptr;
    // the rows a sweep leaves, and no block among them (a star with a few irregular rows, or none): all of it is the listed remainder
    H = DenseHost();
    H.rem_rowptr.assign(rowptr, rowptr + nrows + 1); H.rem_col.assign(colidx, colidx + nnz); H.rem_val.assign(val, val + nnz);
  }
  // LAYERS.  No row lies in two blocks of one launch (that is what makes Y[R] += block safe without atomics), so a row inside two
  // overlapping atom balls leaves the entries of the second ball to the remainder: a few rows with hundreds of entries each, and the
  // pad-8 list walks a row with ONE wave — on the SiO2-like matrix of config 5 the list's 4.9e6 non-zeros took 0.43 ms, all of it the
  // tail of its longest rows.  The remainder is therefore searched again (shorter seeds, smaller row sets) and what it yields becomes a
  // second, third ... launch of the block kernel, ordered behind the first by the stream: one owner per row and launch, still bit-reproducible.
  std::vector<int> layer_item = {0, (int)H.items.size()};
  // (only under the sweep: there the remainder holds nothing but what differs from the star; on a whole matrix every stencil row of
  //  >= 32 entries would be tried as a seed again in every layer — 20 s of upload at 2·10⁶ rows for nothing)
  if (blocks && g_dense_layers > 1 && not_listed != nullptr) {
    for (int l = 1; l < g_dense_layers && l < 8; ++l) {
      DenseHost H2;
      if (!dense_build_host(nrows, ncols_local, H.rem_rowptr.data(), H.rem_col.data(), H.rem_val.data(), g_dense_layer_len, &H2, 8)) break;
      const int sn0 = (int)H.sn.size(), row0 = (int)H.rows.size(), col0 = (int)H.cols.size(); const long val0 = (long)H.vals.size();
      for (DenseSn S : H2.sn) { S.row_off += row0; S.col_off += col0; S.val_off += val0; H.sn.push_back(S); }
      for (DenseItem it : H2.items) { it.sn += sn0; H.items.push_back(it); }
      H.rows.insert(H.rows.end(), H2.rows.begin(), H2.rows.end());
      H.cols.insert(H.cols.end(), H2.cols.begin(), H2.cols.end());
      H.vals.insert(H.vals.end(), H2.vals.begin(), H2.vals.end());
      H.dense_nnz += H2.dense_nnz;
      H.rem_rowptr.swap(H2.rem_rowptr); H.rem_col.swap(H2.rem_col); H.rem_val.swap(H2.rem_val);
      layer_item.push_back((int)H.items.size());
    }
  }
  DenseMat* D = new DenseMat();
  D->nlayers = (int)layer_item.size() - 1;
  for (int l = 0; l <= D->nlayers; ++l) D->layer_item[l] = layer_item[l];
  D->nsn = (int)H.sn.size(); D->nitems = (int)H.items.size(); D->nrows = nrows;
  D->dense_entries = (long)H.vals.size(); D->dense_nnz = H.dense_nnz; D->rem_nnz = H.rem_rowptr[nrows];
  D->d_sn = to_device(H.sn); D->d_items = to_device(H.items); D->d_rows = to_device(H.rows); D->d_cols = to_device(H.cols);
  D->d_vals = to_device(H.vals);
  std::vector<double>().swap(H.vals);
  // remainder in pad-8 form: every row padded to a multiple of 8 entries with (own column, 0.0)
  // With a list the remainder is ADDED to what the sweep wrote: rows whose remainder is empty (every entry inside their block: most
  // rows of a block) are not walked at all — each would cost a read and a write of its Y row for nothing; `all` (every row that is
  // not `not_listed`) stays the list of the column sums over those rows (gcge_hip_dense_row_list).
  std::vector<int> list, all;
  for (int r = 0; r < nrows; ++r)
    if (not_listed == nullptr) list.push_back(r);
    else if (!not_listed[r]) { all.push_back(r); if (H.rem_rowptr[r + 1] > H.rem_rowptr[r]) list.push_back(r); }
  const int nl = (int)list.size();
  std::vector<int> orp((size_t)nl + 1);
  size_t noct = 0;
  for (int i = 0; i < nl; ++i) { const int r = list[i]; orp[i] = (int)noct; noct += ((size_t)(H.rem_rowptr[r + 1] - H.rem_rowptr[r]) + 7) / 8; }
  orp[nl] = (int)noct;
  std::vector<int> pc(noct * 8); std::vector<double> pv(noct * 8);
  for (int i = 0; i < nl; ++i) {
    const int r = list[i];
    size_t o = (size_t)orp[i] * 8;
    for (int k = H.rem_rowptr[r]; k < H.rem_rowptr[r + 1]; ++k, ++o) { pc[o] = H.rem_col[k]; pv[o] = H.rem_val[k]; }
    for (; o < (size_t)orp[i + 1] * 8; ++o) { pc[o] = r; pv[o] = 0.0; }
  }
  D->noct = (long)noct;
  D->d_orp = to_device(orp); D->d_pcol = to_device(pc); D->d_pval = to_device(pv);
  D->nlisted = not_listed != nullptr ? (int)all.size() : nl;
  D->d_rowmap = not_listed != nullptr ? to_device(all) : nullptr;
  D->npad = nl;
  D->d_padlist = not_listed != nullptr ? to_device(list) : nullptr;
  // (a listed remainder stays with the pad-8 kernel: a tile writes all of its rows)
  D->rem_tile = not_listed != nullptr ? nullptr : gcge_hip_tile_build_for(nrows, ncols_local, H.rem_rowptr.data(), H.rem_col.data(), H.rem_val.data(), 1);   // NULL: pad-8
  return D;
}

extern "C" const int* gcge_hip_dense_row_list(const void* dm, int* nlisted) {   // the rows its pad-8 part walks (NULL: all rows)
  const DenseMat* D = (const DenseMat*)dm;
  if (nlisted) *nlisted = D->d_rowmap != nullptr ? D->nlisted : 0;
  return D->d_rowmap;
}
extern "C" int gcge_hip_dense_remainder_is_tiled(const void* dm) { return ((const DenseMat*)dm)->rem_tile != nullptr; }
extern "C" const void* gcge_hip_dense_remainder_tile(const void* dm) { return ((const DenseMat*)dm)->rem_tile; }
extern "C" void gcge_hip_dense_stats(const void* dm, long* nblocks, long* items, long* dense_nnz, long* dense_entries, long* rem_nnz) {
  const DenseMat* D = (const DenseMat*)dm;
  if (nblocks) *nblocks = D->nsn;
  if (items) *items = D->nitems;
  if (dense_nnz) *dense_nnz = D->dense_nnz;
  if (dense_entries) *dense_entries = D->dense_entries;
  if (rem_nnz) *rem_nnz = D->rem_nnz;
}

// Y[:, 0:ncols) = A X[:, 0:ncols): remainder through the pad-8 kernel, then the blocks.  -1: alignment contract of the
// pad-8 kernel not met (the caller keeps the CSR kernel on the full matrix).  which: 0 both, 1 remainder only, 2 blocks only (measurements);
// + 4: a LISTED remainder is ADDED to the rows of Y (the sweep of spmm_star.hip wrote star + diagonal there before)
extern "C" int gcge_hip_dense_spmm(const void* dm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream, int which) {
  const DenseMat* D = (const DenseMat*)dm;
  if (ncols <= 0) return 0;
  if ((ncols & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)d_x & 15) || ((uintptr_t)d_y & 15) || d_x == d_y) return -1;
  const bool add = (which & 4) != 0 && D->d_rowmap != nullptr;
  which &= 3;
  if (!add && D->d_rowmap != nullptr && which != 2) return -1;      // a listed remainder only ever adds (rows without a remainder are not walked)
  if (which != 2 && D->rem_tile != nullptr) {
    const int rc = gcge_hip_tile_spmm(D->rem_tile, d_x, ldx, d_y, ldy, ncols, stream);
    if (rc != 0) return rc;
  } else if (which != 2 && D->npad > 0) {
    gcge_hip_spmm_pad8_auto((double)D->noct / D->npad);
    if (add) gcge_hip_spmm_pad8_row_map_add(D->d_padlist); else gcge_hip_spmm_pad8_row_map(nullptr);
    const int rc = gcge_hip_pad8_spmm(D->npad, D->d_orp, D->d_pcol, D->d_pval, d_x, ldx, d_y, ldy, ncols, stream);
    gcge_hip_spmm_pad8_row_map(nullptr);
    if (rc != 0) return rc;
  }
  if (which != 1)
    for (int l = 0; l < D->nlayers; ++l) {
      const int i0 = D->layer_item[l], ni = D->layer_item[l + 1] - i0;
      if (ni <= 0) continue;
      hipLaunchKernelGGL((ncols % 64 == 0 ? spmm_dense_kernel<true> : spmm_dense_kernel<false>), dim3((unsigned)((ni + 3) / 4)), dim3(256), 0, (hipStream_t)stream, D->d_items + i0, ni,
                         D->d_sn, D->d_rows, D->d_cols, D->d_vals, d_x, (size_t)ldx, d_y, (size_t)ldy, ncols);
    }
  return (int)hipGetLastError();
}
