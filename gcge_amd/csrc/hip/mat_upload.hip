// Matrix handle of the HIP back-end (the CCSMAT counterpart, reference app/app_ccs.h:20-24): upload of the host CSR arrays,
// their analysis into the forms the K1 kernels take (pad-8 copy, row patterns, star + remainder, dense blocks, tiles), the halo
// plan of a row slab, destruction.  Split off app_hip.hip in round 4; the slots live there.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <map>
#include <unordered_map>
#include <vector>

#include "gcge_hip.h"
#include "gcge_hip_internal.h"

extern "C" {
int gcge_hip_spmm_path_get(void);
int gcge_hip_offset_patterns_get(void);
int gcge_hip_pattern_width(int max_row_len);
void* gcge_hip_tile_build(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val);
void gcge_hip_tile_free(void* tm);
int gcge_hip_spmm_tile_mode_get(void);
void* gcge_hip_dense_build(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val);
void gcge_hip_dense_free(void* dm);
int gcge_hip_dense_remainder_is_tiled(const void* dm);
const void* gcge_hip_dense_remainder_tile(const void* dm);
void gcge_hip_dense_stats(const void* dm, long* nblocks, long* items, long* dense_nnz, long* dense_entries, long* rem_nnz);
void gcge_hip_tile_stats(const void* tm, long* ntiles, long* ov_nnz, double* xrows_per_row, double* ell_per_nnz, int* brick, long* strides);
void* gcge_hip_star_build(int nrows, int ncols_local, long row_begin, long nglobal, const int* ghost, const int* rowptr, const int* colidx, const double* val,
                          const int** rem_rowptr, const int** rem_col, const double** rem_val);
void gcge_hip_star_release_remainder(void);
void gcge_hip_star_free(void* sm);
void gcge_hip_star_stats(const void* sm, long* out);
const unsigned char* gcge_hip_star_host_mask(void);
void* gcge_hip_dense_build_rows(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, const unsigned char* not_listed);
}
#define g_spmm_path (gcge_hip_spmm_path_get())
#define g_offset_patterns (gcge_hip_offset_patterns_get())

// ------------------------------------------------------------------ matrix
// Pattern format: rows written as {(column - row, value)} in CSR order; at most 64 KB of table.  Leaves
// A->d_pid == NULL when the matrix has too many distinct rows (irregular matrices give up after a few
// hundred rows, so the scan costs nothing there).
struct PatEntryH { double val; long off; };
// by_offsets: rows are compared by their column offsets only; the table then carries 1.0 for every present entry and the
// values travel per row (A->d_rowval: 8 doubles per row in table-slot order, tables of at most 8 slots): stencils with
// variable coefficients keep the pattern kernels at 64 more bytes per row and 16-column pass.
static void build_patterns(GCGE_HIP_MAT* A, int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val,
                           bool by_offsets = false) {
  A->d_pid = nullptr; A->d_tab = nullptr; A->npat = 0; A->pat_lt = 0; A->pat_near = 0; A->d_rowval = nullptr;
  int maxlen = 0;
  for (int r = 0; r < nrows; ++r) maxlen = std::max(maxlen, rowptr[r + 1] - rowptr[r]);
  const int lt = gcge_hip_pattern_width(maxlen);
  if (lt == 0 || nrows == 0 || (by_offsets && lt > 8)) return;
  const int maxpat = std::min(65535, (int)(64 * 1024 / (lt * sizeof(PatEntryH))));
  std::vector<PatEntryH> tab;
  std::unordered_map<uint64_t, std::vector<int>> byhash;
  std::vector<unsigned short> pid((size_t)nrows);
  auto same = [&](int p, int r) {
    const PatEntryH* e = &tab[(size_t)p * lt];
    const int len = rowptr[r + 1] - rowptr[r];
    for (int k = 0; k < len; ++k) {
      const int q = rowptr[r] + k;
      if (e[k].off != (long)colidx[q] - r || (!by_offsets && memcmp(&e[k].val, &val[q], sizeof(double)) != 0)) return false;
      if (by_offsets && e[k].val == 0.0) return false;   // (a padding slot: the table row is shorter)
    }
    for (int k = len; k < lt; ++k) if (e[k].off != 0 || e[k].val != 0.0) return false;
    return true;
  };
  int prev = -1;
  for (int r = 0; r < nrows; ++r) {
    if (prev >= 0 && same(prev, r)) { pid[r] = (unsigned short)prev; continue; }
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)(rowptr[r + 1] - rowptr[r]);
    for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
      uint64_t vb; memcpy(&vb, &val[q], 8);
      if (by_offsets) vb = 1;
      h = (h ^ (uint64_t)((long)colidx[q] - r)) * 0xBF58476D1CE4E5B9ull; h ^= h >> 29;
      h = (h ^ vb) * 0x94D049BB133111EBull; h ^= h >> 32;
    }
    int found = -1;
    std::vector<int>& cand = byhash[h];
    for (int p : cand) if (same(p, r)) { found = p; break; }
    if (found < 0) {
      if ((int)(tab.size() / lt) >= maxpat) return;   // not a pattern matrix
      found = (int)(tab.size() / lt);
      for (int k = 0; k < lt; ++k) {
        const int q = rowptr[r] + k;
        PatEntryH e = {0.0, 0};
        if (q < rowptr[r + 1]) { e.val = by_offsets ? 1.0 : val[q]; e.off = (long)colidx[q] - r; }
        tab.push_back(e);
      }
      cand.push_back(found);
    }
    pid[r] = (unsigned short)found; prev = found;
  }
  A->npat = (int)(tab.size() / lt); A->pat_lt = lt; A->pat_span = 0;
  // reuse distance that matters for the launch geometry: the longest offset of the MOST FREQUENT pattern
  // (interior rows); boundary and halo patterns may reach much further
  std::vector<long> freq((size_t)A->npat, 0);
  for (int r = 0; r < nrows; ++r) ++freq[pid[r]];
  const int common = (int)(std::max_element(freq.begin(), freq.end()) - freq.begin());
  A->pat_span2 = 0;
  for (int k = 0; k < lt; ++k) {
    const long o = tab[(size_t)common * lt + k].off;
    A->pat_span = std::max(A->pat_span, o < 0 ? -o : o);
  }
  for (int k = 0; k < lt; ++k) {
    const long o = tab[(size_t)common * lt + k].off, ao = o < 0 ? -o : o;
    if (ao < A->pat_span) A->pat_span2 = std::max(A->pat_span2, ao);
  }
  // Chain layout (spmm_pattern_chain_kernel): possible when the interior stencil reaches -S, 0 and +S with S a
  // multiple of 32 rows and all patterns together use at most lt distinct offsets.  Every pattern is then rewritten
  // on the same slots [-S, 0, +S, the other offsets ascending]: entries a row does not have get value 0 but keep
  // their offset as long as the address stays inside the block of vectors (patterns are split by that validity),
  // so what a lane loads through a slot depends on its position only, never on its pattern.
  do {
    const long S = A->pat_span;
    if (lt < 4 || S <= 0 || S % 32 != 0) break;
    // canonical slots: the offsets of the interior stencil, chain first
    std::vector<long> offs;
    { bool m = false, c = false, q = false;
      for (int k = 0; k < lt; ++k) {
        const PatEntryH& e = tab[(size_t)common * lt + k];
        if (e.val == 0.0 && e.off == 0) continue;
        offs.push_back(e.off); m |= e.off == -S; c |= e.off == 0; q |= e.off == S;
      }
      if (!(m && c && q)) break; }
    std::vector<long> slot = {-S, 0, S};
    std::sort(offs.begin(), offs.end());
    // second longest offset L with both signs present: slots 3,4 (line exchange of spmm_pattern_chain2_kernel)
    long Lline = 0;
    for (long o : offs) { const long ao = o < 0 ? -o : o; if (ao < S && ao > Lline && std::binary_search(offs.begin(), offs.end(), -o)) Lline = ao; }
    if (Lline >= 8 && Lline % 8 == 0 && lt >= 5) { slot.push_back(-Lline); slot.push_back(Lline); } else Lline = 0;
    for (long o : offs) if (o != -S && o != 0 && o != S && !(Lline && (o == -Lline || o == Lline))) slot.push_back(o);
    const int nslot_used = (int)slot.size();
    if (nslot_used > lt) break;
    while ((int)slot.size() < lt) slot.push_back(0);            // unused slots: own row, value 0
    // per generic pattern: value on every canonical slot + the entries that fit no slot ("extras": halo columns of
    // a row slab).  An extra may ride in slot 0 of a row of the first S rows (no predecessor in the chain: slot 0 is
    // loaded explicitly when a wave starts) or in slot 2 of a row of the last S rows (no successor reads it).
    const int np = A->npat;
    std::vector<double> pval((size_t)np * lt, 0.0);
    std::vector<std::vector<PatEntryH>> extras((size_t)np);
    bool ok = true;
    for (int p = 0; p < np && ok; ++p)
      for (int k = 0; k < lt; ++k) {
        const PatEntryH& e = tab[(size_t)p * lt + k];
        if (e.val == 0.0 && e.off == 0) continue;
        int sidx = -1;
        for (int q = 0; q < nslot_used; ++q) if (slot[q] == e.off) { sidx = q; break; }
        if (sidx >= 0) pval[(size_t)p * lt + sidx] += e.val;
        else { extras[p].push_back(e); if (extras[p].size() > 2) ok = false; }
      }
    if (!ok) break;
    std::unordered_map<uint64_t, int> id_of;
    std::vector<PatEntryH> ctab;
    std::vector<unsigned short> cpid((size_t)nrows);
    for (int r = 0; r < nrows && ok; ++r) {
      unsigned mask = 0;   // slots whose canonical address leaves the block of vectors
      for (int q = 0; q < nslot_used; ++q) { const long c = (long)r + slot[q]; if (c < 0 || c >= ncols_local) mask |= 1u << q; }
      const unsigned head = r < S, tail = (long)r + S >= nrows;
      const uint64_t key = ((uint64_t)pid[r] << 32) | ((uint64_t)head << 31) | ((uint64_t)tail << 30) | mask;
      auto it = id_of.find(key);
      if (it == id_of.end()) {
        const int id = (int)(ctab.size() / lt);
        if (id >= maxpat) { ok = false; break; }
        std::vector<PatEntryH> row((size_t)lt);
        for (int q = 0; q < lt; ++q) {
          row[q].val = pval[(size_t)pid[r] * lt + q];
          row[q].off = (q < nslot_used && !(mask >> q & 1)) ? slot[q] : 0;
          if (mask >> q & 1) { if (row[q].val != 0.0) ok = false; row[q].val = 0.0; }   // an entry cannot point outside
        }
        for (const PatEntryH& e : extras[pid[r]]) {
          if (head && row[0].val == 0.0) row[0] = e;
          else if (tail && row[2].val == 0.0) row[2] = e;
          else ok = false;
        }
        for (int q = 0; q < lt; ++q) ctab.push_back(row[q]);
        it = id_of.emplace(key, id).first;
      }
      cpid[r] = (unsigned short)it->second;
    }
    if (!ok) break;
    tab.swap(ctab); pid.swap(cpid);
    A->npat = (int)(tab.size() / lt); A->pat_span2 = Lline ? -Lline : -1;
    A->pat_near = 0;
    if (Lline && lt == 7 && nslot_used == 7 && slot[5] == -1 && slot[6] == 1)
      for (const PatEntryH& e : tab) A->pat_near = std::max(A->pat_near, e.off < 0 ? -e.off : e.off);
  } while (0);
  if (by_offsets) {
    // the row's values in the slot order of ITS table row: an entry sits in the slot that carries its offset (the
    // diagonal in slot 1 of a chain-layout table, where slots whose address would leave the block also read offset 0)
    const bool chain_layout = A->pat_span2 <= -1;
    std::vector<double> rv((size_t)nrows * 8, 0.0);
    bool ok = true;
    for (int r = 0; r < nrows && ok; ++r) {
      const PatEntryH* e = &tab[(size_t)pid[r] * lt];
      unsigned used = 0;
      for (int q = rowptr[r]; q < rowptr[r + 1]; ++q) {
        const long off = (long)colidx[q] - r;
        int slot = -1;
        if (chain_layout && off == 0) slot = 1;
        else for (int k = 0; k < lt; ++k) if (!(used >> k & 1) && e[k].off == off && e[k].val != 0.0 && !(chain_layout && off == 0 && k != 1)) { slot = k; break; }
        if (slot < 0 || (used >> slot & 1)) { ok = false; break; }
        used |= 1u << slot;
        rv[(size_t)r * 8 + slot] = val[q];
      }
    }
    if (!ok) { A->npat = 0; return; }   // (cannot happen for tables built above; the matrix then keeps the generic kernels)
    GCGE_HIP_CHECK(hipMalloc(&A->d_rowval, rv.size() * sizeof(double)));
    GCGE_HIP_CHECK(hipMemcpy(A->d_rowval, rv.data(), rv.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  GCGE_HIP_CHECK(hipMalloc(&A->d_pid, (size_t)nrows * sizeof(unsigned short)));
  GCGE_HIP_CHECK(hipMalloc(&A->d_tab, tab.size() * sizeof(PatEntryH)));
  GCGE_HIP_CHECK(hipMemcpy(A->d_pid, pid.data(), (size_t)nrows * sizeof(unsigned short), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(A->d_tab, tab.data(), tab.size() * sizeof(PatEntryH), hipMemcpyHostToDevice));
}

// rows of one slab with LOCAL column indices in [0, ncols_local); columns >= nrows are halo rows.  ghost_global (may be NULL): the
// global rows behind the halo columns, ascending — with it (and nglobal) a slab of a grid matrix cut on plane boundaries keeps the
// plane sweep of spmm_star.hip: the planes below and above the slab are then found among the halo rows.
static double upload_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_local_ghosts(int nrows, int ncols_local, int nglobal, int row_begin,
                                                          const int* rowptr, const int* colidx, const double* val, const int* ghost_global) {
  if (gcge_hip_init(-1) != 0) return nullptr;
  const bool timing = getenv("GCGE_UPLOAD_TIMING") != nullptr;   // phases of the host-side analysis on stderr
  double t_phase = upload_now();
  auto phase = [&](const char* what) { if (timing) { const double t = upload_now(); fprintf(stderr, "gcge_hip upload: %-28s %.3f s\n", what, t - t_phase); t_phase = t; } };
  GCGE_HIP_MAT* A = (GCGE_HIP_MAT*)calloc(1, sizeof(GCGE_HIP_MAT));
  A->nrows = nrows; A->nglobal = nglobal; A->row_begin = row_begin; A->nnz = rowptr[nrows];
  A->nghost = ncols_local - nrows;
  GCGE_REQUIRE(A->nghost >= 0, "gcge_hip_mat_create_local: ncols_local >= nrows");
  if (ghost_global != nullptr && A->nghost > 0) {
    A->h_ghost_global = (int*)malloc((size_t)A->nghost * sizeof(int));
    memcpy(A->h_ghost_global, ghost_global, (size_t)A->nghost * sizeof(int));
  }
  {
    const int nt = gcge_upload_threads();
    std::vector<int> bad((size_t)nt, 0);
    gcge_parallel_chunks(A->nnz, nt, [&](int c, long k0, long k1) { int b = 0; for (long k = k0; k < k1; ++k) b |= (colidx[k] < 0) | (colidx[k] >= ncols_local); bad[c] = b; });
    for (int c = 0; c < nt; ++c) GCGE_REQUIRE(bad[c] == 0, "gcge_hip_mat_create_local: column index in range");
  }
  const size_t nnz = (size_t)A->nnz;
  GCGE_HIP_CHECK(hipMalloc(&A->d_rowptr, ((size_t)nrows + 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&A->d_colidx, (nnz ? nnz : 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&A->d_val, (nnz ? nnz : 1) * sizeof(double)));
  GCGE_HIP_CHECK(hipMemcpy(A->d_rowptr, rowptr, ((size_t)nrows + 1) * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(A->d_colidx, colidx, nnz * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(A->d_val, val, nnz * sizeof(double), hipMemcpyHostToDevice));
  // pad-8 copy: every row padded to a multiple of 8 entries with (own column, 0.0)
  std::vector<int> orp((size_t)nrows + 1);
  size_t noct = 0;
  for (int r = 0; r < nrows; ++r) { orp[r] = (int)noct; noct += ((size_t)(rowptr[r + 1] - rowptr[r]) + 7) / 8; }
  orp[nrows] = (int)noct;
  std::vector<int> pc(noct * 8);
  std::vector<double> pv(noct * 8);
  gcge_parallel_chunks(nrows, gcge_upload_threads(), [&](int, long r0, long r1) {
    for (long r = r0; r < r1; ++r) {
      size_t o = (size_t)orp[r] * 8; int k;
      for (k = rowptr[r]; k < rowptr[r + 1]; ++k, ++o) { pc[o] = colidx[k]; pv[o] = val[k]; }
      for (; o < (size_t)orp[r + 1] * 8; ++o) { pc[o] = (int)r; pv[o] = 0.0; }
    }
  });
  A->noct = (long)noct;
  GCGE_HIP_CHECK(hipMalloc(&A->d_orp, ((size_t)nrows + 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&A->d_pcol, (noct ? noct * 8 : 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMalloc(&A->d_pval, (noct ? noct * 8 : 1) * sizeof(double)));
  GCGE_HIP_CHECK(hipMemcpy(A->d_orp, orp.data(), ((size_t)nrows + 1) * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(A->d_pcol, pc.data(), noct * 8 * sizeof(int), hipMemcpyHostToDevice));
  GCGE_HIP_CHECK(hipMemcpy(A->d_pval, pv.data(), noct * 8 * sizeof(double), hipMemcpyHostToDevice));
  phase("CSR + pad-8 copies");
  build_patterns(A, nrows, ncols_local, rowptr, colidx, val);
  if (A->d_pid == nullptr && g_offset_patterns) build_patterns(A, nrows, ncols_local, rowptr, colidx, val, true);   // same stencil, other coefficients in every row
  // matrices without a pattern form whose rows are mostly ONE star stencil on a grid (free diagonal): those rows leave the CSR
  // arrays for the plane sweep of spmm_star.hip, the others (rows inside dense blocks, ...) keep every entry and take the block form
  phase("pattern search");
  A->star = nullptr; A->star_rem = nullptr;
  if (A->d_pid == nullptr) {
    const int *rr = nullptr, *rc = nullptr; const double* rv = nullptr;
    // (the sweep needs to know where the slab sits in the grid: a whole matrix, or a slab with its global size and halo rows named)
    const bool whole = row_begin == 0 && ncols_local == nrows && (nglobal <= 0 || nglobal == nrows);
    const bool slab = !whole && nglobal > 0 && (ncols_local == nrows || ghost_global != nullptr);
    void* S = (whole || slab) ? gcge_hip_star_build(nrows, ncols_local, row_begin, whole ? nrows : nglobal, ghost_global, rowptr, colidx, val, &rr, &rc, &rv) : nullptr;
    phase("star split");
    if (S != nullptr) {
      void* D = gcge_hip_dense_build_rows(nrows, ncols_local, rr, rc, rv, gcge_hip_star_host_mask());   // (its pad-8 part lists the other rows only)
      if (D != nullptr) { A->star = S; A->star_rem = D; } else gcge_hip_star_free(S);   // (no blocks among the other rows: the forms below)
      gcge_hip_star_release_remainder();
      phase("blocks + listed rows");
    }
  }
  // matrices without a pattern form: dense row blocks (supernodes) on MFMA + remainder CSR, where such blocks exist
  A->dense = (A->d_pid == nullptr && A->star == nullptr) ? gcge_hip_dense_build(nrows, ncols_local, rowptr, colidx, val) : nullptr;
  // ... and, where switched on, row tiles with LDS-staged X rows
  A->tile = ((A->d_pid == nullptr && A->star == nullptr) || gcge_hip_spmm_tile_mode_get() == 2) ? gcge_hip_tile_build(nrows, ncols_local, rowptr, colidx, val) : nullptr;
  // interior rows: none of them references a halo column (slabs: everything but the first and the last plane)
  A->ov_lo = 0; A->ov_hi = nrows;
  if (A->nghost > 0) {
    for (int r = 0; r < nrows; ++r) {
      bool g = false;
      for (int k = rowptr[r]; k < rowptr[r + 1] && !g; ++k) g = colidx[k] >= nrows;
      if (!g) continue;
      if (r < nrows / 2) A->ov_lo = r + 1; else { A->ov_hi = r; break; }
    }
  }
  return A;
}
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_local(int nrows, int ncols_local, int nglobal, int row_begin,
                                                   const int* rowptr, const int* colidx, const double* val) {
  return gcge_hip_mat_create_local_ghosts(nrows, ncols_local, nglobal, row_begin, rowptr, colidx, val, nullptr);
}
// ---------------------------------------------------------------------------------------------------------------- row orders
// (round 5)  A whole matrix (one rank) that shows neither a pattern form nor a grid in the order it arrives in is
// re-ordered INSIDE the handle: grid coordinates recovered from the graph of a star stencil (gcge_hip_reorder_star_grid: scan order,
// the plane sweep applies again) or reverse Cuthill-McKee (a banded matrix gathers from a window of X).  The solver never sees it:
// handles are opaque, blocks of vectors created for the matrix live in the new order, gcge_hip_mv_to_host / from_host
// and the reference-order random fill translate.  mode: 0 automatic (matrices of >= 65 536 rows without a fast form), 1 every matrix
// without a fast form (tests), -1 never.
extern "C" void gcge_hip_star_next_geometry(int nrows, int nx, int ny, int nz, const int* box_of_row);
extern "C" long gcge_hip_reorder_star_grid(int n, const int* rowptr, const int* colidx, const double* val, int* dims, int* box_of_row);
extern "C" int gcge_hip_reorder_rcm(int n, const int* rowptr, const int* colidx, int* perm);
extern "C" double gcge_hip_mean_bandwidth(int n, const int* rowptr, const int* colidx, const int* iperm);
static int g_reorder_mode = 0;
extern "C" void gcge_hip_spmm_reorder_mode(int mode) { g_reorder_mode = mode; }
static std::vector<GcgePerm*> g_perms;          // live orders, one per n
static unsigned g_perm_next_id = 1;
static char g_reorder_kind[64] = "";
extern "C" const char* gcge_hip_mat_row_order(const GCGE_HIP_MAT* A) {
  if (A == nullptr || A->perm == nullptr || A->perm->identity) return "as given";
  return g_reorder_kind[0] ? g_reorder_kind : "reordered";
}
extern "C" GcgePerm* gcge_hip_perm_acquire(GcgePerm* p) { if (p) ++p->refs; return p; }
extern "C" void gcge_hip_perm_release(GcgePerm* p) {
  if (p == nullptr || --p->refs > 0) return;
  for (size_t i = 0; i < g_perms.size(); ++i) if (g_perms[i] == p) { g_perms.erase(g_perms.begin() + (long)i); break; }
  free(p->perm); free(p->iperm); free(p);
}
static GcgePerm* perm_find(int n) { for (GcgePerm* p : g_perms) if (p->n == n) return p; return nullptr; }
static GcgePerm* perm_register(int n, const int* perm /* NULL: identity */) {
  GcgePerm* p = (GcgePerm*)calloc(1, sizeof(GcgePerm));
  p->n = n; p->identity = perm == nullptr; p->id = g_perm_next_id++;
  if (perm != nullptr) {
    p->perm = (int*)malloc((size_t)n * sizeof(int)); p->iperm = (int*)malloc((size_t)n * sizeof(int));
    memcpy(p->perm, perm, (size_t)n * sizeof(int));
    for (int i = 0; i < n; ++i) p->iperm[perm[i]] = i;
  }
  g_perms.push_back(p);
  return p;
}
// P A P^T as CSR with ascending columns: new row i = old row perm[i], old column c -> iperm[c]
static void permute_csr(int n, const int* rowptr, const int* colidx, const double* val, const GcgePerm* P,
                        std::vector<int>& rp, std::vector<int>& ci, std::vector<double>& va) {
  rp.resize((size_t)n + 1); ci.resize((size_t)rowptr[n] ? (size_t)rowptr[n] : 1); va.resize(ci.size());
  rp[0] = 0;
  for (int i = 0; i < n; ++i) rp[i + 1] = rp[i] + (rowptr[P->perm[i] + 1] - rowptr[P->perm[i]]);
  gcge_parallel_chunks(n, gcge_upload_threads(), [&](int, long r0, long r1) {
    std::vector<std::pair<int, double>> tmp;
    for (long i = r0; i < r1; ++i) {
      const int o = P->perm[i];
      tmp.clear();
      for (int k = rowptr[o]; k < rowptr[o + 1]; ++k) tmp.emplace_back(P->iperm[colidx[k]], val[k]);
      std::sort(tmp.begin(), tmp.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      int q = rp[i];
      for (auto& e : tmp) { ci[q] = e.first; va[q] = e.second; ++q; }
    }
  });
}
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create(int nrows, int nglobal, int row_begin, const int* rowptr,
                                             const int* colidx, const double* val) {
  if (row_begin != 0 || nrows != nglobal) {
    fprintf(stderr, "gcge_hip_mat_create: a row slab needs gcge_dist_localize + gcge_hip_mat_create_local\n");
    return nullptr;
  }
  const bool timing = getenv("GCGE_UPLOAD_TIMING") != nullptr;
  std::vector<int> rp, ci; std::vector<double> va;
  // a matrix of this size lives in this process already: the new one shares its row order (B after A of a generalised problem)
  if (GcgePerm* P = perm_find(nrows)) {
    if (P->identity) {
      GCGE_HIP_MAT* A = gcge_hip_mat_create_local(nrows, nrows, nglobal, 0, rowptr, colidx, val);
      if (A != nullptr) A->perm = gcge_hip_perm_acquire(P);
      return A;
    }
    permute_csr(nrows, rowptr, colidx, val, P, rp, ci, va);
    GCGE_HIP_MAT* A = gcge_hip_mat_create_local(nrows, nrows, nglobal, 0, rp.data(), ci.data(), va.data());
    if (A != nullptr) A->perm = gcge_hip_perm_acquire(P);
    return A;
  }
  GCGE_HIP_MAT* A = gcge_hip_mat_create_local(nrows, nrows, nglobal, 0, rowptr, colidx, val);
  if (A == nullptr) return nullptr;
  const bool fast = A->d_pid != nullptr || A->star != nullptr;
  const bool want = g_reorder_mode == 1 || (g_reorder_mode == 0 && nrows >= 65536);
  if (fast || !want || nrows < 64) { A->perm = gcge_hip_perm_acquire(perm_register(nrows, nullptr)); return A; }
  // no fast form in the order given: look for a better one
  double t0 = upload_now();
  std::vector<int> perm((size_t)nrows), box((size_t)nrows);
  int dims[3] = {0, 0, 0};
  bool have = false, grid = false;
  const long placed = gcge_hip_reorder_star_grid(nrows, rowptr, colidx, val, dims, box.data());
  if (placed > 0) {                                       // scan order of the recovered grid: ascending box index
    for (int i = 0; i < nrows; ++i) perm[i] = i;
    std::sort(perm.begin(), perm.end(), [&](int a, int b) { return box[a] < box[b]; });
    have = grid = true;
    snprintf(g_reorder_kind, sizeof g_reorder_kind, "grid %d x %d x %d recovered from the star couplings", dims[0], dims[1], dims[2]);
  } else if (gcge_hip_reorder_rcm(nrows, rowptr, colidx, perm.data()) == 0) {
    std::vector<int> ip((size_t)nrows);
    for (int i = 0; i < nrows; ++i) ip[perm[i]] = i;
    const double b0 = gcge_hip_mean_bandwidth(nrows, rowptr, colidx, nullptr), b1 = gcge_hip_mean_bandwidth(nrows, rowptr, colidx, ip.data());
    have = b1 < 0.5 * b0;                                 // (a matrix that is banded already stays as it is)
    snprintf(g_reorder_kind, sizeof g_reorder_kind, "reverse Cuthill-McKee (mean |i - j| %.0f -> %.0f)", b0, b1);
  }
  if (timing) fprintf(stderr, "gcge_hip upload: %-28s %.3f s (%s)\n", "row order", upload_now() - t0, have ? g_reorder_kind : "kept as given");
  if (!have) { A->perm = gcge_hip_perm_acquire(perm_register(nrows, nullptr)); return A; }
  gcge_hip_mat_destroy(A);
  GcgePerm* P = perm_register(nrows, perm.data());
  permute_csr(nrows, rowptr, colidx, val, P, rp, ci, va);
  if (grid && (long)dims[0] * dims[1] * dims[2] != nrows) {   // a masked grid (ball): name the geometry for the plane sweep's row map
    std::vector<int> bsorted((size_t)nrows);
    for (int i = 0; i < nrows; ++i) bsorted[i] = box[perm[i]];
    gcge_hip_star_next_geometry(nrows, dims[0], dims[1], dims[2], bsorted.data());
    A = gcge_hip_mat_create_local(nrows, nrows, nglobal, 0, rp.data(), ci.data(), va.data());
    gcge_hip_star_next_geometry(0, 0, 0, 0, nullptr);
  } else {
    A = gcge_hip_mat_create_local(nrows, nrows, nglobal, 0, rp.data(), ci.data(), va.data());
  }
  if (A == nullptr) { gcge_hip_perm_release(gcge_hip_perm_acquire(P)); return nullptr; }
  A->perm = gcge_hip_perm_acquire(P);
  return A;
}
// A matrix on a MASKED grid (one rank): row r is grid point box_of_row[r] = x + nx (y + ny z) of an nx x ny x nz box, rows in scan
// order — the real-space DFT matrices behind BASELINE config 5 live on the grid points inside a sphere (PARSEC).  With the
// geometry named, the rows that are a star stencil on that grid take the plane sweep of spmm_star.hip (through a row map);
// without it such a matrix is served by dense blocks + the pad-8 kernel.  The geometry only selects kernels: results are those
// of gcge_hip_mat_create on the same arrays.
extern "C" void gcge_hip_star_next_geometry(int nrows, int nx, int ny, int nz, const int* box_of_row);
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_grid(int nrows, const int* rowptr, const int* colidx, const double* val,
                                                  int nx, int ny, int nz, const int* box_of_row) {
  if (box_of_row != nullptr && nx > 0 && ny > 0 && nz > 0) gcge_hip_star_next_geometry(nrows, nx, ny, nz, box_of_row);
  GCGE_HIP_MAT* A = gcge_hip_mat_create_local(nrows, nrows, nrows, 0, rowptr, colidx, val);      // (the caller named the geometry: the rows stay as given)
  if (A != nullptr) { GcgePerm* P = perm_find(nrows); if (P == nullptr) P = perm_register(nrows, nullptr); if (P->identity) A->perm = gcge_hip_perm_acquire(P); }
  gcge_hip_star_next_geometry(0, 0, 0, 0, nullptr);                   // (not consumed when the matrix took a pattern form)
  return A;
}
extern "C" GCGE_HIP_MAT* gcge_hip_mat_create_csr(const GCGE_CSR* A) {
  if (A->row_begin == 0 && A->nrows == A->ncols) return gcge_hip_mat_create(A->nrows, A->ncols, 0, A->rowptr, A->colidx, A->val);
  // a localized slab: ncols = nrows + nghost (gcge_dist_localize)
  return gcge_hip_mat_create_local(A->nrows, A->ncols, -1, A->row_begin, A->rowptr, A->colidx, A->val);
}
// halo plan: send_rows = local rows to ship (grouped by destination rank, ascending), buffers hold
// buf_cols columns of nsend / nghost rows; exchange() moves sendbuf -> the peers' recvbuf.
extern "C" void gcge_hip_mat_set_halo(GCGE_HIP_MAT* A, int nglobal, int nsend, const int* send_rows, double* sendbuf,
                                      double* recvbuf, int buf_cols, gcge_halo_exchange_fn fn, void* ctx) {
  A->nglobal = nglobal; A->nsend = nsend; A->sendbuf = sendbuf; A->recvbuf = recvbuf; A->buf_cols = buf_cols;
  A->exchange = fn; A->exchange_ctx = ctx;
  for (int i = 0; i < nsend; ++i) GCGE_REQUIRE(send_rows[i] >= 0 && send_rows[i] < A->nrows, "halo send row in range");
  if (A->d_send_rows) hipFree(A->d_send_rows);
  GCGE_HIP_CHECK(hipMalloc(&A->d_send_rows, (nsend ? nsend : 1) * sizeof(int)));
  GCGE_HIP_CHECK(hipMemcpy(A->d_send_rows, send_rows, nsend * sizeof(int), hipMemcpyHostToDevice));
}
// optional: a split form of the exchange installed by gcge_hip_mat_set_halo — begin(sendbuf, recvbuf, ncols, ctx)
// posts the transfers of the packed rows and returns, end(ctx) returns when recvbuf is complete
extern "C" void gcge_hip_mat_set_halo_async(GCGE_HIP_MAT* A, gcge_halo_exchange_fn begin, void (*end)(void*)) {
  A->exchange_begin = begin; A->exchange_end = end;
}
extern "C" void gcge_hip_mat_destroy(GCGE_HIP_MAT* A) {
  if (!A) return;
  hipFree(A->d_rowptr); hipFree(A->d_colidx); hipFree(A->d_val);
  if (A->rect_ncols > 0) { hipFree(A->d_t_rowptr); hipFree(A->d_t_colidx); hipFree(A->d_t_val); gcge_hip_perm_release(A->perm); free(A); return; }   // a prolongation (multigrid.hip)
  hipFree(A->d_orp); hipFree(A->d_pcol); hipFree(A->d_pval);
  if (A->d_pid) { hipFree(A->d_pid); hipFree(A->d_tab); }
  if (A->d_rowval) hipFree(A->d_rowval);
  if (A->d_send_rows) hipFree(A->d_send_rows);
  if (A->tile != nullptr) gcge_hip_tile_free(A->tile);
  if (A->dense != nullptr) gcge_hip_dense_free(A->dense);
  if (A->star_rem != nullptr) gcge_hip_dense_free(A->star_rem);
  if (A->star != nullptr) gcge_hip_star_free(A->star);
  if (A->native_halo != nullptr) gcge_hip_halo_native_free(A);   // RCCL plan + the exchange buffers it owns (rccl_comm.hip)
  free(A->h_ghost_global); free(A->h_part);
  gcge_hip_perm_release(A->perm);
  free(A);
}
// the row partition of all ranks a slab belongs to (world + 1 offsets): recorded by gcge_hip_mat_create_slab, or by whoever built
// the slab through gcge_hip_mat_create_local_ghosts + its own halo plan — MultiGridCreate coarsens a slab only when it knows it
extern "C" void gcge_hip_mat_set_partition(GCGE_HIP_MAT* A, const long* part, int world) {
  free(A->h_part); A->h_part = nullptr; A->part_world = 0;
  if (part == nullptr || world < 1) return;
  A->h_part = (long*)malloc((size_t)(world + 1) * sizeof(long));
  memcpy(A->h_part, part, (size_t)(world + 1) * sizeof(long));
  A->part_world = world;
}
extern "C" int gcge_hip_mat_nrows(const GCGE_HIP_MAT* A) { return A->nrows; }
extern "C" long gcge_hip_mat_nnz(const GCGE_HIP_MAT* A) { return A->nnz; }
// number of row patterns the SpMM pattern path works with (0: the matrix is served by the generic pad-8 kernels)
extern "C" int gcge_hip_mat_patterns(const GCGE_HIP_MAT* A) { return A->d_pid ? A->npat : 0; }
// 1: the pattern table is in chain layout (the +-S rows of the stencil stay in registers between iterations);
// 2: additionally slots 3,4 hold the +-L rows that the waves of a block exchange through LDS
extern "C" int gcge_hip_mat_pattern_chain(const GCGE_HIP_MAT* A) {
  if (!A->d_pid || A->pat_span2 > -1) return 0;
  return A->pat_span2 <= -8 ? 2 : 1;
}

// which K1 form MatDotMultiVec takes for this matrix at block widths >= 16 (bench.py names the kernel in its roofline)
extern "C" const char* gcge_hip_mat_spmm_form(const GCGE_HIP_MAT* A) {
  if (A->d_pid != nullptr && g_spmm_path == 0) {
    const int ch = gcge_hip_mat_pattern_chain(A);
    if (A->d_rowval != nullptr) return ch == 2 ? "spmm_pattern_chain2+values" : "spmm_pattern+values";   // offsets-only table, values per row
    return ch == 2 ? "spmm_pattern_chain2" : ch == 1 ? "spmm_pattern_chain" : "spmm_pattern";
  }
  if (A->star != nullptr && g_spmm_path == 0) return gcge_hip_dense_remainder_is_tiled(A->star_rem) ? "spmm_star+spmm_dense+spmm_tile" : "spmm_star+spmm_dense+spmm_pad8";
  if (A->dense != nullptr && g_spmm_path != 1 && g_spmm_path != 3 && g_spmm_path != 4) return gcge_hip_dense_remainder_is_tiled(A->dense) ? "spmm_dense+spmm_tile" : "spmm_dense+spmm_pad8";
  if (A->tile != nullptr && g_spmm_path != 1 && g_spmm_path != 3) return "spmm_tile";
  return "spmm_pad8";
}

// what the upload made of a matrix without a pattern form (measurement aid): out[0..4] = dense blocks, row blocks of 32, non-zeros
// in blocks, stored block entries, remainder non-zeros; out[5..11] = tiles of the remainder (0: pad-8), X rows staged per matrix
// row, ELL entries per non-zero, overflow entries, brick dimensions.  0: the matrix has no block form.
extern "C" int gcge_hip_mat_form_stats(const GCGE_HIP_MAT* A, double* out) {
  for (int i = 0; i < 12; ++i) out[i] = 0.0;
  const void* DM = A->star != nullptr ? A->star_rem : A->dense;       // (with a grid form: the block form of the rows it leaves)
  if (DM == nullptr) return 0;
  long nb = 0, items = 0, dn = 0, de = 0, rn = 0;
  gcge_hip_dense_stats(DM, &nb, &items, &dn, &de, &rn);
  out[0] = (double)nb; out[1] = (double)items; out[2] = (double)dn; out[3] = (double)de; out[4] = (double)rn;
  if (const void* T = gcge_hip_dense_remainder_tile(DM)) {
    long nt = 0, ov = 0; double xr = 0, el = 0; int brick[3] = {0, 0, 0}; long strides[2];
    gcge_hip_tile_stats(T, &nt, &ov, &xr, &el, brick, strides);
    out[5] = (double)nt; out[6] = xr; out[7] = el; out[8] = (double)ov; out[9] = brick[0]; out[10] = brick[1]; out[11] = brick[2];
  }
  return 1;
}
extern "C" int gcge_hip_star_masked_form(const void* sm);
extern "C" int gcge_hip_mat_star_masked_form(const GCGE_HIP_MAT* A) { return A->star != nullptr ? gcge_hip_star_masked_form(A->star) : 0; }
// the grid form (spmm_star.hip): out[0..5] = nx, ny, nz, arm length of the star, rows it multiplies, rows of the matrix.  0: none.
extern "C" int gcge_hip_mat_star_stats(const GCGE_HIP_MAT* A, long* out) {
  if (A->star == nullptr) return 0;
  gcge_hip_star_stats(A->star, out);
  return 1;
}

