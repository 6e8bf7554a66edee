// Row orders for matrices that show NO grid in the order they arrive in (round 5) — host code.
//
// Every fast K1 form of this back-end reads structure off the ROW ORDER: constant column offsets (patterns), a lexicographic or
// scan-ordered grid (plane sweep), neighbouring rows sharing columns (tiles).  A matrix file is free to number its unknowns any
// other way (the reference's run list holds SuiteSparse matrices, test/submit.sh:9-10; its FE provider numbers tetrahedral meshes,
// test/get_mat_phg.c:148), and then every non-zero gathers a 512-byte row of X from anywhere in a multi-GB block: 8-12 % of the
// roofline (profiles/r05_generic/).  The handles are opaque to the solver (SURVEY 8b "Layout opacity"), so the back-end may keep
// A' = P A P^T and every block of vectors in the permuted order — only the transfers to / from the host (gcge_hip_mv_to_host /
// from_host, the reference-order random fill) translate.  This file finds P:
//   1. gcge_hip_reorder_star_grid: a matrix whose rows are (mostly) ONE isotropic star stencil on a grid — the finite-difference
//      Hamiltonians behind BASELINE config 5 — gets its grid coordinates back from the graph of its distance-1 couplings alone
//      (flood fill with a transported frame), whatever the numbering: P = scan order, and the plane sweep applies again;
//   2. gcge_hip_reorder_rcm: anything else gets reverse Cuthill-McKee (banded: the gathers of neighbouring rows overlap).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <map>
#include <queue>
#include <unordered_map>
#include <vector>

// ---------------------------------------------------------------------------------------------------------------- RCM
// perm[new] = old.  Components in order of their lowest-degree node; inside a component: BFS from a pseudo-peripheral node (two
// sweeps of the Gibbs-Poole-Stockmeyer kind), neighbours visited by ascending degree; the whole order reversed.
extern "C" int gcge_hip_reorder_rcm(int n, const int* rowptr, const int* colidx, int* perm) {
  std::vector<int> deg(n), order; order.reserve(n);
  for (int r = 0; r < n; ++r) deg[r] = rowptr[r + 1] - rowptr[r];
  std::vector<char> seen(n, 0);
  std::vector<int> byDeg(n);
  for (int r = 0; r < n; ++r) byDeg[r] = r;
  std::sort(byDeg.begin(), byDeg.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
  std::vector<int> level, nbr;
  auto bfs_last = [&](int start, std::vector<int>& visited) -> int {   // BFS over the still-unseen part; returns a node of the last level with minimal degree
    visited.clear(); visited.push_back(start);
    std::vector<char>& mark = seen;                     // temporarily marks with 2
    mark[start] = 2;
    size_t head = 0, level_begin = 0, last_begin = 0;
    while (head < visited.size()) {
      const size_t level_end = visited.size();
      last_begin = level_begin;
      for (; head < level_end; ++head) {
        const int u = visited[head];
        for (int k = rowptr[u]; k < rowptr[u + 1]; ++k) { const int v = colidx[k]; if (v < n && !mark[v]) { mark[v] = 2; visited.push_back(v); } }
      }
      level_begin = level_end;
    }
    int best = visited[last_begin];
    for (size_t i = last_begin; i < visited.size(); ++i) if (deg[visited[i]] < deg[best]) best = visited[i];
    for (int u : visited) mark[u] = 0;
    return best;
  };
  std::vector<int> visited;
  for (int s0 : byDeg) {
    if (seen[s0]) continue;
    int s = bfs_last(s0, visited);                      // far end of the component from a low-degree node ...
    s = bfs_last(s, visited);                           // ... and the far end from there: a pseudo-peripheral node
    size_t head = order.size();
    order.push_back(s); seen[s] = 1;
    while (head < order.size()) {
      const int u = order[head++];
      nbr.clear();
      for (int k = rowptr[u]; k < rowptr[u + 1]; ++k) { const int v = colidx[k]; if (v < n && !seen[v]) { seen[v] = 1; nbr.push_back(v); } }
      std::sort(nbr.begin(), nbr.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
      order.insert(order.end(), nbr.begin(), nbr.end());
    }
  }
  if ((int)order.size() != n) return -1;
  for (int i = 0; i < n; ++i) perm[i] = order[n - 1 - i];
  return 0;
}

// mean |row - column| over the entries: what a reorder is judged by (a banded matrix gathers from a window of X, not from all of it)
extern "C" double gcge_hip_mean_bandwidth(int n, const int* rowptr, const int* colidx, const int* iperm /* old -> new, or NULL */) {
  double s = 0.0; long cnt = 0;
  const long step = n > (1 << 20) ? n / (1 << 20) : 1;
  for (long r = 0; r < n; r += step)
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int c = colidx[k]; if (c >= n) continue;
      const long a = iperm ? iperm[r] : r, b = iperm ? iperm[c] : c;
      s += (double)(a > b ? a - b : b - a); ++cnt;
    }
  return cnt ? s / (double)cnt : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------- star grids
// Grid coordinates of the rows of a matrix that is (mostly) one isotropic star stencil, from its distance-1 couplings alone.
//   c1 = the off-diagonal value of largest magnitude among the most frequent ones (the distance-1 coefficient of a finite-difference
//   Laplacian of any order); N1(r) = the columns of row r holding exactly c1 (<= 6).  In a grid two members of N1(r) lie on ONE
//   line through r iff r is their only common N1-neighbour (members on different axes close a 4-cycle through r + e_a + e_b).
//   A frame F(r): direction (+-x, +-y, +-z) -> member of N1(r) is transported along every step: for n = F(r)[d],
//   F(n)[-d] = r, F(n)[d] = the member of N1(n) opposite r, F(n)[e] = the member adjacent to F(r)[e] (the 4-cycle).  A flood fill from
//   one interior row assigns coordinates; a row reached with two different coordinates ends the attempt (the graph is no grid).
// Rows the fill cannot reach (inside a dense "atom" block every coupling is perturbed) are placed by their exact stencil entries to
// placed rows — an entry equal to c_k to a row q puts r at distance k from q on one of three lines; two such entries on different
// lines fix the point — and whatever is left takes the free positions in ascending order (the star split of the upload is exact for
// ANY one-to-one placement: a misplaced row costs speed, never the result).
// Out: dims[3] (box of the coordinates), box_of_row[r] = x + nx (y + ny z).  Returns the number of rows placed by the fill (0: no such grid).
static inline uint64_t dbits(double v) { uint64_t b; memcpy(&b, &v, 8); return b; }

static int g_reason = 0;      // why the last attempt gave up (diagnostics): line number of the return
extern "C" int gcge_hip_reorder_last_reason(void) { return g_reason; }
#define GIVE_UP do { g_reason = __LINE__; return 0; } while (0)
extern "C" long gcge_hip_reorder_star_grid(int n, const int* rowptr, const int* colidx, const double* val, int* dims, int* box_of_row) {
  g_reason = 0;
  if (n < 64) GIVE_UP;
  // 1. the stencil's coefficient values: the frequent off-diagonal values of a sample of rows
  std::vector<uint64_t> vals_;                                          // (sorted and run-length counted: a std::map of millions of distinct values is slow)
  const long step = n > 200000 ? n / 200000 : 1;
  long sampled = 0;
  for (long r = 0; r < n; r += step) {
    if (rowptr[r + 1] - rowptr[r] > 64) continue;                        // (rows inside dense blocks: hundreds of unrelated values)
    ++sampled;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) if (colidx[k] != r && val[k] != 0.0) vals_.push_back(dbits(val[k]));
  }
  if (sampled < 32) GIVE_UP;
  std::sort(vals_.begin(), vals_.end());
  std::vector<std::pair<uint64_t, long>> hist;
  for (size_t i = 0; i < vals_.size(); ) { size_t j = i; while (j < vals_.size() && vals_[j] == vals_[i]) ++j; hist.emplace_back(vals_[i], (long)(j - i)); i = j; }
  long maxc = 0;
  for (auto& kv : hist) maxc = std::max(maxc, kv.second);
  if (maxc < 2 * sampled) GIVE_UP;                                     // (an interior star row carries every coefficient 6 times)
  std::vector<double> coef;                                             // by descending magnitude: c1, c2, ...
  for (auto& kv : hist) if (2 * kv.second >= maxc) { double v; memcpy(&v, &kv.first, 8); coef.push_back(v); }
  std::sort(coef.begin(), coef.end(), [](double a, double b) { return std::fabs(a) > std::fabs(b); });
  if (coef.empty() || coef.size() > 16) GIVE_UP;
  const uint64_t c1 = dbits(coef[0]);
  // 2. N1 lists (<= 6 per row, -1 padded)
  std::vector<int> n1((size_t)n * 6, -1); std::vector<unsigned char> n1c(n, 0);
  for (int r = 0; r < n; ++r) {
    int c = 0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
      if (colidx[k] != r && colidx[k] < n && dbits(val[k]) == c1) { if (c == 6) { c = 7; break; } n1[(size_t)r * 6 + c++] = colidx[k]; }
    n1c[r] = (unsigned char)(c <= 6 ? c : 0);                           // (more than 6: no star row; it takes no part)
    if (c > 6) for (int i = 0; i < 6; ++i) n1[(size_t)r * 6 + i] = -1;
  }
  // exact test of one entry: A[a][b] == the coefficient with these bits (columns ascending inside a row)
  auto entry_is = [&](int a, int b, uint64_t bits) -> bool {
    const int* lo_ = colidx + rowptr[a]; const int* hi_ = colidx + rowptr[a + 1];
    const int* it = std::lower_bound(lo_, hi_, b);
    return it != hi_ && *it == b && dbits(val[it - colidx]) == bits;
  };
  const bool have_c2 = coef.size() >= 2;
  const uint64_t c2 = have_c2 ? dbits(coef[1]) : 0;
  auto is_n1 = [&](int a, int b) -> bool { for (int i = 0; i < n1c[a]; ++i) if (n1[(size_t)a * 6 + i] == b) return true; return false; };
  auto common_n1 = [&](int a, int b) -> int { int c = 0; for (int i = 0; i < n1c[a]; ++i) if (is_n1(b, n1[(size_t)a * 6 + i])) ++c; return c; };
  // a and b (both members of N1(r)) lie on one line through r: with a second coefficient, A[a][b] == c2 exactly (positive evidence: an
  // entry perturbed by a dense block decides nothing); 7-point stencils: r is their only common N1-neighbour
  auto on_one_line = [&](int a, int b) -> bool { return have_c2 ? entry_is(a, b, c2) : (common_n1(a, b) == 1 && !is_n1(a, b)); };
  // 3. flood fill with a transported frame; directions 0..5 = +x -x +y -y +z -z, opposite(d) = d ^ 1
  std::vector<int> cx(n, INT32_MIN), cy(n), cz(n);
  std::vector<int> frame((size_t)n * 6, -1);
  int seed = -1;
  for (long r = n / 2; r < n && seed < 0; ++r) if (n1c[r] == 6) {       // an interior row whose six neighbours pair up into three lines
    int* f = &frame[(size_t)r * 6]; const int* nb = &n1[(size_t)r * 6];
    bool used[6] = {false, false, false, false, false, false}; int axis = 0; bool ok = true;
    for (int i = 0; i < 6 && ok; ++i) {
      if (used[i]) continue;
      int opp = -1;
      for (int j = i + 1; j < 6; ++j) if (!used[j] && on_one_line(nb[i], nb[j])) { if (opp >= 0) { ok = false; break; } opp = j; }
      if (opp < 0 || axis >= 3) { ok = false; break; }
      f[2 * axis] = nb[i]; f[2 * axis + 1] = nb[opp]; used[i] = used[opp] = true; ++axis;
    }
    if (ok && axis == 3) seed = (int)r; else for (int i = 0; i < 6; ++i) f[i] = -1;
  }
  if (seed < 0) GIVE_UP;
  static const int dx[6] = {1, -1, 0, 0, 0, 0}, dy[6] = {0, 0, 1, -1, 0, 0}, dz[6] = {0, 0, 0, 0, 1, -1};
  std::vector<int> queue; queue.reserve(n);
  std::unordered_map<uint64_t, int> occmap; occmap.reserve((size_t)n * 2);          // packed coordinate -> row, while the box is unknown
  auto key = [](int x, int y, int z) -> uint64_t { return ((uint64_t)(uint32_t)(x + (1 << 20)) << 42) | ((uint64_t)(uint32_t)(y + (1 << 20)) << 21) | (uint64_t)(uint32_t)(z + (1 << 20)); };
  cx[seed] = cy[seed] = cz[seed] = 0; queue.push_back(seed); occmap[key(0, 0, 0)] = seed;
  long placed = 0;
  // passes over everything placed so far until a pass places nothing new: a row popped before its neighbours had coordinates gets
  // its frame completed (and its remaining directions expanded) when it is visited again
  for (size_t before = 0; before != queue.size(); ) {
   before = queue.size();
   for (size_t head = 0; head < queue.size(); ++head) {
    const int r = queue[head];
    int* fr = &frame[(size_t)r * 6];
    // complete my frame from neighbours that already have coordinates
    for (int i = 0; i < n1c[r]; ++i) {
      const int m = n1[(size_t)r * 6 + i];
      if (cx[m] == INT32_MIN) continue;
      const int ddx = cx[m] - cx[r], ddy = cy[m] - cy[r], ddz = cz[m] - cz[r];
      if (std::abs(ddx) + std::abs(ddy) + std::abs(ddz) != 1) GIVE_UP;      // two N1-neighbours that are not grid neighbours: no grid
      const int d = ddx == 1 ? 0 : ddx == -1 ? 1 : ddy == 1 ? 2 : ddy == -1 ? 3 : ddz == 1 ? 4 : 5;
      if (fr[d] >= 0 && fr[d] != m) GIVE_UP;
      fr[d] = m;
    }
    for (int d = 0; d < 6; ++d) {
      const int nn = fr[d];
      if (nn < 0) continue;
      const int px = cx[r] + dx[d], py = cy[r] + dy[d], pz = cz[r] + dz[d];
      const bool fresh = cx[nn] == INT32_MIN;
      if (!fresh) { if (cx[nn] != px || cy[nn] != py || cz[nn] != pz) GIVE_UP; }
      else {
        if (occmap.count(key(px, py, pz))) GIVE_UP;                          // two rows on one grid point: no grid
        cx[nn] = px; cy[nn] = py; cz[nn] = pz; occmap[key(px, py, pz)] = nn;
      }
      // the frame travels whether or not nn had its coordinates already (a row placed by the rule below knows one direction only)
      int* fn = &frame[(size_t)nn * 6];
      if (fn[d ^ 1] >= 0 && fn[d ^ 1] != r) GIVE_UP;
      fn[d ^ 1] = r;
      for (int i = 0; i < n1c[nn] && fn[d] < 0; ++i) {                      // the member of N1(n) opposite r: continues the line
        const int m = n1[(size_t)nn * 6 + i];
        if (m != r && on_one_line(m, r)) fn[d] = m;
      }
      for (int e = 0; e < 6; ++e) {                                        // the other axes: adjacent to my own e-neighbour (4-cycle)
        if ((e >> 1) == (d >> 1) || fr[e] < 0 || fn[e] >= 0) continue;
        const int re = fr[e];
        for (int i = 0; i < n1c[nn]; ++i) { const int m = n1[(size_t)nn * 6 + i]; if (m != r && is_n1(m, re)) { fn[e] = m; break; } }
      }
      if (!fresh) continue;
      queue.push_back(nn);
    }
    // neighbours whose direction no frame knows (a coupling of the 4-cycle is perturbed or lies outside): the free grid point next to r
    // that is next to EVERY placed N1-neighbour of m — placed when exactly one qualifies (two placed neighbours on a line: their
    // midpoint; on a diagonal: the corner of their square that is still free)
    for (int i = 0; i < n1c[r]; ++i) {
      const int m = n1[(size_t)r * 6 + i];
      if (cx[m] != INT32_MIN) continue;
      int found = -1, nfound = 0;
      for (int d = 0; d < 6; ++d) {
        const int px = cx[r] + dx[d], py = cy[r] + dy[d], pz = cz[r] + dz[d];
        if (occmap.count(key(px, py, pz))) continue;
        bool ok = true; int others = 0;
        for (int j = 0; j < n1c[m] && ok; ++j) {
          const int q = n1[(size_t)m * 6 + j];
          if (q == r || cx[q] == INT32_MIN) continue;
          ++others;
          ok = std::abs(cx[q] - px) + std::abs(cy[q] - py) + std::abs(cz[q] - pz) == 1;
        }
        if (ok && others > 0) { found = d; ++nfound; }
      }
      if (nfound == 1) {
        const int px = cx[r] + dx[found], py = cy[r] + dy[found], pz = cz[r] + dz[found];
        cx[m] = px; cy[m] = py; cz[m] = pz; occmap[key(px, py, pz)] = m;
        fr[found] = m; frame[(size_t)m * 6 + (found ^ 1)] = r;
        queue.push_back(m);
      }
    }
   }
  }
  placed = (long)queue.size();
  if (getenv("GCGE_REORDER_TRACE") != nullptr) {
    const int* f = &frame[(size_t)seed * 6];
    fprintf(stderr, "gcge_hip_reorder_star_grid: %zu coefficients (c1 = %.6g%s), seed row %d (frame %d %d %d %d %d %d), fill placed %ld of %d rows\n", coef.size(), coef[0],
            have_c2 ? ", lines by c2" : ", lines by common neighbours", seed, f[0], f[1], f[2], f[3], f[4], f[5], placed, n);
  }
  if (placed * 10 < (long)n * 6 && getenv("GCGE_REORDER_FORCE") == nullptr) GIVE_UP;                                  // fewer than 60 % of the rows are a star grid: not this kind of matrix
  // 4. box, occupancy
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (int r = 0; r < n; ++r) if (cx[r] != INT32_MIN) {
    lo[0] = std::min(lo[0], cx[r]); hi[0] = std::max(hi[0], cx[r]); lo[1] = std::min(lo[1], cy[r]); hi[1] = std::max(hi[1], cy[r]);
    lo[2] = std::min(lo[2], cz[r]); hi[2] = std::max(hi[2], cz[r]);
  }
  const long nx = (long)hi[0] - lo[0] + 1, ny = (long)hi[1] - lo[1] + 1, nz = (long)hi[2] - lo[2] + 1;
  if (getenv("GCGE_REORDER_TRACE") != nullptr) {
    long nframe[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < n; ++r) if (cx[r] != INT32_MIN) { int c = 0; for (int d = 0; d < 6; ++d) c += frame[(size_t)r * 6 + d] >= 0; ++nframe[c]; }
    if (const char* dump = getenv("GCGE_REORDER_DUMP")) { FILE* f = fopen(dump, "wb"); if (f) { fwrite(cx.data(), 4, n, f); fwrite(cy.data(), 4, n, f); fwrite(cz.data(), 4, n, f); fwrite(frame.data(), 4, (size_t)n * 6, f); fclose(f); } }
    fprintf(stderr, "  box of the placed rows: x %d..%d, y %d..%d, z %d..%d; frames with 0..6 known directions: %ld %ld %ld %ld %ld %ld %ld\n", lo[0], hi[0], lo[1], hi[1],
            lo[2], hi[2], nframe[0], nframe[1], nframe[2], nframe[3], nframe[4], nframe[5], nframe[6]);
  }
  if (nx * ny * nz > 4L * n || nx * ny * nz > INT32_MAX) GIVE_UP;
  std::vector<int> occ((size_t)(nx * ny * nz), -1);
  auto at = [&](long x, long y, long z) -> long { return x + nx * (y + ny * z); };
  for (int r = 0; r < n; ++r) if (cx[r] != INT32_MIN) {
    const long p = at(cx[r] - lo[0], cy[r] - lo[1], cz[r] - lo[2]);
    if (occ[p] >= 0) GIVE_UP;
    occ[p] = r; box_of_row[r] = (int)p;
  }
  // 5. rows the fill did not reach: by their exact stencil entries to placed rows (two lines fix a point), a few rounds
  std::map<uint64_t, int> dist_of;                                          // coefficient value -> distance k
  for (size_t k = 0; k < coef.size(); ++k) dist_of[dbits(coef[k])] = (int)k + 1;
  for (int round = 0; round < 3; ++round) {
    long newly = 0;
    for (int r = 0; r < n; ++r) {
      if (cx[r] != INT32_MIN) continue;
      std::map<long, int> votes;                                            // candidate position -> supporting entries
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int q = colidx[k];
        if (q >= n || q == r || cx[q] == INT32_MIN) continue;
        auto it = dist_of.find(dbits(val[k]));
        if (it == dist_of.end()) continue;
        const int kk = it->second;
        for (int d = 0; d < 6; ++d) {
          const long x = cx[q] - lo[0] + (long)kk * dx[d], y = cy[q] - lo[1] + (long)kk * dy[d], z = cz[q] - lo[2] + (long)kk * dz[d];
          if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz || occ[at(x, y, z)] >= 0) continue;
          ++votes[at(x, y, z)];
        }
      }
      long best = -1; int bv = 0, second = 0;
      for (auto& kv : votes) { if (kv.second > bv) { second = bv; bv = kv.second; best = kv.first; } else if (kv.second > second) second = kv.second; }
      if (best >= 0 && bv >= 2 && bv > second) {
        occ[best] = r; box_of_row[r] = (int)best;
        cx[r] = (int)(best % nx) + lo[0]; cy[r] = (int)((best / nx) % ny) + lo[1]; cz[r] = (int)(best / (nx * ny)) + lo[2];
        ++newly;
      }
    }
    if (newly == 0) break;
  }
  // 6. whatever is left takes the free positions in ascending order (needs as many free positions as rows: a complete box or a mask with room)
  {
    long p = 0;
    for (int r = 0; r < n; ++r) {
      if (cx[r] != INT32_MIN) continue;
      while (p < nx * ny * nz && occ[p] >= 0) ++p;
      if (p >= nx * ny * nz) GIVE_UP;
      occ[p] = r; box_of_row[r] = (int)p; cx[r] = 0;
    }
  }
  dims[0] = (int)nx; dims[1] = (int)ny; dims[2] = (int)nz;
  return placed;
}
