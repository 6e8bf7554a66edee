// app_hip — the MI355X back-end behind GCGE's operator table.
//
// Counterpart of the reference's app/app_ccs.c (sparse matrix + block of vectors)
// with every O(n) operand resident in HBM: OPS_HIP_Set fills the same slots
// OPS_CCS_Set does (app_ccs.c:213-249), each slot keeps the contract of SURVEY.md
// Appendix A, and small dense results are returned to HOST pointers, complete on
// return (the solver layers are synchronous).  Handles are opaque:
//   matrix       GCGE_HIP_MAT  (CSR + pad-8 copy on the device)      <- CCSMAT   (app_ccs.h:20-24)
//   multivector  GcgeHipMV     (row-major n x ld block on the device) <- LAPACKVEC (app_lapack.h:17-20)
#include <hip/hip_runtime.h>
#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <unordered_map>
#include <algorithm>

#include "gcge_hip.h"
#include "gcge_solver.h"
#include "gcge_hip_internal.h"

extern "C" {
int gcge_hip_pad8_spmm_dot(int nrows, const int* d_orp, const int* d_pcol, const double* d_pval, const double* d_x,
                           long ldx, double* d_y, long ldy, int ncols, double* d_dots, void* stream, long x_own_row0);
void gcge_hip_spmm_pad8_auto(double avg_octets_per_row);
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
int gcge_hip_dense_spmm(const void* dm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream, int which);
int gcge_hip_tile_spmm(const void* tm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream);
void* gcge_hip_star_build(int nrows, int ncols_local, long row_begin, long nglobal, const int* ghost, const int* rowptr, const int* colidx, const double* val,
                          const int** rem_rowptr, const int** rem_col, const double** rem_val);
void gcge_hip_star_release_remainder(void);
void gcge_hip_star_free(void* sm);
void gcge_hip_star_stats(const void* sm, long* out);
int gcge_hip_star_spmm(const void* sm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, void* stream);
const unsigned char* gcge_hip_star_host_mask(void);
int gcge_hip_star_spmm_dots(const void* sm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, double* d_dots, void* stream);
int gcge_hip_star_spmm_part(const void* sm, const double* d_x, long ldx, double* d_y, long ldy, int ncols, double* d_dots, void* stream, int part);
int gcge_hip_star_interior(const void* sm, int* ilo, int* ihi);
int gcge_hip_star_coldots2_rows(int nlist, const int* d_list, const double* d_x, long ldx, const double* d_y, long ldy, int m, double* d_out, void* stream);
const int* gcge_hip_dense_row_list(const void* dm, int* nlisted);
void* gcge_hip_dense_build_rows(int nrows, int ncols_local, const int* rowptr, const int* colidx, const double* val, const unsigned char* not_listed);
int gcge_hip_pattern_spmm(int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt, long span, long span2, const double* d_x, long ldx,
                          double* d_y, long ldy, int ncols, double* d_dots, double* d_dots_yy, void* stream);
int gcge_hip_colscale(int nrows, double* d_y, long ldy, int m, const double* d_s, void* stream);
int gcge_hip_panel_dot1(int nrows, const double* d_x, long ldx, int k, const double* d_y, long ldy, double* d_out, void* stream);
int gcge_hip_rank1_update(int nrows, const double* d_x, long ldx, const double* d_c, const double* d_beta, double* d_y, long ldy, int m, void* stream);
int gcge_hip_colscale1(int nrows, double* d_y, long ldy, double s, void* stream);
int gcge_hip_mgs_step(int nrows, double* d_xk, long ld, double s, const double* d_c, int w, double* d_dots, void* stream);
int gcge_hip_fill_uniform(int nrows, long row_begin, long nglobal, double* d_y, long ldy, int c0, int m,
                          unsigned long long seed, void* stream);
int gcge_hip_colmajor_to_rowmajor(int nrows, int m, const double* d_src, long lds, double* d_dst, long ldd, void* stream);
int gcge_hip_rowmajor_to_colmajor(int nrows, int m, const double* d_src, long lds, double* d_dst, long ldd, void* stream);
}

extern "C" int gcge_hip_pattern_spmm_vals(int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt, long span, long span2,
                                          const double* d_x, long ldx, double* d_y, long ldy, int ncols, double* d_dots, double* d_dots_yy,
                                          void* stream, long near, const double* d_rowval);
extern "C" int gcge_hip_pattern_cg_vals(int mode, int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                   long span, long span2, const double* d_x, long ldx, double* d_r, long ldr, double* d_pnew,
                                   long ldp, int ncols, const double* d_alpha, const double* d_beta, const int* d_flag,
                                   double* d_dots, double* d_dots_yy, void* stream, const double* d_b, long ldb, long near,
                                   const double* d_rowval);
extern "C" int gcge_hip_pattern_spmm_near(int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt, long span, long span2,
                                          const double* d_x, long ldx, double* d_y, long ldy, int ncols, double* d_dots, double* d_dots_yy,
                                          void* stream, long near);
extern "C" int gcge_hip_pattern_cg_near(int mode, int nrows, const unsigned short* d_pid, const void* d_tab, int npat, int lt,
                                   long span, long span2, const double* d_x, long ldx, double* d_r, long ldr, double* d_pnew,
                                   long ldp, int ncols, const double* d_alpha, const double* d_beta, const int* d_flag,
                                   double* d_dots, double* d_dots_yy, void* stream, const double* d_b, long ldb, long near);


__global__ __launch_bounds__(256) void halo_pack(int nsend, const int* __restrict__ rows, const double* __restrict__ x,
    long ldx, int m, double* __restrict__ buf) {
  const long total = (long)nsend * m;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long i = idx / m; const int j = (int)(idx - i * m);
    buf[idx] = x[(long)rows[i] * ldx + j];
  }
}
__global__ __launch_bounds__(256) void halo_unpack(int nghost, const double* __restrict__ buf, int m, double* __restrict__ xg,
    long ldx) {
  const long total = (long)nghost * m;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long g = idx / m; const int j = (int)(idx - g * m);
    xg[g * ldx + j] = buf[idx];
  }
}

struct GcgeHipMV {
  double* d;
  size_t bytes;   // size of the allocation behind d
  long ld;
  int nrows, nrows_alloc, ncols;
  const GCGE_HIP_MAT_* mat;   // shape donor (row partition)
  GcgePerm* perm;             // the row order of the matrix this block was created for (NULL / identity: the caller's order); see mat_upload.hip
  // column-wise Gram-Schmidt over the slots (see "one sweep per column" below): the state lives in the block it belongs to
  int pend_col; double pend_fac;                       // a scaling of column pend_col held back (pend_col < 0: none)
  int spec_c0, spec_c1; unsigned long spec_epoch;      // Gram column of [spec_c0, spec_c1) computed on the way by the call of epoch spec_epoch
  std::vector<double>* spec_dots;                      // (NULL: none)
};

static hipStream_t g_stream = nullptr;
static int g_inited = 0;
static double* g_stage_d = nullptr; static size_t g_stage_d_len = 0;   // device staging (doubles)
static double* g_stage_h = nullptr; static size_t g_stage_h_len = 0;   // pinned host staging
static int g_spmm_path = 0;   // 0: automatic (pattern > dense blocks + remainder > X tiles > pad-8 > CSR), 2: no pattern path, 3: pad-8 / CSR only, 4: no block form (tile form if present)
extern "C" void gcge_hip_set_spmm_path(int path) { g_spmm_path = path; }
static int g_offset_patterns = 1;   // 1: stencils with row-dependent coefficients take the pattern kernels with streamed values
extern "C" void gcge_hip_set_offset_patterns(int on) { g_offset_patterns = on; }
static int g_rand_mode = 0; static unsigned long long g_rand_seed = 0x5DEECE66Dull;

// ------------------------------------------------------------------ per-slot wall time (measurement aid, off by default)
// gcge_hip_slot_timing(1): every slot call is followed by a stream synchronisation and its wall time is added to a bucket
// (slot, width class of the column range it worked on); gcge_hip_slot_timing_report prints the buckets.  Used to see where
// a solver stack that was NOT written for this layout spends its time (tests/refstack_on_hip.py).
#include <chrono>
#include <map>
#include <tuple>
#include <string>
static int g_slot_timing = 0;
static std::map<std::string, std::pair<long, double>> g_slot_time;
struct SlotTimer {
  const char* name; int cols; std::chrono::steady_clock::time_point t0; bool on;
  SlotTimer(const char* n, int c) : name(n), cols(c), on(g_slot_timing != 0) { if (on) t0 = std::chrono::steady_clock::now(); }
  ~SlotTimer() {
    if (!on) return;
    hipStreamSynchronize(g_stream);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const char* cls = cols <= 1 ? "1 col" : cols <= 8 ? "2-8 cols" : cols <= 32 ? "9-32 cols" : "> 32 cols";
    auto& b = g_slot_time[std::string(name) + " [" + cls + "]"];
    b.first += 1; b.second += dt;
  }
};
extern "C" void gcge_hip_slot_timing(int on) { g_slot_timing = on; if (on) g_slot_time.clear(); }
extern "C" int gcge_hip_slot_timing_report(char* buf, int len) {
  std::string out;
  for (auto& kv : g_slot_time) {
    char line[256];
    snprintf(line, sizeof line, "%-44s calls %8ld  %9.3f s  %8.3f ms/call\n", kv.first.c_str(), kv.second.first, kv.second.second,
             1e3 * kv.second.second / (double)kv.second.first);
    out += line;
  }
  if (buf != nullptr && len > 0) { strncpy(buf, out.c_str(), (size_t)len - 1); buf[len - 1] = 0; }
  return (int)out.size();
}

// ------------------------------------------------------------------ in-solve rate of the dense kernels per shape (measurement aid)
// gcge_hip_dense_profile(1): every Gram (K2) and panel update (K3) launched by the slots is bracketed by two HIP events on the
// back-end's stream (no synchronisation); gcge_hip_dense_profile_report sums them per (kernel, k, m): calls, average time and
// 2 n k m flop / time — the TF a solve actually sees for each shape, not a stand-alone benchmark's.
struct DenseEvent { hipEvent_t e0, e1; int kind, k, m; long n; double bytes; };   // kind 0: Gram, 1: panel update; bytes the launch must move
static std::vector<DenseEvent> g_dense_prof;
static std::map<std::tuple<int, int, int>, std::tuple<long, double, double, double>> g_dense_acc;   // (kind, k, m) -> calls, ms, flop, bytes: events already folded
static int g_dense_prof_on = 0;
// events are recycled (~1e5 launches per bench must not create and destroy 2e5 events inside the timed region)
static std::vector<hipEvent_t> g_dense_ev_pool;
static hipEvent_t dense_ev_get() {
  if (!g_dense_ev_pool.empty()) { hipEvent_t e = g_dense_ev_pool.back(); g_dense_ev_pool.pop_back(); return e; }
  hipEvent_t e; GCGE_HIP_CHECK(hipEventCreate(&e)); return e;
}
// fold the recorded intervals into the per-shape sums (all of them: the caller has synchronised; otherwise the completed front)
static void dense_prof_fold(bool all) {
  size_t i = 0;
  for (; i < g_dense_prof.size(); ++i) {
    DenseEvent& e = g_dense_prof[i];
    if (!all && hipEventQuery(e.e1) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.e0, e.e1) == hipSuccess) {
      auto& a = g_dense_acc[std::make_tuple(e.kind, e.k, e.m)];
      std::get<0>(a) += 1; std::get<1>(a) += ms; std::get<2>(a) += 2.0 * (double)e.n * e.k * e.m; std::get<3>(a) += e.bytes;
    }
    g_dense_ev_pool.push_back(e.e0); g_dense_ev_pool.push_back(e.e1);
  }
  g_dense_prof.erase(g_dense_prof.begin(), g_dense_prof.begin() + (long)i);
}
extern "C" void gcge_hip_dense_profile(int on) {
  for (auto& e : g_dense_prof) { g_dense_ev_pool.push_back(e.e0); g_dense_ev_pool.push_back(e.e1); }
  g_dense_prof.clear();
  g_dense_acc.clear();
  g_dense_prof_on = on;
  if (on) while (g_dense_ev_pool.size() < 8192) { hipEvent_t e; GCGE_HIP_CHECK(hipEventCreate(&e)); g_dense_ev_pool.push_back(e); }   // before anything is timed
}
struct DenseProfScope {
  DenseEvent ev; bool on;
  DenseProfScope(int kind, long n, int k, int m, double bytes) : on(g_dense_prof_on != 0) {
    if (!on) return;
    ev.kind = kind; ev.k = k; ev.m = m; ev.n = n; ev.bytes = bytes;
    ev.e0 = dense_ev_get(); ev.e1 = dense_ev_get();
    GCGE_HIP_CHECK(hipEventRecord(ev.e0, g_stream));
  }
  ~DenseProfScope() {
    if (!on) return;
    hipEventRecord(ev.e1, g_stream); g_dense_prof.push_back(ev);
    if (g_dense_prof.size() >= 4096) dense_prof_fold(false);          // (a bench of 20 solves brackets ~1e5 launches)
  }
};
// the same sums as numbers: rows of 7 doubles (kind 0 Gram / 1 panel update, k, m, calls, milliseconds in all, flop in all, bytes the
// launches had to move in all: both operands once, + the panel read where beta != 0 and it is not updated in place), at most max_rows
// of them, largest total time first; returns the number of shapes seen
extern "C" int gcge_hip_dense_profile_shapes(double* out, int max_rows) {
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  dense_prof_fold(true);
  std::vector<std::pair<double, std::tuple<int, int, int>>> order;
  for (auto& kv : g_dense_acc) order.emplace_back(-std::get<1>(kv.second), kv.first);
  std::sort(order.begin(), order.end());
  int r = 0;
  for (auto& o : order) {
    if (r >= max_rows) break;
    const auto& a = g_dense_acc[o.second];
    double* q = out + 7 * (size_t)r++;
    q[0] = std::get<0>(o.second); q[1] = std::get<1>(o.second); q[2] = std::get<2>(o.second);
    q[3] = (double)std::get<0>(a); q[4] = std::get<1>(a); q[5] = std::get<2>(a); q[6] = std::get<3>(a);
  }
  return (int)order.size();
}
extern "C" int gcge_hip_dense_profile_report(char* buf, int len) {
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  dense_prof_fold(true);
  std::string out;
  for (auto& kv : g_dense_acc) {
    char line[256];
    const long calls = std::get<0>(kv.second); const double ms = std::get<1>(kv.second), fl = std::get<2>(kv.second);
    snprintf(line, sizeof line, "%-13s k = %4d  m = %4d  calls %6ld  %9.3f ms per call  %6.1f TF  (%.1f ms in all)\n",
             std::get<0>(kv.first) == 0 ? "Gram" : "panel update", std::get<1>(kv.first), std::get<2>(kv.first), calls, ms / calls,
             ms > 0 ? fl / (ms * 1e-3) * 1e-12 : 0.0, ms);
    out += line;
  }
  if (buf != nullptr && len > 0) { strncpy(buf, out.c_str(), (size_t)len - 1); buf[len - 1] = 0; }
  return (int)out.size();
}

// ------------------------------------------------------------------ column-wise Gram-Schmidt over the slots, one sweep per column
// The reference's OrthSelf (src/ops_orth.c:45-118) issues per column k: QtAP (k x 1 Gram of the remaining panel with x_k), a
// scaling of x_k, a rank-1 update of the columns behind it — three strided sweeps over the panel.  The back-end fuses them
// WITHOUT changing what is computed: the scaling is held back until the next call; if that call is the rank-1 update from
// that very column, one kernel scales x_k, updates the panel and accumulates the Gram column of x_{k+1} that the next
// QtAP will ask for — same operands, same products, and the SAME order of the row sums as the separate kernel
// (vec_kernels.hip: mgs_step_kernel mirrors panel_dot1_partial), so the fused path is bit-identical to the unfused one; a
// Gram column that differs in the last bit is enough to change which noise-level columns a degenerate block drops.
// Anything else flushes the held-back scaling first.  The speculative Gram column is served only to the IMMEDIATELY following data call (epoch
// check) on the same block and column range; every entry point that touches block data goes through enter().
static unsigned long g_epoch = 0;
static GcgeHipMV* g_pend_owner = nullptr;   // the one block with a held-back scaling (its pend_col >= 0), or NULL
static int g_mgs_fuse = 1;
extern "C" void gcge_hip_set_mgs_fusion(int on) { g_mgs_fuse = on; }
static long g_mgs_fused_steps = 0, g_mgs_spec_hits = 0;
extern "C" void gcge_hip_mgs_fusion_stats(long* fused_steps, long* served_grams) { if (fused_steps) *fused_steps = g_mgs_fused_steps; if (served_grams) *served_grams = g_mgs_spec_hits; }
static void flush_pending();
static inline void enter(bool keep_pending = false) { ++g_epoch; if (!keep_pending && g_pend_owner != nullptr) flush_pending(); }
extern "C" void gcge_hip_flush_pending(void) { enter(); }   /* for other translation units that take device pointers of blocks */
// the same without counting as a call (the epoch decides whether a speculative Gram column is still the latest word on its block):
// called at the top of every EXPORTED raw kernel that takes device pointers (gcge_hip.h "raw kernels") — a caller may hold a
// pointer from before the scaling was held back
extern "C" void gcge_hip_apply_pending(void) { if (g_pend_owner != nullptr) flush_pending(); }

static double* stage_d(size_t len) {
  if (len > g_stage_d_len) {
    if (g_stage_d) GCGE_HIP_CHECK(hipFree(g_stage_d));
    g_stage_d_len = len + len / 4 + 4096;
    GCGE_HIP_CHECK(hipMalloc(&g_stage_d, g_stage_d_len * sizeof(double)));
  }
  return g_stage_d;
}
static double* g_stage_d2 = nullptr; static size_t g_stage_d2_len = 0;   // second device staging (split products)
static double* stage_d2(size_t len) {
  if (len > g_stage_d2_len) {
    if (g_stage_d2) GCGE_HIP_CHECK(hipFree(g_stage_d2));
    g_stage_d2_len = len * 2 + 1024;
    GCGE_HIP_CHECK(hipMalloc(&g_stage_d2, g_stage_d2_len * sizeof(double)));
  }
  return g_stage_d2;
}
static double* stage_h(size_t len) {
  if (len > g_stage_h_len) {
    if (g_stage_h) GCGE_HIP_CHECK(hipHostFree(g_stage_h));
    g_stage_h_len = len + len / 4 + 4096;
    GCGE_HIP_CHECK(hipHostMalloc(&g_stage_h, g_stage_h_len * sizeof(double)));
  }
  return g_stage_h;
}

extern "C" int gcge_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int gcge_hip_init(int device) {
  if (g_inited) return 0;
  if (gcge_hip_device_count() <= 0) {
    fprintf(stderr, "gcge_hip_init: no HIP device visible — the HIP back-end has no CPU fallback\n");
    return -1;
  }
  if (device >= 0) GCGE_HIP_CHECK(hipSetDevice(device));
  g_stream = nullptr;   // the legacy default stream: ordered with torch's current stream and hipMemcpy
  g_inited = 1;
  return 0;
}
extern "C" void gcge_hip_pool_release(void);
extern "C" void gcge_hip_finalize(void) {
  gcge_hip_pool_release();
  if (g_stage_d) hipFree(g_stage_d);
  if (g_stage_h) hipHostFree(g_stage_h);
  g_stage_d = nullptr; g_stage_h = nullptr; g_stage_d_len = g_stage_h_len = 0; g_inited = 0;
}
extern "C" void gcge_hip_sync(void) { GCGE_HIP_CHECK(hipStreamSynchronize(g_stream)); }
extern "C" void* gcge_hip_stream(void) { return (void*)g_stream; }
extern "C" void gcge_hip_set_random_mode(int mode, unsigned long long seed) { g_rand_mode = mode; g_rand_seed = seed; }

// ------------------------------------------------------------------ SpMM launch profiling
// HIP events around every K1 launch on the launch stream (bench.py: roofline.achieved =
// algorithmic bytes / average launch duration, measured live inside the timed region).
struct SpmmEvent { hipEvent_t e0, e1; int m; double bytes; int kind; long rows; };   // rows: local rows of the matrix (a solver may run the same kernels on several: the levels of a multigrid cycle)   // kind 0: product (plain or with the column sums), 2 / 3: CG passes
static std::vector<SpmmEvent> g_prof;
static int g_prof_on = 0;
extern "C" void gcge_hip_profile_enable(int on) {
  for (auto& e : g_prof) { hipEventDestroy(e.e0); hipEventDestroy(e.e1); }
  g_prof.clear();
  g_prof_on = on;
}
// sums over the recorded launches with exactly `ncols` columns (0: all); returns the count
// kind 0: MatDotMultiVec products (plain or with the column sums); 2 / 3: first / second pass of the fused CG
extern "C" long gcge_hip_profile_kind(int kind, int ncols, double* total_ms, double* total_alg_bytes);
extern "C" long gcge_hip_profile_spmm(int ncols, double* total_ms, double* total_alg_bytes) {
  return gcge_hip_profile_kind(0, ncols, total_ms, total_alg_bytes);
}
extern "C" long gcge_hip_profile_kind_rows(int kind, int ncols, long nrows, double* total_ms, double* total_alg_bytes);
extern "C" long gcge_hip_profile_kind(int kind, int ncols, double* total_ms, double* total_alg_bytes) {
  return gcge_hip_profile_kind_rows(kind, ncols, 0, total_ms, total_alg_bytes);
}
// ... restricted to the launches on matrices of `nrows` local rows (0: all).  With BlockAMG as the solver the fused CG runs the same
// kernels on every level of the hierarchy; a roofline figure belongs to ONE problem size (bench.py: the finest level).
extern "C" long gcge_hip_profile_kind_rows(int kind, int ncols, long nrows, double* total_ms, double* total_alg_bytes) {
  long cnt = 0; double ms = 0.0, by = 0.0;
  GCGE_HIP_CHECK(hipDeviceSynchronize());
  for (auto& e : g_prof) {
    if (e.kind != kind || (ncols > 0 && e.m != ncols) || (nrows > 0 && e.rows != nrows)) continue;
    float t = 0.f;
    GCGE_HIP_CHECK(hipEventElapsedTime(&t, e.e0, e.e1));
    ms += t; by += e.bytes; ++cnt;
  }
  if (total_ms) *total_ms = ms;
  if (total_alg_bytes) *total_alg_bytes = by;
  return cnt;
}

// (matrix handle: upload, analysis into the K1 forms, halo plan, destruction — csrc/hip/mat_upload.hip)
extern "C" int gcge_hip_spmm_path_get(void) { return g_spmm_path; }
extern "C" int gcge_hip_offset_patterns_get(void) { return g_offset_patterns; }

// ------------------------------------------------------------------ device buffer pool
// hipMalloc / hipFree of the multi-GB blocks cost 0.25-0.3 s each on this stack (page-table set-up; hipFree also
// synchronises the device): ~1.9 s of a 21 s solve.  Freed blocks are kept by exact size and handed out again —
// every use of the back-end happens on one stream, so a recycled block is ordered behind the kernels that last
// touched it.  The reference allocates its work blocks per solve as well (test_eig_sol_gcg.c:66-92); a second
// solve of the same shape then allocates nothing.  gcge_hip_pool_release() returns everything to the driver.
static std::unordered_map<size_t, std::vector<void*>> g_pool;
static size_t g_pool_bytes = 0;
static int g_pool_on = 1;
extern "C" void gcge_hip_pool_release(void) {
  for (auto& kv : g_pool) for (void* q : kv.second) hipFree(q);
  g_pool.clear(); g_pool_bytes = 0;
}
extern "C" void gcge_hip_pool_enable(int on) { g_pool_on = on; if (!on) gcge_hip_pool_release(); }
// bytes the pool holds at the moment (free for MultiVecCreate*, but "used" in hipMemGetInfo)
extern "C" size_t gcge_hip_pool_cached_bytes(void) { return g_pool_bytes; }
static void* pool_alloc(size_t bytes) {
  auto it = g_pool.find(bytes);
  if (it != g_pool.end() && !it->second.empty()) {
    void* q = it->second.back(); it->second.pop_back(); g_pool_bytes -= bytes;
    return q;
  }
  void* q = nullptr;
  if (hipMalloc(&q, bytes) != hipSuccess) {   // out of memory: give the cached blocks back and try once more
    (void)hipGetLastError();
    gcge_hip_pool_release();
    GCGE_HIP_CHECK(hipMalloc(&q, bytes));
  }
  return q;
}
static void pool_free(void* q, size_t bytes) {
  if (!g_pool_on || bytes < ((size_t)1 << 20)) {   // small blocks are not worth tracking
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream)); hipFree(q); return;
  }
  g_pool[bytes].push_back(q); g_pool_bytes += bytes;
}

// ------------------------------------------------------------------ multivector
static inline const GcgePerm* real_perm(const GcgePerm* p) { return (p != nullptr && !p->identity) ? p : nullptr; }
static GcgeHipMV* mv_new(int nrows, int nghost, int ncols, const GCGE_HIP_MAT_* mat, GcgePerm* perm) {
  GcgeHipMV* v = (GcgeHipMV*)calloc(1, sizeof(GcgeHipMV));
  v->nrows = nrows; v->nrows_alloc = nrows + nghost; v->ncols = ncols; v->mat = mat; v->pend_col = -1;
  v->perm = real_perm(perm) != nullptr ? gcge_hip_perm_acquire(perm) : nullptr;   // (the caller's order needs no record: only matrices pin "as given" for their size)
  v->ld = ((long)(ncols > 0 ? ncols : 1) + 7) / 8 * 8;
  const size_t bytes = (size_t)v->nrows_alloc * v->ld * sizeof(double);
  v->bytes = bytes ? bytes : 8;
  v->d = (double*)pool_alloc(v->bytes);
  GCGE_HIP_CHECK(hipMemsetAsync(v->d, 0, bytes, g_stream));   // zero-filled like app_ccs.c:47
  return v;
}
static void HIP_MultiVecCreateByMat(void*** mv, int num_vec, void* mat, struct OPS_* ops) {
  const GCGE_HIP_MAT_* A = (const GCGE_HIP_MAT_*)mat;
  if (A->rect_ncols > 0) { *mv = (void**)mv_new(A->rect_ncols, 0, num_vec, nullptr, nullptr); return; }   // app_ccs.c:43: rows = the matrix's COLUMNS
  *mv = (void**)mv_new(A->nrows, A->nghost, num_vec, A, A->perm);
}
static void HIP_MultiVecCreateByMultiVec(void*** mv, int num_vec, void** src, struct OPS_* ops) {
  const GcgeHipMV* s = (const GcgeHipMV*)src;
  *mv = (void**)mv_new(s->nrows, s->nrows_alloc - s->nrows, num_vec, s->mat, s->perm);
}
static void HIP_MultiVecDestroy(void*** mv, int num_vec, struct OPS_* ops) {
  GcgeHipMV* v = *(GcgeHipMV**)mv;
  enter();
  if (v) { delete v->spec_dots; pool_free(v->d, v->bytes); gcge_hip_perm_release(v->perm); free(v); }
  *mv = nullptr;
}
static void flush_pending() {
  GcgeHipMV* v = g_pend_owner;
  g_pend_owner = nullptr;
  if (v != nullptr && v->pend_col >= 0) {
    const int col = v->pend_col;
    v->pend_col = -1;
    gcge_hip_colscale1(v->nrows, v->d + col, v->ld, v->pend_fac, g_stream);
  }
}
extern "C" int gcge_hip_mv_nrows(void** mv) { return ((GcgeHipMV*)mv)->nrows; }
// id of the row order a block lives in (0: the caller's): scratch blocks kept between calls belong to ONE order (block_pcg.hip)
extern "C" unsigned gcge_hip_mv_row_order_id(void** mv) { const GcgePerm* p = real_perm(((GcgeHipMV*)mv)->perm); return p ? p->id : 0u; }
extern "C" int gcge_hip_mv_ncols(void** mv) { return ((GcgeHipMV*)mv)->ncols; }
extern "C" double* gcge_hip_mv_device_ptr(void** mv, long* ld) {
  enter();
  GcgeHipMV* v = (GcgeHipMV*)mv; if (ld) *ld = v->ld; return v->d;
}

// host column-major  <->  device row-major, in panels so the staging stays bounded
extern "C" void gcge_hip_mv_from_host(void** mv, int c0, int c1, const double* host, long ldh) {
  enter();
  GcgeHipMV* v = (GcgeHipMV*)mv;
  const int n = v->nrows;
  GCGE_REQUIRE(c0 >= 0 && c1 <= v->ncols && ldh >= n, "gcge_hip_mv_from_host: ranges");
  const int panel = 32;
  const GcgePerm* P = real_perm(v->perm);          // the block lives in the back-end's own row order: device row i = the caller's row perm[i]
  std::vector<double> tmp;
  if (P != nullptr) tmp.resize((size_t)n * panel);
  for (int c = c0; c < c1; c += panel) {
    const int m = (c1 - c < panel) ? c1 - c : panel;
    double* st = stage_d((size_t)n * m);
    const double* src = host + (size_t)(c - c0) * ldh; size_t lds_ = (size_t)ldh;
    if (P != nullptr) {
      for (int j = 0; j < m; ++j) { const double* hj = host + (size_t)(c - c0 + j) * ldh; double* tj = tmp.data() + (size_t)j * n; for (int i = 0; i < n; ++i) tj[i] = hj[P->perm[i]]; }
      src = tmp.data(); lds_ = (size_t)n;
    }
    GCGE_HIP_CHECK(hipMemcpy2DAsync(st, (size_t)n * sizeof(double), src, lds_ * sizeof(double), (size_t)n * sizeof(double), m,
                                    hipMemcpyHostToDevice, g_stream));
    gcge_hip_colmajor_to_rowmajor(n, m, st, n, v->d + c, v->ld, g_stream);
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  }
}
extern "C" void gcge_hip_mv_to_host(void** mv, int c0, int c1, double* host, long ldh) {
  enter();
  GcgeHipMV* v = (GcgeHipMV*)mv;
  const int n = v->nrows;
  GCGE_REQUIRE(c0 >= 0 && c1 <= v->ncols && ldh >= n, "gcge_hip_mv_to_host: ranges");
  const int panel = 32;
  const GcgePerm* P = real_perm(v->perm);
  std::vector<double> tmp;
  if (P != nullptr) tmp.resize((size_t)n * panel);
  for (int c = c0; c < c1; c += panel) {
    const int m = (c1 - c < panel) ? c1 - c : panel;
    double* st = stage_d((size_t)n * m);
    gcge_hip_rowmajor_to_colmajor(n, m, v->d + c, v->ld, st, n, g_stream);
    double* dst = P != nullptr ? tmp.data() : host + (size_t)(c - c0) * ldh;
    GCGE_HIP_CHECK(hipMemcpy2DAsync(dst, (P != nullptr ? (size_t)n : (size_t)ldh) * sizeof(double), st,
                                    (size_t)n * sizeof(double), (size_t)n * sizeof(double), m,
                                    hipMemcpyDeviceToHost, g_stream));
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
    if (P != nullptr)
      for (int j = 0; j < m; ++j) { double* hj = host + (size_t)(c - c0 + j) * ldh; const double* tj = tmp.data() + (size_t)j * n; for (int i = 0; i < n; ++i) hj[P->perm[i]] = tj[i]; }
  }
}

static void HIP_MultiVecView(void** x, int start, int end, struct OPS_* ops) {
  GcgeHipMV* v = (GcgeHipMV*)x;
  const int m = end - start;
  if (m <= 0) return;
  std::vector<double> h((size_t)v->nrows * m);
  gcge_hip_mv_to_host(x, start, end, h.data(), v->nrows);
  for (int r = 0; r < v->nrows; ++r) {
    for (int c = 0; c < m; ++c) ops->Printf("%6.4e\t", h[(size_t)c * v->nrows + r]);
    ops->Printf("\n");
  }
}

// app_lapack.c:322-333 — mode 0: the reference's rand() stream, column by column
static void HIP_MultiVecSetRandomValue(void** x, int start, int end, struct OPS_* ops) {
  enter();
  GcgeHipMV* v = (GcgeHipMV*)x;
  const int m = end - start;
  if (m <= 0) return;
  GCGE_REQUIRE(start >= 0 && end <= v->ncols, "MultiVecSetRandomValue: column range");
  if (g_rand_mode == 1) {
    const long rb = v->mat ? v->mat->row_begin : 0, ng = v->mat ? v->mat->nglobal : v->nrows;
    gcge_hip_fill_uniform(v->nrows, rb, ng, v->d, v->ld, start, m, g_rand_seed, g_stream);
    g_rand_seed += 0x9E3779B97F4A7C15ull;   // a later fill of the same columns differs
    return;
  }
  // Row slabs: every rank walks the GLOBAL rand() sequence (columns outer, global rows inner) and keeps its own
  // rows, so the start block is the single-rank one whatever the number of ranks (identical local streams would
  // make it periodic in the slab index).  O(n_global) draws per column: use mode 1 for large n.
  const long before = (v->mat && v->mat->nglobal > v->nrows) ? v->mat->row_begin : 0;
  const long after = (v->mat && v->mat->nglobal > v->nrows) ? (long)v->mat->nglobal - v->mat->row_begin - v->nrows : 0;
  const int panel = 16;
  std::vector<double> h((size_t)v->nrows * panel);
  for (int c = start; c < end; c += panel) {
    const int mm = (end - c < panel) ? end - c : panel;
    for (int j = 0; j < mm; ++j) {
      for (long k = 0; k < before; ++k) (void)rand();
      for (int r = 0; r < v->nrows; ++r) h[(size_t)j * v->nrows + r] = ((double)rand()) / ((double)RAND_MAX + 1);
      for (long k = 0; k < after; ++k) (void)rand();
    }
    gcge_hip_mv_from_host(x, c, c + mm, h.data(), v->nrows);
  }
}

// app_lapack.c:334-395
static void HIP_MultiVecAxpby(double alpha, void** x, double beta, void** y, int* start, int* end, struct OPS_* ops) {
  GcgeHipMV *vx = (GcgeHipMV*)x, *vy = (GcgeHipMV*)y;
  const int m = end[1] - start[1];
  enter();
  SlotTimer tm_(x ? "MultiVecAxpby" : "MultiVecAxpby (scale)", m);
  GCGE_REQUIRE(end[0] - start[0] == m, "MultiVecAxpby: equal column counts");
  if (m <= 0 || vy->nrows == 0) return;
  GCGE_REQUIRE(start[1] >= 0 && end[1] <= vy->ncols, "MultiVecAxpby: y column range");
  if (vx) {
    GCGE_REQUIRE(vx->nrows == vy->nrows, "MultiVecAxpby: equal row counts");
    GCGE_REQUIRE(start[0] >= 0 && end[0] <= vx->ncols, "MultiVecAxpby: x column range");
  }
  if (vx == nullptr && m == 1 && beta != 0.0) {   // one column scaled in place (q_k = x_k / r_kk of a column-wise Gram-Schmidt)
    if (g_mgs_fuse) { g_pend_owner = vy; vy->pend_col = start[1]; vy->pend_fac = beta; return; }   // held back: see enter()
    gcge_hip_colscale1(vy->nrows, vy->d + start[1], vy->ld, beta, g_stream);
    return;
  }
  gcge_hip_axpby(vy->nrows, alpha, vx ? vx->d + start[0] : nullptr, vx ? vx->ld : 0, beta, vy->d + start[1],
                 vy->ld, m, g_stream);
}

// app_lapack.c:463-534
static void HIP_MultiVecLinearComb(void** x, void** y, int is_vec, int* start, int* end, double* coef, int ldc,
                                   double* beta, int incb, struct OPS_* ops) {
  GcgeHipMV *vx = (GcgeHipMV*)x, *vy = (GcgeHipMV*)y;
  const int k = end[0] - start[0], m = end[1] - start[1];
  enter(true);   // a held-back scaling of x_k survives until it is known whether this is its rank-1 update
  SlotTimer tm_(k == 1 ? "MultiVecLinearComb (rank 1)" : "MultiVecLinearComb", m);
  {
    const bool step = g_mgs_fuse && k == 1 && vx == vy && vx != nullptr && coef != nullptr && start[1] == start[0] + 1 && m >= 1 && m <= 64 &&
                      beta != nullptr && incb == 0 && *beta == 1.0 && start[0] >= 0 && end[1] <= vy->ncols && vy->nrows > 0;
    double fac = 1.0;
    if (g_pend_owner != nullptr) {
      if (step && g_pend_owner == vx && vx->pend_col == start[0]) { fac = vx->pend_fac; vx->pend_col = -1; g_pend_owner = nullptr; }
      else flush_pending();
    }
    if (step) {
      // x_k *= fac, columns (k, k + m] += x_k coef^T, and the Gram column of x_{k+1} with the updated panel in the same sweep
      GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));   // staging buffers are reused
      double* hc = stage_h(2 * (size_t)m);
      for (int j = 0; j < m; ++j) hc[j] = coef[(size_t)j * ldc];
      double* dc = stage_d(2 * (size_t)m);
      GCGE_HIP_CHECK(hipMemcpyAsync(dc, hc, m * sizeof(double), hipMemcpyHostToDevice, g_stream));
      const int rc = gcge_hip_mgs_step(vy->nrows, vy->d + start[0], vy->ld, fac, dc, m, dc + m, g_stream);
      GCGE_REQUIRE(rc == 0, "MultiVecLinearComb: kernel launch");
      GCGE_HIP_CHECK(hipMemcpyAsync(hc + m, dc + m, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
      GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
      vy->spec_c0 = start[1]; vy->spec_c1 = end[1]; vy->spec_epoch = g_epoch;
      if (vy->spec_dots == nullptr) vy->spec_dots = new std::vector<double>();
      vy->spec_dots->assign(hc + m, hc + 2 * m);
      ++g_mgs_fused_steps;
      return;
    }
  }
  if (k == 0 || m == 0 || vy->nrows == 0) return;
  GCGE_REQUIRE(start[1] >= 0 && end[1] <= vy->ncols && m > 0, "MultiVecLinearComb: y column range");
  if (vx != nullptr && coef != nullptr) {
    GCGE_REQUIRE(vx->nrows == vy->nrows, "MultiVecLinearComb: equal row counts");
    GCGE_REQUIRE(start[0] >= 0 && end[0] <= vx->ncols && k > 0 && ldc >= k, "MultiVecLinearComb: x column range / ldc");
  }
  if (vx == nullptr || coef == nullptr) {       // scaling only: y_j *= beta_j
    if (beta == nullptr) return;
    if (incb == 0) { gcge_hip_axpby(vy->nrows, 0.0, nullptr, 0, *beta, vy->d + start[1], vy->ld, m, g_stream); return; }
    double* hb = stage_h(m);
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
    for (int j = 0; j < m; ++j) hb[j] = beta[(size_t)j * incb];
    double* db = stage_d(m);
    GCGE_HIP_CHECK(hipMemcpyAsync(db, hb, m * sizeof(double), hipMemcpyHostToDevice, g_stream));
    gcge_hip_colscale(vy->nrows, vy->d + start[1], vy->ld, m, db, g_stream);
    return;
  }
  if (k == 1) {   // rank-1 update y_j = x c_j + beta_j y_j: one sweep over the m contiguous columns of every row
    const size_t len = 2 * (size_t)m;
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));   // staging buffers are reused
    double* hc = stage_h(len);
    for (int j = 0; j < m; ++j) { hc[j] = coef[(size_t)j * ldc]; hc[m + j] = beta == nullptr ? 0.0 : (incb == 0 ? *beta : beta[(size_t)j * incb]); }
    double* dc = stage_d(len);
    GCGE_HIP_CHECK(hipMemcpyAsync(dc, hc, len * sizeof(double), hipMemcpyHostToDevice, g_stream));
    const bool unit_beta = beta != nullptr && incb == 0 && *beta == 1.0;
    int rc = gcge_hip_rank1_update(vy->nrows, vx->d + start[0], vx->ld, dc, unit_beta ? nullptr : dc + m, vy->d + start[1], vy->ld, m, g_stream);
    GCGE_REQUIRE(rc == 0, "MultiVecLinearComb: kernel launch");
    return;
  }
  // panels of <= 128 output columns; coefficient panel uploaded row-major (k x mp) [+ beta]
  for (int j0 = 0; j0 < m; j0 += 128) {
    const int mp = (m - j0 < 128) ? m - j0 : 128;
    const size_t len = (size_t)k * mp + mp;
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));   // staging buffers are reused
    double* hc = stage_h(len);
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < mp; ++j) hc[(size_t)i * mp + j] = coef[(size_t)(j0 + j) * ldc + i];
    if (beta != nullptr) for (int j = 0; j < mp; ++j) hc[(size_t)k * mp + j] = (incb == 0) ? *beta : beta[(size_t)(j0 + j) * incb];
    double* dc = stage_d(len);
    GCGE_HIP_CHECK(hipMemcpyAsync(dc, hc, len * sizeof(double), hipMemcpyHostToDevice, g_stream));
    // bytes a launch moves: X read, the panel written; the panel is READ as well where beta != 0 — unless it is updated in place
    // (y == x with the output columns inside the input range: the rows of X just read hold it)
    const bool inplace_ = vx == vy && start[1] + j0 >= start[0] && start[1] + j0 + mp <= end[0];
    DenseProfScope prof_(1, vy->nrows, k, mp, 8.0 * (double)vy->nrows * (k + mp + ((beta != nullptr && !inplace_) ? mp : 0)));
    int rc = gcge_hip_lincomb(vy->nrows, vx->d + start[0], vx->ld, k, dc, mp, beta ? dc + (size_t)k * mp : nullptr,
                              vy->d + start[1] + j0, vy->ld, g_stream);
    GCGE_REQUIRE(rc == 0, "MultiVecLinearComb: kernel launch");
  }
}

// app_lapack.c:299-313 -> DenseMatQtAP(matA == NULL) :64-183.  Result to HOST, column-major ldIP.
static void reduce_inner_prod(char nsd, int nr, int nc, double* ip, int ldIP);
// reduce != 0: the SUM OVER THE RANKS is returned; returns 1 if that sum was formed (on the device, before the one copy to the host:
// RCCL inside the back-end), 0 if the caller still has to reduce the host result through GCGE_COMM (another transport; the served
// Gram column of a fused Gram-Schmidt step)
static int local_inner_prod(char nsd, void** x, void** y, int is_vec, int* start, int* end, double* ip, int ldIP, struct OPS_* ops, int reduce = 0);
// the slot: the local rows' part — or, where the caller opted in (GCGE_SetLocalInnerProdReduces: stacks that sum through MPI only,
// the reference's BlockPCG src/ops_lin_sol.c:306-321), the sum over the ranks
static void HIP_MultiVecLocalInnerProd(char nsd, void** x, void** y, int is_vec, int* start, int* end, double* ip, int ldIP, struct OPS_* ops) {
  const int want = GCGE_GetLocalInnerProdReduces();
  if (!local_inner_prod(nsd, x, y, is_vec, start, end, ip, ldIP, ops, want) && want) reduce_inner_prod(nsd, end[0] - start[0], end[1] - start[1], ip, ldIP);
}
extern "C" void gcge_hip_local_inner_prod(char nsd, void** x, void** y, int* start, int* end, double* ip, int ldIP, struct OPS_* ops) {
  local_inner_prod(nsd, x, y, 0, start, end, ip, ldIP, ops);           // for block_pcg.hip: the local rows' part, never reduced
}
static int local_inner_prod(char nsd, void** x, void** y, int is_vec, int* start, int* end,
                                       double* ip, int ldIP, struct OPS_* ops, int reduce) {
  GcgeHipMV *vx = (GcgeHipMV*)x, *vy = (GcgeHipMV*)y;
  const int k = end[0] - start[0], m = end[1] - start[1];
  enter();
  SlotTimer tm_(m == 1 ? "MultiVecLocalInnerProd (k x 1)" : "MultiVecLocalInnerProd", m == 1 ? k : m);
  if (k <= 0 || m <= 0) return 0;
  // Round 5: with RCCL inside the back-end the partial Gram is summed over the ranks WHERE IT IS — on the device,
  // on the back-end's stream, in front of the one device-to-host copy — instead of host -> pinned -> device -> ncclAllReduce -> pinned
  // -> host behind it: one stream synchronisation per inner product instead of two, no second staging.  ~1 700 inner products per
  // solve are latency, not bytes.  Reference: MPI_Allreduce of the local result, src/ops_multi_vec.c:206-228.
  GCGE_COMM* comm_ = reduce ? GCGE_GetComm() : nullptr;
  const bool dev_reduce = comm_ != nullptr && gcge_hip_comm_is_native(comm_) && getenv("GCGE_GRAM_HOST_REDUCE") == nullptr;
  if (m == 1 && nsd != 'D' && vx == vy && vx->spec_dots != nullptr && !vx->spec_dots->empty() && vx->spec_epoch + 1 == g_epoch &&
      start[0] == vx->spec_c0 && start[1] == vx->spec_c0 && end[0] == vx->spec_c1 && (int)vx->spec_dots->size() == k && ldIP >= k) {
    // the Gram column the previous call (the rank-1 update of a Gram-Schmidt step ON THIS BLOCK) accumulated on its way: no
    // call of any kind has been made since (epoch), same block, same column range
    for (int i = 0; i < k; ++i) ip[i] = (*vx->spec_dots)[i];
    vx->spec_dots->clear(); ++g_mgs_spec_hits;
    return 0;
  }
  if (vx->spec_dots != nullptr) vx->spec_dots->clear();
  if (vy->spec_dots != nullptr) vy->spec_dots->clear();
  GCGE_REQUIRE(vx->nrows == vy->nrows, "MultiVecInnerProd: equal row counts");
  GCGE_REQUIRE(start[0] >= 0 && end[0] <= vx->ncols && start[1] >= 0 && end[1] <= vy->ncols, "MultiVecInnerProd: column ranges");
  GCGE_REQUIRE(nsd == 'D' ? ldIP >= 1 : ldIP >= k, "MultiVecInnerProd: ldIP");
  if (nsd == 'D') {
    GCGE_REQUIRE(k == m, "MultiVecInnerProd 'D': square");
    double* dd = stage_d(m);
    gcge_hip_coldots(vx->nrows, vx->d + start[0], vx->ld, vy->d + start[1], vy->ld, m, dd, g_stream);
    if (dev_reduce) gcge_hip_comm_allreduce_device(dd, m);
    double* hd = stage_h(m);
    GCGE_HIP_CHECK(hipMemcpyAsync(hd, dd, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
    for (int j = 0; j < m; ++j) ip[(size_t)ldIP * j] = hd[j];
    return dev_reduce ? 1 : 0;
  }
  double* dg = stage_d((size_t)k * m);
  if (m == 1) gcge_hip_panel_dot1(vx->nrows, vx->d + start[0], vx->ld, k, vy->d + start[1], vy->ld, dg, g_stream);   // panel . column
  else { DenseProfScope prof_(0, vx->nrows, k, m, 8.0 * (double)vx->nrows * ((vx == vy && start[0] == start[1] && k == m) ? k : k + m)); gcge_hip_gram(vx->nrows, vx->d + start[0], vx->ld, k, vy->d + start[1], vy->ld, m, dg, g_stream); }
  if (dev_reduce) gcge_hip_comm_allreduce_device(dg, k * m);      // (the whole k x m block: contiguous on the device whatever ldIP is)
  double* hg = stage_h((size_t)k * m);
  GCGE_HIP_CHECK(hipMemcpyAsync(hg, dg, (size_t)k * m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  if (nsd == 'S') {   // lower triangle is authoritative, mirrored (app_lapack.c:119-130)
    GCGE_REQUIRE(k == m, "MultiVecInnerProd 'S': square");
    for (int j = 0; j < m; ++j)
      for (int i = j; i < k; ++i) { const double v = hg[(size_t)i * m + j]; ip[(size_t)ldIP * j + i] = v; ip[(size_t)ldIP * i + j] = v; }
  } else {
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < k; ++i) ip[(size_t)ldIP * j + i] = hg[(size_t)i * m + j];
  }
  return dev_reduce ? 1 : 0;
}

// row-partitioned matrices: fetch the halo rows of X[:, c_begin : c_begin + m) from their owners
static void halo_fetch(GCGE_HIP_MAT_* A, GcgeHipMV* vx, int c_begin, int m) {
  if (A->nghost <= 0) return;
  const double* dx = vx->d + c_begin;
  GCGE_REQUIRE(A->exchange != nullptr && A->buf_cols > 0, "MatDotMultiVec: halo plan installed (gcge_hip_mat_set_halo)");
  for (int c0 = 0; c0 < m; c0 += A->buf_cols) {
    const int mc = (m - c0 < A->buf_cols) ? m - c0 : A->buf_cols;
    if (A->nsend > 0) {
      long tot = (long)A->nsend * mc, g = (tot + 255) / 256; if (g > 4096) g = 4096;
      hipLaunchKernelGGL(halo_pack, dim3((unsigned)g), dim3(256), 0, g_stream, A->nsend, A->d_send_rows, dx + c0, vx->ld, mc, A->sendbuf);
    }
    A->exchange(A->sendbuf, A->recvbuf, mc, A->exchange_ctx);
    long tot = (long)A->nghost * mc, g = (tot + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(halo_unpack, dim3((unsigned)g), dim3(256), 0, g_stream, A->nghost, A->recvbuf, mc,
                       vx->d + (long)A->nrows * vx->ld + c_begin + c0, vx->ld);
  }
}


__global__ void add3_kernel(double* __restrict__ dst, const double* __restrict__ a, const double* __restrict__ b,
                            const double* __restrict__ c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (a[i] + b[i]) + c[i];
}

// rows [r0, r1) of Y = A X, optionally with the column sums x.y (and y.y) over those rows (d_dots, d_yy: device, m each).
// want_fused: the caller asked for dots and the matrix/operands qualify for a fused kernel.
// cg != NULL: one of the two passes of a block-CG iteration instead of the product (pattern matrices only,
// gcge_hip_pattern_cg): mode 2 = the sums without storing Y, mode 3 = R / P update with the product recomputed.
struct CgPass { int mode; double* r; long ldr; double* pnew; long ldp; const double *alpha, *beta; const int* flag; const double* b; long ldb; };
static int spmm_rows(GCGE_HIP_MAT_* A, long r0, long r1, const double* dx, long ldx, double* dy, long ldy, int m,
                     double* d_dots, double* d_yy, const CgPass* cg = nullptr) {
  const int nr = (int)(r1 - r0);
  if (nr <= 0) {
    if (d_dots) GCGE_HIP_CHECK(hipMemsetAsync(d_dots, 0, m * sizeof(double), g_stream));
    if (d_yy) GCGE_HIP_CHECK(hipMemsetAsync(d_yy, 0, m * sizeof(double), g_stream));
    return 0;
  }
  if (cg != nullptr) {
    if (A->d_pid == nullptr || g_spmm_path != 0) return -1;
    return gcge_hip_pattern_cg_vals(cg->mode, nr, A->d_pid + r0, A->d_tab, A->npat, A->pat_lt, A->pat_span, A->pat_span2,
                               dx + r0 * ldx, ldx, cg->r ? cg->r + r0 * cg->ldr : nullptr, cg->ldr,
                               cg->pnew ? cg->pnew + r0 * cg->ldp : nullptr, cg->ldp, m, cg->alpha, cg->beta, cg->flag,
                               d_dots, d_yy, g_stream, cg->b ? cg->b + r0 * cg->ldb : nullptr, cg->ldb, A->pat_near,
                               A->d_rowval ? A->d_rowval + 8 * r0 : nullptr);
  }
  double* y = dy + r0 * ldy;
  int rc = -1;
  if (A->d_pid != nullptr && g_spmm_path == 0)
    rc = gcge_hip_pattern_spmm_vals(nr, A->d_pid + r0, A->d_tab, A->npat, A->pat_lt, A->pat_span, A->pat_span2, dx + r0 * ldx, ldx,
                                    y, ldy, m, d_dots, d_yy, g_stream, A->pat_near, A->d_rowval ? A->d_rowval + 8 * r0 : nullptr);
  if (rc != -1) return rc;
  // whole-matrix products only from here: neither the rows of a block nor those of a tile are a row range
  if (A->star != nullptr && d_dots == nullptr && r0 == 0 && r1 == A->nrows && g_spmm_path == 0) {
    rc = gcge_hip_star_spmm(A->star, dx, ldx, dy, ldy, m, g_stream);              // star + diagonal of EVERY row ...
    if (rc == 0) rc = gcge_hip_dense_spmm(A->star_rem, dx, ldx, dy, ldy, m, g_stream, 4);   // ... + what the other rows hold beyond it
  }
  if (rc != -1) return rc;
  if (A->dense != nullptr && d_dots == nullptr && r0 == 0 && r1 == A->nrows && g_spmm_path != 1 && g_spmm_path != 3 && g_spmm_path != 4)
    rc = gcge_hip_dense_spmm(A->dense, dx, ldx, dy, ldy, m, g_stream, 0);
  if (rc != -1) return rc;
  if (A->tile != nullptr && d_dots == nullptr && r0 == 0 && r1 == A->nrows && g_spmm_path != 1 && g_spmm_path != 3)
    rc = gcge_hip_tile_spmm(A->tile, dx, ldx, dy, ldy, m, g_stream);   // whole-matrix products only: a tile's rows are not a row range
  if (rc != -1) return rc;
  if (d_dots) {   // generic fused kernel (the caller checked its contract), y.y by a second pass over y
    rc = gcge_hip_pad8_spmm_dot(nr, A->d_orp + r0, A->d_pcol, A->d_pval, dx, ldx, y, ldy, m, d_dots, g_stream, r0);   // (x.y over the strip's OWN rows of x)
    if (rc == 0 && d_yy) rc = gcge_hip_coldots(nr, y, ldy, y, ldy, m, d_yy, g_stream);
    return rc;
  }
  if (m >= 16) {
    gcge_hip_spmm_pad8_auto(A->nrows > 0 ? (double)A->noct / A->nrows : 1.0);
    rc = gcge_hip_pad8_spmm(nr, A->d_orp + r0, A->d_pcol, A->d_pval, dx, ldx, y, ldy, m, g_stream);
  }
  if (rc == -1) rc = gcge_hip_csr_spmm(nr, A->d_rowptr + r0, A->d_colidx, A->d_val, dx, ldx, y, ldy, m, g_stream);
  return rc;
}

static int g_halo_overlap = 1;
extern "C" void gcge_hip_set_halo_overlap(int on) { g_halo_overlap = on; }

// Y[:, 0:m) = A X[:, c_begin : c_begin + m) for a matrix in grid form (spmm_star.hip), whole or a row slab cut on plane
// boundaries.  On a slab with a split exchange the planes that need no halo row are swept while the halo is in flight (the
// reference's distributed product does the same with its diagonal block: app/app_phg.c:307-357), the first and last STAR_R
// planes and the rows outside the grid form (blocks + listed rows, which may reference any halo row) follow its arrival.
// dd != NULL (4 m doubles, device): the column sums x.y and y.y over the star rows (dd[0:2m)) and over the other rows (dd[2m:4m)).
// -1 before anything was launched or sent: operands the sweep does not take.
static long g_star_products = 0, g_star_split_products = 0;
extern "C" void gcge_hip_star_product_stats(long* products, long* split) { if (products) *products = g_star_products; if (split) *split = g_star_split_products; }
static int g_star_race_probe = 0;   // MEASUREMENT ONLY (results are wrong): blocks + listed rows on a second stream beside the sweep, unordered
extern "C" void gcge_hip_star_race_probe(int on) { g_star_race_probe = on; }
static int star_product(GCGE_HIP_MAT_* A, GcgeHipMV* vx, int c_begin, double* dy, long ldy, int m, double* dd) {
  const double* dx = vx->d + c_begin;
  const long ldx = vx->ld;
  if ((m & 1) || (ldx & 1) || (ldy & 1) || ((uintptr_t)dx & 15) || ((uintptr_t)dy & 15) || dx == dy) return -1;
  const bool split = A->nghost > 0 && g_halo_overlap && A->exchange_begin != nullptr && A->exchange_end != nullptr && m <= A->buf_cols &&
                     gcge_hip_star_interior(A->star, nullptr, nullptr);
  int rc;
  ++g_star_products;
  if (split) {
    ++g_star_split_products;
    if (A->nsend > 0) {
      long tot = (long)A->nsend * m, g = (tot + 255) / 256; if (g > 4096) g = 4096;
      hipLaunchKernelGGL(halo_pack, dim3((unsigned)g), dim3(256), 0, g_stream, A->nsend, A->d_send_rows, dx, ldx, m, A->sendbuf);
    }
    A->exchange_begin(A->sendbuf, A->recvbuf, m, A->exchange_ctx);
    rc = gcge_hip_star_spmm_part(A->star, dx, ldx, dy, ldy, m, dd, g_stream, 1);        // overlaps the transfers
    A->exchange_end(A->exchange_ctx);
    long tot = (long)A->nghost * m, g = (tot + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(halo_unpack, dim3((unsigned)g), dim3(256), 0, g_stream, A->nghost, A->recvbuf, m, vx->d + (long)A->nrows * ldx + c_begin, ldx);
    if (rc == 0) rc = gcge_hip_star_spmm_part(A->star, dx, ldx, dy, ldy, m, dd, g_stream, 2);
  } else {
    halo_fetch(A, vx, c_begin, m);
    if (g_star_race_probe) {
      static hipStream_t side = nullptr; static hipEvent_t e0, e1;
      if (!side) { GCGE_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking)); GCGE_HIP_CHECK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); GCGE_HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); }
      GCGE_HIP_CHECK(hipEventRecord(e0, g_stream)); GCGE_HIP_CHECK(hipStreamWaitEvent(side, e0, 0));
      if (g_star_race_probe == 2) gcge_hip_dense_spmm(A->star_rem, dx, ldx, dy, ldy, m, side, 4);
      rc = gcge_hip_star_spmm_part(A->star, dx, ldx, dy, ldy, m, dd, g_stream, 0);
      if (g_star_race_probe == 1) gcge_hip_dense_spmm(A->star_rem, dx, ldx, dy, ldy, m, side, 4);
      GCGE_HIP_CHECK(hipEventRecord(e1, side)); GCGE_HIP_CHECK(hipStreamWaitEvent(g_stream, e1, 0));
      return rc;
    }
    rc = gcge_hip_star_spmm_part(A->star, dx, ldx, dy, ldy, m, dd, g_stream, 0);
  }
  GCGE_REQUIRE(rc == 0, "star product: sweep");
  rc = gcge_hip_dense_spmm(A->star_rem, dx, ldx, dy, ldy, m, g_stream, 4);               // += what the rows with more than the star hold beyond it
  GCGE_REQUIRE(rc == 0, "star product: blocks and listed rows");
  if (dd != nullptr) {
    int nlist = 0; const int* list = gcge_hip_dense_row_list(A->star_rem, &nlist);
    GCGE_REQUIRE(gcge_hip_star_coldots2_rows(nlist, list, dx, ldx, dy, ldy, m, dd + 2 * (size_t)m, g_stream) == 0, "star product: sums over the listed rows");
  }
  return 0;
}

// Y[:, 0:m) = A X[:, c_begin : c_begin+m) on a row slab, halo included; d_dots / d_yy as in spmm_rows (3 m doubles of
// scratch behind each when the product is split).  With a split exchange the interior rows are multiplied while the
// halo rows travel, the two boundary strips follow.
static int spmm_halo(GCGE_HIP_MAT_* A, GcgeHipMV* vx, int c_begin, double* dy, long ldy, int m, double* d_dots, double* d_yy,
                     const CgPass* cg = nullptr) {
  const double* dx = vx->d + c_begin;
  if (A->star != nullptr && g_spmm_path == 0 && cg == nullptr && d_dots == nullptr && d_yy == nullptr) {
    const int rc = star_product(A, vx, c_begin, dy, ldy, m, nullptr);
    if (rc != -1) return rc;
  }
  const bool split = g_halo_overlap && A->nghost > 0 && A->exchange_begin != nullptr && A->exchange_end != nullptr &&
                     m <= A->buf_cols && A->ov_hi - A->ov_lo >= A->nrows / 2 &&
                     A->tile == nullptr && A->dense == nullptr;   // (the block and tile forms multiply whole matrices, not row strips)
  if (!split) {
    halo_fetch(A, vx, c_begin, m);
    return spmm_rows(A, 0, A->nrows, dx, vx->ld, dy, ldy, m, d_dots, d_yy, cg);
  }
  GCGE_REQUIRE(A->buf_cols > 0, "MatDotMultiVec: halo plan installed (gcge_hip_mat_set_halo)");
  if (A->nsend > 0) {
    long tot = (long)A->nsend * m, g = (tot + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(halo_pack, dim3((unsigned)g), dim3(256), 0, g_stream, A->nsend, A->d_send_rows, dx, vx->ld, m, A->sendbuf);
  }
  A->exchange_begin(A->sendbuf, A->recvbuf, m, A->exchange_ctx);
  double* d1 = d_dots ? d_dots + m : nullptr; double* d2 = d_dots ? d_dots + 2 * m : nullptr;
  double* y1 = d_yy ? d_yy + m : nullptr;     double* y2 = d_yy ? d_yy + 2 * m : nullptr;
  int rc = spmm_rows(A, A->ov_lo, A->ov_hi, dx, vx->ld, dy, ldy, m, d1, y1, cg);      // interior, overlaps the transfers
  A->exchange_end(A->exchange_ctx);
  {
    long tot = (long)A->nghost * m, g = (tot + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(halo_unpack, dim3((unsigned)g), dim3(256), 0, g_stream, A->nghost, A->recvbuf, m,
                       vx->d + (long)A->nrows * vx->ld + c_begin, vx->ld);
  }
  if (rc == 0) rc = spmm_rows(A, 0, A->ov_lo, dx, vx->ld, dy, ldy, m, d2, y2, cg);     // leading boundary strip
  double* d3 = d_dots ? stage_d2(2 * (size_t)m) : nullptr;
  if (rc == 0) rc = spmm_rows(A, A->ov_hi, A->nrows, dx, vx->ld, dy, ldy, m, d3, d_yy ? d3 + m : nullptr, cg);   // trailing strip
  if (d_dots) hipLaunchKernelGGL(add3_kernel, dim3((m + 63) / 64), dim3(64), 0, g_stream, d_dots, d1, d2, d3, m);
  if (d_yy) hipLaunchKernelGGL(add3_kernel, dim3((m + 63) / 64), dim3(64), 0, g_stream, d_yy, y1, y2, d3 + m, m);
  return rc;
}

// app_ccs.c:50-139;  mat == NULL copies (identity B)
static void HIP_MatDotMultiVec(void* mat, void** x, void** y, int* start, int* end, struct OPS_* ops) {
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  GcgeHipMV *vx = (GcgeHipMV*)x, *vy = (GcgeHipMV*)y;
  const int m = end[0] - start[0];
  enter();
  SlotTimer tm_(mat ? "MatDotMultiVec" : "MatDotMultiVec (copy)", m);
  GCGE_REQUIRE(m == end[1] - start[1], "MatDotMultiVec: equal column counts");
  if (m <= 0) return;
  GCGE_REQUIRE(vx != vy || end[0] <= start[1] || end[1] <= start[0], "MatDotMultiVec: x and y ranges must not overlap");
  GCGE_REQUIRE(start[0] >= 0 && end[0] <= vx->ncols && start[1] >= 0 && end[1] <= vy->ncols, "MatDotMultiVec: column ranges");
  if (A != nullptr && A->rect_ncols > 0) {   // a prolongation of the multigrid hierarchy (multigrid.hip): rows of level l x rows of level l + 1
    GCGE_REQUIRE(vx->nrows == A->rect_ncols && vy->nrows == A->nrows, "MatDotMultiVec: shapes of a rectangular matrix");
    GCGE_REQUIRE(gcge_hip_csr_spmm(A->nrows, A->d_rowptr, A->d_colidx, A->d_val, vx->d + start[0], vx->ld, vy->d + start[1], vy->ld, m, g_stream) == 0,
                 "MatDotMultiVec: kernel launch (rectangular matrix)");
    return;
  }
  GCGE_REQUIRE(vx->nrows == vy->nrows, "MatDotMultiVec: equal row counts");
  if (A != nullptr) GCGE_REQUIRE(A->nrows == vy->nrows && A->nrows + A->nghost <= vx->nrows_alloc, "MatDotMultiVec: matrix/vector shapes");
  // blocks and matrix must live in ONE row order (the back-end re-orders matrices without a grid: mat_upload.hip "row orders")
  if (A != nullptr) GCGE_REQUIRE(real_perm(vx->perm) == real_perm(A->perm) && real_perm(vy->perm) == real_perm(A->perm), "MatDotMultiVec: the blocks were created for a matrix in another row order");
  if (A == nullptr) {
    gcge_hip_axpby(vy->nrows, 1.0, vx->d + start[0], vx->ld, 0.0, vy->d + start[1], vy->ld, m, g_stream);
    return;
  }
  double* dy = vy->d + start[1];
  int rc = -1;
  SpmmEvent ev;
  if (g_prof_on) {   // (on a row slab the interval also holds the halo exchange)
    GCGE_HIP_CHECK(hipEventCreate(&ev.e0)); GCGE_HIP_CHECK(hipEventCreate(&ev.e1));
    ev.kind = 0;
    ev.m = m; ev.rows = A->nrows;   // algorithmic bytes: values+indices once, row pointers once, X once, Y once
    ev.bytes = 12.0 * (double)A->nnz + 4.0 * ((double)A->nrows + 1.0) + 16.0 * (double)A->nrows * m;
    GCGE_HIP_CHECK(hipEventRecord(ev.e0, g_stream));
  }
  // Column ranges that are not 16-byte pairs (an odd first column or count: the residual check of a solve with an odd number
  // of locked pairs) would send a matrix WITHOUT a pattern form to the scalar CSR kernel — 23.6 ms instead of 3.7 on BASELINE
  // config 5's matrix.  The whole-matrix forms multiply the enclosing even range of X into a scratch block instead (columns of
  // the padding are allocated and zero), the requested columns are copied out.
  const int xs = start[0] & ~1, xe = (end[0] + 1) & ~1;
  if (A->d_pid == nullptr && (A->star != nullptr || A->dense != nullptr || A->tile != nullptr) && g_spmm_path == 0 &&
      m >= 8 && ((start[0] | start[1] | m) & 1) && xe <= vx->ld) {
    const int mw = xe - xs;
    const long ldt = ((long)mw + 7) / 8 * 8;
    const size_t bytes = (size_t)A->nrows * ldt * sizeof(double);
    double* t = (double*)pool_alloc(bytes);
    halo_fetch(A, vx, xs, mw);                                             // (row slabs: the halo rows of the widened range)
    rc = spmm_rows(A, 0, A->nrows, vx->d + xs, vx->ld, t, ldt, mw, nullptr, nullptr);
    if (rc == 0) rc = gcge_hip_axpby(vy->nrows, 1.0, t + (start[0] - xs), ldt, 0.0, dy, vy->ld, m, g_stream);
    pool_free(t, bytes);   // (one stream: whoever takes the block next is ordered behind the copy)
  } else
  if (A->nghost > 0 && m > A->buf_cols) {   // wider than the exchange buffers: column chunks, one after the other
    rc = 0;
    for (int c0 = 0; c0 < m && rc == 0; c0 += A->buf_cols) {
      const int mc = (m - c0 < A->buf_cols) ? m - c0 : A->buf_cols;
      rc = spmm_halo(A, vx, start[0] + c0, dy + c0, vy->ld, mc, nullptr, nullptr);
    }
  } else {
    rc = spmm_halo(A, vx, start[0], dy, vy->ld, m, nullptr, nullptr);
  }
  if (g_prof_on) { GCGE_HIP_CHECK(hipEventRecord(ev.e1, g_stream)); g_prof.push_back(ev); }
  GCGE_REQUIRE(rc == 0, "MatDotMultiVec: kernel launch");
}
// Fused  y = A x  and  dots[j] = sum_r x[r,j] y[r,j]  (the p.w of a CG step) — LOCAL part only; the
// caller reduces over ranks.  Falls back to SpMM + column dots when the fast kernel's alignment
// contract is not met.  Internal entry point of the fused block CG (block_pcg.hip).
extern "C" void gcge_hip_spmm_dot2_mv(void* mat, void** x, void** y, int* start, int* end, double* host_dots,
                                      double* host_yy, struct OPS_* ops);
extern "C" void gcge_hip_spmm_dot_mv(void* mat, void** x, void** y, int* start, int* end, double* host_dots,
                                     struct OPS_* ops) {
  gcge_hip_spmm_dot2_mv(mat, x, y, start, end, host_dots, nullptr, ops);
}
// host_yy != NULL: additionally yy[j] = sum_r y[r,j]^2 (local part) — free on the pattern path
extern "C" void gcge_hip_spmm_dot2_mv(void* mat, void** x, void** y, int* start, int* end, double* host_dots,
                                      double* host_yy, struct OPS_* ops) {
  enter();
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  GcgeHipMV *vx = (GcgeHipMV*)x, *vy = (GcgeHipMV*)y;
  const int m = end[0] - start[0];
  if (m <= 0) return;
  const bool aligned = A != nullptr && (m % 2 == 0) && (vx->ld % 2 == 0) && (vy->ld % 2 == 0) &&
                       (((uintptr_t)(vx->d + start[0]) & 15) == 0) && (((uintptr_t)(vy->d + start[1]) & 15) == 0);
  const bool use_pat = aligned && A->d_pid != nullptr && g_spmm_path == 0;
  // generic matrices with long rows (>= 2.5 octets on average): the plain pad-8 kernel with one or two rows per wave
  // plus separate column dots beats the fused kernel (SiO2-like, 36 nnz/row: 6.8 + 1.5 ms against 11 ms)
  const bool long_rows = A != nullptr && A->nrows > 0 && (double)A->noct / A->nrows >= 2.5;
  const bool fast = aligned && (use_pat || (m >= 16 && m <= 128 && !long_rows)) && (A->nghost == 0 || m <= A->buf_cols);
  if (!fast && host_yy != nullptr && A != nullptr && A->star != nullptr && g_spmm_path == 0 && vx != vy &&
      vx->nrows == vy->nrows && A->nrows == vy->nrows && A->nrows + A->nghost <= vx->nrows_alloc) {
    // grid form: the sweep over the star rows sums x.y and y.y of its rows on the way (registers), a short sweep over the LIST of
    // the other rows adds theirs — no pass over the two blocks afterwards
    SpmmEvent ev;
    if (g_prof_on) {
      GCGE_HIP_CHECK(hipEventCreate(&ev.e0)); GCGE_HIP_CHECK(hipEventCreate(&ev.e1));
      ev.m = m; ev.rows = A->nrows; ev.kind = 0;
      ev.bytes = 12.0 * (double)A->nnz + 4.0 * ((double)A->nrows + 1.0) + 16.0 * (double)A->nrows * m;
      GCGE_HIP_CHECK(hipEventRecord(ev.e0, g_stream));
    }
    double* dd = stage_d(4 * (size_t)m);
    const int rc = star_product(A, vx, start[0], vy->d + start[1], vy->ld, m, dd);
    if (rc == 0) {
      if (g_prof_on) { GCGE_HIP_CHECK(hipEventRecord(ev.e1, g_stream)); g_prof.push_back(ev); }
      double* hd = stage_h(4 * (size_t)m);
      GCGE_HIP_CHECK(hipMemcpyAsync(hd, dd, 4 * (size_t)m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
      GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
      for (int j = 0; j < m; ++j) { host_dots[j] = hd[j] + hd[2 * m + j]; host_yy[j] = hd[m + j] + hd[3 * m + j]; }
      return;
    }
    if (g_prof_on) { hipEventDestroy(ev.e0); hipEventDestroy(ev.e1); }   // (operands the sweep does not take: the generic route below)
  }
  if (!fast) {
    HIP_MatDotMultiVec(mat, x, y, start, end, ops);
    if (host_yy && vx->nrows == vy->nrows) {   // x.y and y.y in one sweep over the two blocks
      double* dd = stage_d(2 * (size_t)m);
      GCGE_REQUIRE(gcge_hip_coldots2(vx->nrows, vx->d + start[0], vx->ld, vy->d + start[1], vy->ld, m, dd, g_stream) == 0, "spmm_dot: column sums");
      double* hd = stage_h(2 * (size_t)m);
      GCGE_HIP_CHECK(hipMemcpyAsync(hd, dd, 2 * (size_t)m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
      GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
      memcpy(host_dots, hd, m * sizeof(double));
      memcpy(host_yy, hd + m, m * sizeof(double));
      return;
    }
    local_inner_prod('D', x, y, 0, start, end, host_dots, 1, ops);       // (LOCAL parts whatever GCGE_SetLocalInnerProdReduces says: the caller reduces)
    if (host_yy) {
      int s2[2] = {start[1], start[1]}, e2[2] = {end[1], end[1]};
      local_inner_prod('D', y, y, 0, s2, e2, host_yy, 1, ops);
    }
    return;
  }
  GCGE_REQUIRE(vx != vy && vx->nrows == vy->nrows && A->nrows == vy->nrows, "spmm_dot: shapes");
  GCGE_REQUIRE(start[0] >= 0 && end[0] <= vx->ncols && start[1] >= 0 && end[1] <= vy->ncols, "spmm_dot: column ranges");
  GCGE_REQUIRE(A->nrows + A->nghost <= vx->nrows_alloc, "spmm_dot: halo rows allocated");
  GCGE_REQUIRE(A->nghost == 0 || m <= A->buf_cols, "spmm_dot: block wider than the halo buffers");
  double* dd = stage_d(6 * (size_t)m);            // x.y sums (3 m: total + the strips of a split product), then y.y sums
  double* dyy = host_yy ? dd + 3 * (size_t)m : nullptr;
  SpmmEvent ev;
  if (g_prof_on) {   // the fused kernel IS the K1 launch of a CG step (same algorithmic bytes: the dots add no HBM traffic)
    GCGE_HIP_CHECK(hipEventCreate(&ev.e0)); GCGE_HIP_CHECK(hipEventCreate(&ev.e1));
    ev.m = m; ev.rows = A->nrows; ev.kind = 0;
    ev.bytes = 12.0 * (double)A->nnz + 4.0 * ((double)A->nrows + 1.0) + 16.0 * (double)A->nrows * m;
    GCGE_HIP_CHECK(hipEventRecord(ev.e0, g_stream));
  }
  int rc = spmm_halo(A, vx, start[0], vy->d + start[1], vy->ld, m, dd, dyy);
  if (g_prof_on) { GCGE_HIP_CHECK(hipEventRecord(ev.e1, g_stream)); g_prof.push_back(ev); }
  GCGE_REQUIRE(rc == 0, "spmm_dot: kernel launch");
  double* hd = stage_h(2 * (size_t)m);
  GCGE_HIP_CHECK(hipMemcpyAsync(hd, dd, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  if (dyy) GCGE_HIP_CHECK(hipMemcpyAsync(hd + m, dyy, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  memcpy(host_dots, hd, m * sizeof(double));
  if (host_yy) memcpy(host_yy, hd + m, m * sizeof(double));
}

// The same with the sums LEFT ON THE DEVICE and nothing waited for (the device-scalar loop of block_pcg.hip on matrices whose
// product is stored): y[:, cy : cy + m) = A x[:, cx : cx + m), d_out[0, m) = x.y, d_out[m, 2m) = y.y over the local rows (d_out: 2 m
// doubles).  -1, nothing touched: operands the fused kernels do not take (odd widths or offsets, unaligned blocks).
__global__ void dot2_sum_kernel(int m, const double* __restrict__ a0, const double* __restrict__ a1, const double* __restrict__ b0,
                                const double* __restrict__ b1, double* __restrict__ out) {   // out = [a0 + b0 | a1 + b1] (b: NULL = none)
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) { out[j] = a0[j] + (b0 != nullptr ? b0[j] : 0.0); out[m + j] = a1[j] + (b1 != nullptr ? b1[j] : 0.0); }
}
// 1: gcge_hip_spmm_dot2_dev takes these operands (its contract, for callers that must decide BEFORE touching anything)
extern "C" int gcge_hip_spmm_dot2_dev_ok(void* mat, void** x, void** y, int cx, int cy, int m) {
  const GCGE_HIP_MAT_* A = (const GCGE_HIP_MAT_*)mat;
  const GcgeHipMV *vx = (const GcgeHipMV*)x, *vy = (const GcgeHipMV*)y;
  if (A == nullptr || A->rect_ncols > 0 || vx == nullptr || vy == nullptr) return 0;
  if (m <= 0 || (m & 1) || (cx & 1) || (cy & 1) || (vx->ld & 1) || (vy->ld & 1) || vx == vy) return 0;
  if (((uintptr_t)(vx->d + cx) & 15) || ((uintptr_t)(vy->d + cy) & 15)) return 0;
  if (vx->nrows != vy->nrows || A->nrows != vy->nrows || A->nrows + A->nghost > vx->nrows_alloc) return 0;
  if (A->nghost > 0 && m > A->buf_cols) return 0;
  return 1;
}
extern "C" int gcge_hip_spmm_dot2_dev(void* mat, void** x, void** y, int cx, int cy, int m, double* d_out) {
  enter();
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  GcgeHipMV *vx = (GcgeHipMV*)x, *vy = (GcgeHipMV*)y;
  if (!gcge_hip_spmm_dot2_dev_ok(mat, x, y, cx, cy, m)) return -1;
  GCGE_REQUIRE(cx >= 0 && cx + m <= vx->ncols && cy >= 0 && cy + m <= vy->ncols, "spmm_dot2_dev: column ranges");
  SpmmEvent ev;
  if (g_prof_on) {
    GCGE_HIP_CHECK(hipEventCreate(&ev.e0)); GCGE_HIP_CHECK(hipEventCreate(&ev.e1));
    ev.m = m; ev.rows = A->nrows; ev.kind = 0;
    ev.bytes = 12.0 * (double)A->nnz + 4.0 * ((double)A->nrows + 1.0) + 16.0 * (double)A->nrows * m;
    GCGE_HIP_CHECK(hipEventRecord(ev.e0, g_stream));
  }
  const bool use_pat = A->d_pid != nullptr && g_spmm_path == 0;
  const bool long_rows = A->nrows > 0 && (double)A->noct / A->nrows >= 2.5;
  const bool fast = use_pat || (m >= 16 && m <= 128 && !long_rows);
  int rc = -1;
  if (!fast && A->star != nullptr && g_spmm_path == 0) {
    double* dd = stage_d(4 * (size_t)m);                               // sweep: x.y | y.y, listed rows: x.y | y.y
    rc = star_product(A, vx, cx, vy->d + cy, vy->ld, m, dd);
    if (rc == 0) hipLaunchKernelGGL(dot2_sum_kernel, dim3((m + 127) / 128), dim3(128), 0, g_stream, m, (const double*)dd, (const double*)(dd + m),
                                    (const double*)(dd + 2 * (size_t)m), (const double*)(dd + 3 * (size_t)m), d_out);
  }
  if (rc != 0 && fast) {
    double* dd = stage_d(6 * (size_t)m);                               // x.y (3 m: total + the strips of a split product), then y.y
    rc = spmm_halo(A, vx, cx, vy->d + cy, vy->ld, m, dd, dd + 3 * (size_t)m);
    GCGE_REQUIRE(rc == 0, "spmm_dot2_dev: kernel launch");
    hipLaunchKernelGGL(dot2_sum_kernel, dim3((m + 127) / 128), dim3(128), 0, g_stream, m, (const double*)dd, (const double*)(dd + 3 * (size_t)m),
                       (const double*)nullptr, (const double*)nullptr, d_out);
  } else if (rc != 0) {
    rc = spmm_halo(A, vx, cx, vy->d + cy, vy->ld, m, nullptr, nullptr);
    GCGE_REQUIRE(rc == 0, "spmm_dot2_dev: kernel launch");
    GCGE_REQUIRE(gcge_hip_coldots2(vx->nrows, vx->d + cx, vx->ld, vy->d + cy, vy->ld, m, d_out, g_stream) == 0, "spmm_dot2_dev: column sums");
  }
  if (g_prof_on) { GCGE_HIP_CHECK(hipEventRecord(ev.e1, g_stream)); g_prof.push_back(ev); }
  return 0;
}

// ---- the two passes of a fused block-CG iteration (block_pcg.hip) on a pattern matrix -----------------------------
// The product w = A p is formed twice and never stored: pass 1 reads p and returns p.w and w.w (that fixes alpha and
// beta), pass 2 reads p again, rebuilds w in registers and applies  r -= alpha w ; p_new = r + beta p  on the spot.
// 1 + 4 block streams per iteration instead of 2 (product) + 5 (update sweep).  Both return -1 without touching
// anything when the matrix or the operands do not qualify (no pattern form, odd widths, halo wider than the buffers).
extern "C" int gcge_hip_cg_fusable(void* mat, void** p, int ncols) {
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat; GcgeHipMV* vp = (GcgeHipMV*)p;
  if (A == nullptr || A->d_pid == nullptr || g_spmm_path != 0 || getenv("GCGE_CG_NO_RECOMPUTE") != nullptr) return 0;
  if ((ncols & 1) || (vp->ld & 1) || ((uintptr_t)vp->d & 15)) return 0;
  if (A->nghost > 0 && ncols > A->buf_cols) return 0;
  return 1;
}
// Does forming the product twice pay?  Only where the product kernel is bound by HBM: the chain kernel with line exchange
// (about 3 loads per row).  The plain pattern kernel issues 7+ cache-served loads per row and is bound by those, so a
// second product costs more than the two block streams it saves (FE pair n = 10^6: 2.3 against 1.9 ms per iteration).
extern "C" int gcge_hip_cg_recompute_pays(void* mat) {
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  if (A == nullptr || A->d_pid == nullptr) return 0;
  return gcge_hip_mat_pattern_chain(A) == 2;
}
// d_out[0, m) = sum_r p[r,j] (A p)[r,j], d_out[m, 2m) = sum_r (A p)[r,j]^2 over the LOCAL rows, left on the DEVICE (d_out holds
// >= 6 m doubles, the rest is scratch of the split product); fetches the halo rows of p; nothing is waited for
extern "C" int gcge_hip_cg_pass1_dev(void* mat, void** p, int c0, int m, double* d_out) {
  enter();
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat; GcgeHipMV* vp = (GcgeHipMV*)p;
  if (!gcge_hip_cg_fusable(mat, p, m) || (c0 & 1)) return -1;
  GCGE_REQUIRE(c0 >= 0 && c0 + m <= vp->ncols && A->nrows == vp->nrows && A->nrows + A->nghost <= vp->nrows_alloc, "cg_pass1: shapes");
  double* dd = d_out;
  double* dyy = dd + 3 * (size_t)m;
  SpmmEvent ev;
  if (g_prof_on) {   // algorithmic bytes: matrix once, p once
    GCGE_HIP_CHECK(hipEventCreate(&ev.e0)); GCGE_HIP_CHECK(hipEventCreate(&ev.e1));
    ev.m = m; ev.rows = A->nrows; ev.kind = 2;
    ev.bytes = 12.0 * (double)A->nnz + 4.0 * ((double)A->nrows + 1.0) + 8.0 * (double)A->nrows * m;
    GCGE_HIP_CHECK(hipEventRecord(ev.e0, g_stream));
  }
  const CgPass cg = {2, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0};
  const int rc = spmm_halo(A, vp, c0, nullptr, 0, m, dd, dyy, &cg);
  if (g_prof_on) { GCGE_HIP_CHECK(hipEventRecord(ev.e1, g_stream)); g_prof.push_back(ev); }
  GCGE_REQUIRE(rc == 0, "cg_pass1: kernel launch");
  GCGE_HIP_CHECK(hipMemcpyAsync(dd + m, dyy, m * sizeof(double), hipMemcpyDeviceToDevice, g_stream));   // both sums side by side
  return 0;
}
// the same with the sums returned to the host (one stream synchronisation)
extern "C" int gcge_hip_cg_pass1_mv(void* mat, void** p, int c0, int m, double* host_pw, double* host_ww) {
  double* dd = stage_d(6 * (size_t)m);
  if (gcge_hip_cg_pass1_dev(mat, p, c0, m, dd) != 0) return -1;
  double* hd = stage_h(2 * (size_t)m);
  GCGE_HIP_CHECK(hipMemcpyAsync(hd, dd, 2 * (size_t)m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  memcpy(host_pw, hd, m * sizeof(double));
  memcpy(host_ww, hd + m, m * sizeof(double));
  return 0;
}
// r[:, c0:c0+m) -= (A p) diag(alpha); pnew[:, c0:c0+m) = r diag(cr) + p diag(cb); d_rho[j] = sum_r cr_j r[r,j]^2 (local), left on
// the DEVICE.  The halo rows of p must be the ones pass 1 fetched (p unchanged since).  d_alpha / d_beta / d_flag: device, m each.
extern "C" int gcge_hip_cg_pass2_dev(void* mat, void** p, void** r, void** pnew, int c0, int m, const double* d_alpha,
                                     const double* d_beta, const int* d_flag, double* d_rho) {
  enter();
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  GcgeHipMV *vp = (GcgeHipMV*)p, *vr = (GcgeHipMV*)r, *vn = (GcgeHipMV*)pnew;
  if (!gcge_hip_cg_fusable(mat, p, m) || (c0 & 1) || (vr->ld & 1) || (vn->ld & 1) || ((uintptr_t)vr->d & 15) ||
      ((uintptr_t)vn->d & 15) || vn == vp) return -1;
  GCGE_REQUIRE(c0 >= 0 && c0 + m <= vp->ncols && c0 + m <= vr->ncols && c0 + m <= vn->ncols, "cg_pass2: column ranges");
  GCGE_REQUIRE(A->nrows == vp->nrows && A->nrows == vr->nrows && A->nrows == vn->nrows, "cg_pass2: row counts");
  SpmmEvent ev;
  if (g_prof_on) {   // algorithmic bytes: matrix once, p and r read, r and p_new written
    GCGE_HIP_CHECK(hipEventCreate(&ev.e0)); GCGE_HIP_CHECK(hipEventCreate(&ev.e1));
    ev.m = m; ev.rows = A->nrows; ev.kind = 3;
    ev.bytes = 12.0 * (double)A->nnz + 4.0 * ((double)A->nrows + 1.0) + 32.0 * (double)A->nrows * m;
    GCGE_HIP_CHECK(hipEventRecord(ev.e0, g_stream));
  }
  const CgPass cg = {3, vr->d + c0, vr->ld, vn->d + c0, vn->ld, d_alpha, d_beta, d_flag, nullptr, 0};
  const int rc = spmm_rows(A, 0, A->nrows, vp->d + c0, vp->ld, nullptr, 0, m, d_rho, nullptr, &cg);
  if (g_prof_on) { GCGE_HIP_CHECK(hipEventRecord(ev.e1, g_stream)); g_prof.push_back(ev); }
  GCGE_REQUIRE(rc == 0, "cg_pass2: kernel launch");
  return 0;
}
extern "C" int gcge_hip_cg_pass2_mv(void* mat, void** p, void** r, void** pnew, int c0, int m, const double* d_alpha,
                                    const double* d_beta, const int* d_flag, double* host_rho) {
  double* dd = stage_d(6 * (size_t)m);
  if (gcge_hip_cg_pass2_dev(mat, p, r, pnew, c0, m, d_alpha, d_beta, d_flag, dd) != 0) return -1;
  double* hd = stage_h((size_t)m);
  GCGE_HIP_CHECK(hipMemcpyAsync(hd, dd, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  memcpy(host_rho, hd, m * sizeof(double));
  return 0;
}

// The same second pass WITHOUT a stored residual (kernel MODE 7): r_k = p_k - beta_{k-1} p_{k-1} is rebuilt from the previous
// direction (pprev, read only; d_betaprev: the beta that formed p_k — zeros in the first iteration, where pprev may be p
// itself), pnew[:, c0:c0+m) = r' diag(cr) + p diag(cb) with r' = r_k - (A p) diag(alpha), d_rho[j] = sum_r cr_j r'[r,j]^2.
// Reads p, pprev, writes pnew: 3 block streams instead of 4.
extern "C" int gcge_hip_cg_pass2i_dev(void* mat, void** p, void** pprev, void** pnew, int c0, int m, const double* d_alpha,
                                      const double* d_beta, const int* d_flag, const double* d_betaprev, double* d_rho) {
  enter();
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  GcgeHipMV *vp = (GcgeHipMV*)p, *vq = (GcgeHipMV*)pprev, *vn = (GcgeHipMV*)pnew;
  if (!gcge_hip_cg_fusable(mat, p, m) || (c0 & 1) || (vq->ld & 1) || (vn->ld & 1) || ((uintptr_t)vq->d & 15) ||
      ((uintptr_t)vn->d & 15) || vn == vp || vn == vq || d_betaprev == nullptr) return -1;
  GCGE_REQUIRE(c0 >= 0 && c0 + m <= vp->ncols && c0 + m <= vq->ncols && c0 + m <= vn->ncols, "cg_pass2i: column ranges");
  GCGE_REQUIRE(A->nrows == vp->nrows && A->nrows == vq->nrows && A->nrows == vn->nrows, "cg_pass2i: row counts");
  SpmmEvent ev;
  if (g_prof_on) {   // algorithmic bytes: matrix once, p and p_prev read, p_new written
    GCGE_HIP_CHECK(hipEventCreate(&ev.e0)); GCGE_HIP_CHECK(hipEventCreate(&ev.e1));
    ev.m = m; ev.rows = A->nrows; ev.kind = 3;
    ev.bytes = 12.0 * (double)A->nnz + 4.0 * ((double)A->nrows + 1.0) + 24.0 * (double)A->nrows * m;
    GCGE_HIP_CHECK(hipEventRecord(ev.e0, g_stream));
  }
  const CgPass cg = {7, vq->d + c0, vq->ld, vn->d + c0, vn->ld, d_alpha, d_beta, d_flag, d_betaprev, 0};
  const int rc = spmm_rows(A, 0, A->nrows, vp->d + c0, vp->ld, nullptr, 0, m, d_rho, nullptr, &cg);
  if (g_prof_on) { GCGE_HIP_CHECK(hipEventRecord(ev.e1, g_stream)); g_prof.push_back(ev); }
  GCGE_REQUIRE(rc == 0, "cg_pass2i: kernel launch");
  return 0;
}
extern "C" int gcge_hip_cg_pass2i_mv(void* mat, void** p, void** pprev, void** pnew, int c0, int m, const double* d_alpha,
                                     const double* d_beta, const int* d_flag, const double* d_betaprev, double* host_rho) {
  double* dd = stage_d(6 * (size_t)m);
  if (gcge_hip_cg_pass2i_dev(mat, p, pprev, pnew, c0, m, d_alpha, d_beta, d_flag, d_betaprev, dd) != 0) return -1;
  double* hd = stage_h((size_t)m);
  GCGE_HIP_CHECK(hipMemcpyAsync(hd, dd, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  memcpy(host_rho, hd, m * sizeof(double));
  return 0;
}

// Start of the block CG in one sweep (kernel MODE 5): r[:, rc0:rc0+m) = b[:, bc0:bc0+m) - A x[:, xc0:xc0+m), p0 = r (same
// columns rc0.. of the block p0), host_rho[j] = sum over the LOCAL rows of r[r,j]^2.  Fetches the halo rows of x.
// -1 without touching anything when matrix or operands do not qualify.
extern "C" int gcge_hip_cg_start_mv(void* mat, void** x, int xc0, void** b, int bc0, void** r, void** p0, int rc0, int m,
                                    double* host_rho) {
  enter();
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  GcgeHipMV *vx = (GcgeHipMV*)x, *vb = (GcgeHipMV*)b, *vr = (GcgeHipMV*)r, *vp = (GcgeHipMV*)p0;
  if (A == nullptr || A->d_pid == nullptr || g_spmm_path != 0 || getenv("GCGE_CG_NO_RECOMPUTE") != nullptr) return -1;
  if ((m & 1) || (xc0 & 1) || (bc0 & 1) || (rc0 & 1) || (vx->ld & 1) || (vb->ld & 1) || (vr->ld & 1) || (vp->ld & 1)) return -1;
  if (((uintptr_t)vx->d & 15) || ((uintptr_t)vb->d & 15) || ((uintptr_t)vr->d & 15) || ((uintptr_t)vp->d & 15)) return -1;
  if (vr == vx || vp == vx || (A->nghost > 0 && m > A->buf_cols)) return -1;
  GCGE_REQUIRE(xc0 >= 0 && xc0 + m <= vx->ncols && bc0 >= 0 && bc0 + m <= vb->ncols && rc0 >= 0 && rc0 + m <= vr->ncols &&
               rc0 + m <= vp->ncols, "cg_start: column ranges");
  GCGE_REQUIRE(A->nrows == vx->nrows && A->nrows == vb->nrows && A->nrows == vr->nrows && A->nrows == vp->nrows &&
               A->nrows + A->nghost <= vx->nrows_alloc, "cg_start: shapes");
  double* dd = stage_d(6 * (size_t)m);
  const CgPass cg = {5, vr->d + rc0, vr->ld, vp->d + rc0, vp->ld, nullptr, nullptr, nullptr, vb->d + bc0, vb->ld};
  const int rc = spmm_halo(A, vx, xc0, nullptr, 0, m, dd, nullptr, &cg);
  GCGE_REQUIRE(rc == 0, "cg_start: kernel launch");
  double* hd = stage_h((size_t)m);
  GCGE_HIP_CHECK(hipMemcpyAsync(hd, dd, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  memcpy(host_rho, hd, m * sizeof(double));
  return 0;
}

// The same start for right-hand sides b_j = scale_j x_j (x = the initial guess): the GCG driver's systems
// A w = (lambda + sigma) x start from w = x, so b is never formed and never read (kernel MODE 6).  host_scale: m factors.
extern "C" int gcge_hip_cg_start_scaled_mv(void* mat, void** x, int xc0, const double* host_scale, void** r, void** p0, int rc0,
                                           int m, double* host_rho) {
  enter();
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  GcgeHipMV *vx = (GcgeHipMV*)x, *vr = (GcgeHipMV*)r, *vp = (GcgeHipMV*)p0;
  if (A == nullptr || A->d_pid == nullptr || g_spmm_path != 0 || getenv("GCGE_CG_NO_RECOMPUTE") != nullptr) return -1;
  if ((m & 1) || (xc0 & 1) || (rc0 & 1) || (vx->ld & 1) || (vr->ld & 1) || (vp->ld & 1)) return -1;
  if (((uintptr_t)vx->d & 15) || ((uintptr_t)vr->d & 15) || ((uintptr_t)vp->d & 15)) return -1;
  if (vr == vx || vp == vx || (A->nghost > 0 && m > A->buf_cols)) return -1;
  GCGE_REQUIRE(xc0 >= 0 && xc0 + m <= vx->ncols && rc0 >= 0 && rc0 + m <= vr->ncols && rc0 + m <= vp->ncols, "cg_start: column ranges");
  GCGE_REQUIRE(A->nrows == vx->nrows && A->nrows == vr->nrows && A->nrows == vp->nrows &&
               A->nrows + A->nghost <= vx->nrows_alloc, "cg_start: shapes");
  double* dd = stage_d(7 * (size_t)m);           // [0, 6 m): sums (3 m of scratch behind them when the product is split), [6 m, 7 m): scale
  double* hs = stage_h(2 * (size_t)m);
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));   // the staging buffers are reused
  memcpy(hs, host_scale, m * sizeof(double));
  GCGE_HIP_CHECK(hipMemcpyAsync(dd + 6 * (size_t)m, hs, m * sizeof(double), hipMemcpyHostToDevice, g_stream));
  const CgPass cg = {6, vr->d + rc0, vr->ld, vp->d + rc0, vp->ld, dd + 6 * (size_t)m, nullptr, nullptr, nullptr, 0};
  const int rc = spmm_halo(A, vx, xc0, nullptr, 0, m, dd, nullptr, &cg);
  if (rc != 0) return -1;
  GCGE_HIP_CHECK(hipMemcpyAsync(hs + m, dd, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  memcpy(host_rho, hs + m, m * sizeof(double));
  return 0;
}

// ---- two steps of a V-cycle in one sweep each (GCGE_SetBlockAMGFusions, include/gcge_solver.h; csrc/host/lin_sol.c) ------------
// r[:, rc0:rc0+m) = b[:, bc0:bc0+m) - A x[:, xc0:xc0+m): the start sweep of the block CG (kernel MODE 5) with ONE store — the
// product is rounded on its own and then subtracted from b, exactly what MatDotMultiVec + MultiVecAxpby(1, b, -1, r) leave
// (reference src/ops_lin_sol.c:596-606): 3 block streams instead of 5.  Pattern matrices only; 0 = declined, nothing touched.
static int HIP_AmgResidual(void* mat, void** b, int bc0, void** x, int xc0, void** r, int rc0, int m, struct OPS_* ops) {
  (void)ops;
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  GcgeHipMV *vx = (GcgeHipMV*)x, *vb = (GcgeHipMV*)b, *vr = (GcgeHipMV*)r;
  if (A == nullptr || A->rect_ncols > 0 || A->d_pid == nullptr || g_spmm_path != 0 || m <= 0) return 0;
  if ((m & 1) || (xc0 & 1) || (bc0 & 1) || (rc0 & 1) || (vx->ld & 1) || (vb->ld & 1) || (vr->ld & 1)) return 0;
  if (((uintptr_t)vx->d & 15) || ((uintptr_t)vb->d & 15) || ((uintptr_t)vr->d & 15)) return 0;
  if (vr == vx || vr == vb || (A->nghost > 0 && m > A->buf_cols)) return 0;
  if (xc0 < 0 || xc0 + m > vx->ncols || bc0 < 0 || bc0 + m > vb->ncols || rc0 < 0 || rc0 + m > vr->ncols) return 0;
  if (A->nrows != vx->nrows || A->nrows != vb->nrows || A->nrows != vr->nrows || A->nrows + A->nghost > vx->nrows_alloc) return 0;
  if (real_perm(vx->perm) != real_perm(A->perm) || real_perm(vb->perm) != real_perm(A->perm) || real_perm(vr->perm) != real_perm(A->perm)) return 0;
  enter();
  SlotTimer tm_("AMG residual (fused)", m);
  double* dd = stage_d(6 * (size_t)m);                         // the sweep's column sums |r_j|^2: not used here
  const CgPass cg = {5, vr->d + rc0, vr->ld, vr->d + rc0, vr->ld, nullptr, nullptr, nullptr, vb->d + bc0, vb->ld};
  const int rc = spmm_halo(A, vx, xc0, nullptr, 0, m, dd, nullptr, &cg);
  GCGE_REQUIRE(rc == 0, "AMG residual: kernel launch");
  return 1;
}

// xf[:, f0:f0+m) += P xc[:, c0:c0+m) for a prolongation with ONE entry per row (aggregation: gcge_multigrid.h) — the product is
// rounded, then added (no fused multiply-add), as MatDotMultiVec into a work block + MultiVecAxpby(1, work, 1, xf) do
// (reference src/ops_lin_sol.c:626-640): the fine block is read and written once, the work block not at all.
typedef double v2d_pa __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void prolong_add_kernel(long nrows, const int* __restrict__ colidx, const double* __restrict__ val,
    const double* __restrict__ xc, long ldc, double* __restrict__ xf, long ldf, int m2, int tpr) {
#pragma clang fp contract(off)
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, rpb = 256 / tpr;
  if (tx >= m2) return;
  const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + rpb - 1) / rpb * rpb;
  const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
  for (long row = (long)blockIdx.x * slab + ty; row < rend; row += 2L * rpb) {
    const long row2 = row + rpb;
    const bool two = row2 < rend;
    const long rb = two ? row2 : row;
    const int ca = colidx[row], cb = colidx[rb];
    const double va = val[row], vb = val[rb];
    const v2d_pa ea = *reinterpret_cast<const v2d_pa*>(xc + (long)ca * ldc + 2 * tx);
    const v2d_pa eb = *reinterpret_cast<const v2d_pa*>(xc + (long)cb * ldc + 2 * tx);
    const v2d_pa fa = __builtin_nontemporal_load(reinterpret_cast<const v2d_pa*>(xf + row * ldf + 2 * tx));
    const v2d_pa fb = __builtin_nontemporal_load(reinterpret_cast<const v2d_pa*>(xf + rb * ldf + 2 * tx));
    const v2d_pa ta = {va * ea.x, va * ea.y}, tb = {vb * eb.x, vb * eb.y};
    __builtin_nontemporal_store(v2d_pa{ta.x + fa.x, ta.y + fa.y}, reinterpret_cast<v2d_pa*>(xf + row * ldf + 2 * tx));
    if (two) __builtin_nontemporal_store(v2d_pa{tb.x + fb.x, tb.y + fb.y}, reinterpret_cast<v2d_pa*>(xf + rb * ldf + 2 * tx));
  }
}
static int HIP_AmgProlongAdd(void* matP, void** xc, int c0, void** xf, int f0, int m, struct OPS_* ops) {
  (void)ops;
  GCGE_HIP_MAT_* P = (GCGE_HIP_MAT_*)matP;
  GcgeHipMV *vc = (GcgeHipMV*)xc, *vf = (GcgeHipMV*)xf;
  if (P == nullptr || P->rect_ncols <= 0 || P->rect_one_per_row == 0 || m <= 0 || m / 2 > 256) return 0;
  if ((m & 1) || (c0 & 1) || (f0 & 1) || (vc->ld & 1) || (vf->ld & 1) || ((uintptr_t)vc->d & 15) || ((uintptr_t)vf->d & 15) || vc == vf) return 0;
  if (vc->nrows != P->rect_ncols || vf->nrows != P->nrows || c0 < 0 || c0 + m > vc->ncols || f0 < 0 || f0 + m > vf->ncols) return 0;
  enter();
  SlotTimer tm_("AMG prolongation + correction (fused)", m);
  const int m2 = m / 2;
  int tpr = 1; while (tpr < m2) tpr *= 2;
  const int rpb = 256 / tpr;
  long g = ((long)P->nrows + (long)rpb * 8 - 1) / ((long)rpb * 8); if (g > 8192) g = 8192; if (g < 1) g = 1;
  hipLaunchKernelGGL(prolong_add_kernel, dim3((unsigned)g), dim3(256), 0, g_stream, (long)P->nrows, (const int*)P->d_colidx, (const double*)P->d_val,
                     (const double*)(vc->d + c0), (long)vc->ld, vf->d + f0, (long)vf->ld, m2, tpr);
  GCGE_REQUIRE(hipGetLastError() == hipSuccess, "AMG prolongation + correction: kernel launch");
  return 1;
}

// b[:, bc0:bc0+m) = x[:, xc0:xc0+m) diag(scale): the right-hand sides (lambda_j + sigma) x_j of the GCG driver's W systems for a
// BlockAMG that takes them as scale factors (GCGE_SetBlockAMGFormRhs) — one read, one write, each product rounded once like the
// column scaling after a copy (MatDotMultiVec(B = NULL) + MultiVecLinearComb: reference src/ops_eig_sol_gcg.c:560-577)
__global__ __launch_bounds__(256) void scaled_copy_kernel(long nrows, const double* __restrict__ x, long ldx, double* __restrict__ b, long ldb,
    int m2, const double* __restrict__ scale, int tpr) {
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, rpb = 256 / tpr;
  if (tx >= m2) return;
  const v2d_pa sc = {scale[2 * tx], scale[2 * tx + 1]};
  const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + rpb - 1) / rpb * rpb;
  const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
  long row = (long)blockIdx.x * slab + ty;
  for (; row + 3L * rpb < rend; row += 4L * rpb) {
    v2d_pa a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d_pa*>(x + (row + (long)u * rpb) * ldx + 2 * tx));
#pragma unroll
    for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v2d_pa{a[u].x * sc.x, a[u].y * sc.y}, reinterpret_cast<v2d_pa*>(b + (row + (long)u * rpb) * ldb + 2 * tx));
  }
  for (; row < rend; row += rpb) {
    const v2d_pa a = *reinterpret_cast<const v2d_pa*>(x + row * ldx + 2 * tx);
    *reinterpret_cast<v2d_pa*>(b + row * ldb + 2 * tx) = v2d_pa{a.x * sc.x, a.y * sc.y};
  }
}
static int HIP_AmgFormRhs(void** b, int bc0, void** x, int xc0, const double* scale, int m, struct OPS_* ops) {
  (void)ops;
  GcgeHipMV *vb = (GcgeHipMV*)b, *vx = (GcgeHipMV*)x;
  if (scale == nullptr || m <= 0 || m / 2 > 256 || (m & 1) || (bc0 & 1) || (xc0 & 1) || (vb->ld & 1) || (vx->ld & 1)) return 0;
  if (((uintptr_t)vb->d & 15) || ((uintptr_t)vx->d & 15) || vb->nrows != vx->nrows) return 0;
  if (bc0 < 0 || bc0 + m > vb->ncols || xc0 < 0 || xc0 + m > vx->ncols) return 0;
  if (vb == vx && bc0 < xc0 + m && xc0 < bc0 + m) return 0;
  enter();
  SlotTimer tm_("AMG right-hand sides (x diag(scale))", m);
  double* dd = stage_d((size_t)m);
  double* hs = stage_h((size_t)m);
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));   // the staging buffers are reused
  memcpy(hs, scale, m * sizeof(double));
  GCGE_HIP_CHECK(hipMemcpyAsync(dd, hs, m * sizeof(double), hipMemcpyHostToDevice, g_stream));
  const int m2 = m / 2;
  int tpr = 1; while (tpr < m2) tpr *= 2;
  const int rpb = 256 / tpr;
  long g = ((long)vx->nrows + (long)rpb * 8 - 1) / ((long)rpb * 8); if (g > 8192) g = 8192; if (g < 1) g = 1;
  hipLaunchKernelGGL(scaled_copy_kernel, dim3((unsigned)g), dim3(256), 0, g_stream, (long)vx->nrows, (const double*)(vx->d + xc0), (long)vx->ld,
                     vb->d + bc0, (long)vb->ld, m2, (const double*)dd, tpr);
  GCGE_REQUIRE(hipGetLastError() == hipSuccess, "AMG right-hand sides: kernel launch");
  return 1;
}

// Residuals of Ritz pairs of a standard problem in one read of x (GCGE_RESIDUAL_FN, include/gcge_ops.h; kernel MODE 4 of
// spmm_pattern.hip): res_sq[j] = sum over the local rows of ((A x_j) - lambda_j x_j)^2.  Declines (0) for B != NULL and
// blocks that cannot be walked in 16-byte column pairs; matrices without pattern form take resid_sq_stored above.  Odd
// column ranges are widened to even ones (the extra columns are computed and dropped).
// ... and for the matrices whose product cannot carry the sums (no pattern form: the plane sweep, dense blocks, pad-8): the product
// into a scratch block, then ONE sweep over it and x — 2 block streams behind the product instead of the 9 of the five slot calls
// (round 4; config 5: 13 -> 5 ms per outer iteration).  Chunks of <= 64 columns (the halo buffers' width on slabs).
extern "C" int gcge_hip_resid_sq(int nrows, const double* d_w, long ldw, const double* d_x, long ldx, int m, const double* d_lambda,
                                 double* d_out, void* stream);
// Round 5: the GENERALISED problem (B != NULL; reference src/ops_eig_sol_gcg.c:195-315: A x, B x, lambda B x, the difference, its
// column norms = 11 block streams through five slots) takes the same route with two scratch blocks: A x and B x by the products
// (whatever K1 form each matrix has), then ONE sweep sum_r ((A x)[r,j] - lambda_j (B x)[r,j])^2 over the two — 2 + 2 + 2 streams.
static int resid_sq_stored(GCGE_HIP_MAT_* A, GcgeHipMV* vx, int start, int end, const double* lambda, double* res_sq, GCGE_HIP_MAT_* Bm = nullptr) {
  if (getenv("GCGE_NO_STORED_RESIDUAL_HOOK") != nullptr) return 0;
  if ((vx->ld & 1) || ((uintptr_t)vx->d & 15) || A->nrows + A->nghost > vx->nrows_alloc) return 0;
  if (Bm != nullptr && (Bm->nrows != A->nrows || Bm->rect_ncols > 0 || Bm->nrows + Bm->nghost > vx->nrows_alloc)) return 0;
  const int c0 = start & ~1, c1 = (end + 1) & ~1;
  if (c1 > vx->ld) return 0;
  const int chunk = 64;
  const size_t bytes = (size_t)A->nrows * chunk * sizeof(double);
  double* t = (double*)pool_alloc(bytes);
  double* tb = Bm != nullptr ? (double*)pool_alloc(bytes) : nullptr;
  int ok = 1;
  for (int b0 = c0; b0 < c1 && ok; b0 += chunk) {
    const int m = std::min(chunk, c1 - b0);
    if (A->nghost > 0 && m > A->buf_cols) { ok = 0; break; }
    if (Bm != nullptr && Bm->nghost > 0 && m > Bm->buf_cols) { ok = 0; break; }
    double* dd = stage_d(2 * (size_t)m);
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));   // the pinned staging may still feed an upload of the previous chunk / slot call
    double* hl = stage_h(2 * (size_t)m);
    for (int j = 0; j < m; ++j) hl[j] = (b0 + j >= start && b0 + j < end) ? lambda[b0 + j - start] : 0.0;
    GCGE_HIP_CHECK(hipMemcpyAsync(dd + m, hl, m * sizeof(double), hipMemcpyHostToDevice, g_stream));
    const int rc = spmm_halo(A, vx, b0, t, (long)m, m, nullptr, nullptr);
    GCGE_REQUIRE(rc == 0, "residual norms: product");
    if (Bm != nullptr) GCGE_REQUIRE(spmm_halo(Bm, vx, b0, tb, (long)m, m, nullptr, nullptr) == 0, "residual norms: product with B");
    GCGE_REQUIRE(gcge_hip_resid_sq(A->nrows, t, (long)m, Bm != nullptr ? tb : vx->d + b0, Bm != nullptr ? (long)m : vx->ld, m, dd + m, dd, g_stream) == 0, "residual norms: sweep");
    GCGE_HIP_CHECK(hipMemcpyAsync(hl + m, dd, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
    for (int j = 0; j < m; ++j) if (b0 + j >= start && b0 + j < end) res_sq[b0 + j - start] = hl[m + j];
  }
  pool_free(t, bytes);
  if (tb != nullptr) pool_free(tb, bytes);
  return ok;
}
static int HIP_ResidualSq(void* mat, void* matB, void** x, int start, int end, const double* lambda, double* res_sq) {
  enter();
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat; GcgeHipMV* vx = (GcgeHipMV*)x;
  if (A == nullptr || end <= start) return 0;
  const int c0 = start & ~1, c1 = (end + 1) & ~1, m = c1 - c0;
  if (c1 > vx->ld || A->nrows != vx->nrows) return 0;
  if (matB != nullptr) return getenv("GCGE_NO_GENERAL_RESIDUAL_HOOK") == nullptr ? resid_sq_stored(A, vx, start, end, lambda, res_sq, (GCGE_HIP_MAT_*)matB) : 0;
  if (!gcge_hip_cg_fusable(mat, x, m)) return resid_sq_stored(A, vx, start, end, lambda, res_sq);
  double* dd = stage_d(7 * (size_t)m);
  double* d_lam = dd + 6 * (size_t)m;
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));   // the pinned staging may still feed an upload of the previous slot call
  double* hl = stage_h(2 * (size_t)m);
  for (int j = 0; j < m; ++j) hl[j] = (c0 + j >= start && c0 + j < end) ? lambda[c0 + j - start] : 0.0;
  GCGE_HIP_CHECK(hipMemcpyAsync(d_lam, hl, m * sizeof(double), hipMemcpyHostToDevice, g_stream));
  const CgPass cg = {4, nullptr, 0, nullptr, 0, d_lam, nullptr, nullptr, nullptr, 0};
  const int rc = spmm_halo(A, vx, c0, nullptr, 0, m, dd, nullptr, &cg);
  GCGE_REQUIRE(rc == 0, "residual norms: kernel launch");
  GCGE_HIP_CHECK(hipMemcpyAsync(hl + m, dd, m * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  GCGE_HIP_CHECK(hipStreamSynchronize(g_stream));
  for (int j = start; j < end; ++j) res_sq[j - start] = hl[m + (j - c0)];
  return 1;
}

extern "C" void* gcge_hip_residual_hook(void) { return (void*)HIP_ResidualSq; }   /* for tests */

// app_ccs.c:140-150 — symmetric matrices: the product itself; a rectangular matrix (a prolongation P_l, used transposed as the
// restriction by DefaultMultiVecFromItoJ, src/ops_multi_grid.c:95-113) through the transposed CSR triple kept beside it
static void HIP_MatTransDotMultiVec(void* mat, void** x, void** y, int* start, int* end, struct OPS_* ops) {
  GCGE_HIP_MAT_* A = (GCGE_HIP_MAT_*)mat;
  if (A == nullptr || A->rect_ncols == 0) { HIP_MatDotMultiVec(mat, x, y, start, end, ops); return; }
  GcgeHipMV *vx = (GcgeHipMV*)x, *vy = (GcgeHipMV*)y;
  const int m = end[0] - start[0];
  enter();
  SlotTimer tm_("MatTransDotMultiVec", m);
  GCGE_REQUIRE(m == end[1] - start[1], "MatTransDotMultiVec: equal column counts");
  if (m <= 0) return;
  GCGE_REQUIRE(start[0] >= 0 && end[0] <= vx->ncols && start[1] >= 0 && end[1] <= vy->ncols, "MatTransDotMultiVec: column ranges");
  GCGE_REQUIRE(vx != vy && vx->nrows == A->nrows && vy->nrows == A->rect_ncols, "MatTransDotMultiVec: shapes of a rectangular matrix");
  GCGE_REQUIRE(gcge_hip_csr_spmm(A->rect_ncols, A->d_t_rowptr, A->d_t_colidx, A->d_t_val, vx->d + start[0], vx->ld, vy->d + start[1], vy->ld, m, g_stream) == 0,
               "MatTransDotMultiVec: kernel launch (rectangular matrix)");
}

// Local part + sum over the ranks of the communicator registered at call time (GCGE_GetComm(): RCCL inside the back-end,
// csrc/hip/rccl_comm.hip, or a caller's transport); results packed for the reduction when ld != rows.  Own functions
// (not the Default* of a solver library): whichever OPS_Setup completes the table — ours or the reference's, whose
// default of the same name only reduces under OPS_USE_MPI, src/ops_multi_vec.c:202-230 — finds these slots filled.
static void HIP_MultiVecInnerProd(char nsd, void** x, void** y, int is_vec, int* start, int* end, double* ip, int ldIP,
                                  struct OPS_* ops) {
  if (!local_inner_prod(nsd, x, y, is_vec, start, end, ip, ldIP, ops, 1))
    reduce_inner_prod(nsd, end[0] - start[0], end[1] - start[1], ip, ldIP);
}
static void reduce_inner_prod(char nsd, int nr, int nc, double* ip, int ldIP) {
  GCGE_COMM* comm = GCGE_GetComm();
  if (comm == nullptr || nr <= 0 || nc <= 0) return;
  if (nsd == 'D') nr = 1;   // one value per column, stride ldIP
  if (nr == ldIP) { comm->allreduce_sum(ip, nr * nc, comm->ctx); return; }
  std::vector<double> pack((size_t)nr * nc);
  for (int c = 0; c < nc; ++c) memcpy(pack.data() + (size_t)c * nr, ip + (size_t)c * ldIP, nr * sizeof(double));
  comm->allreduce_sum(pack.data(), nr * nc, comm->ctx);
  for (int c = 0; c < nc; ++c) memcpy(ip + (size_t)c * ldIP, pack.data() + (size_t)c * nr, nr * sizeof(double));
}
// src/ops_multi_vec.c:351-411: qAp = Q[:, s0:e0)^T A P[:, s1:e1); A != NULL leaves A P in mv_ws[:, 0:m) (the
// orthonormalisation re-uses it); 'T' stores the transpose (m x k)
static void HIP_MultiVecQtAP(char ntsA, char ntsd, void** mvQ, void* matA, void** mvP, int is_vec, int* startQP, int* endQP,
                             double* qAp, int ldQAP, void** mv_ws, struct OPS_* ops) {
  int s[2], e[2];
  const int k = endQP[0] - startQP[0], m = endQP[1] - startQP[1];
  if (k <= 0 || m <= 0) return;
  if (matA == nullptr) {
    if (ntsd == 'T') {
      s[0] = startQP[1]; e[0] = endQP[1]; s[1] = startQP[0]; e[1] = endQP[0];
      ops->MultiVecInnerProd('N', mvP, mvQ, is_vec, s, e, qAp, ldQAP, ops);
    } else ops->MultiVecInnerProd(ntsd, mvQ, mvP, is_vec, startQP, endQP, qAp, ldQAP, ops);
    return;
  }
  s[0] = startQP[1]; e[0] = endQP[1]; s[1] = 0; e[1] = m;
  if (ntsA == 'T') ops->MatTransDotMultiVec(matA, mvP, mv_ws, s, e, ops);
  else ops->MatDotMultiVec(matA, mvP, mv_ws, s, e, ops);
  if (ntsd == 'T') {
    s[0] = 0; e[0] = m; s[1] = startQP[0]; e[1] = endQP[0];
    ops->MultiVecInnerProd('N', mv_ws, mvQ, is_vec, s, e, qAp, ldQAP, ops);
  } else {
    s[0] = startQP[0]; e[0] = endQP[0]; s[1] = 0; e[1] = m;
    ops->MultiVecInnerProd(ntsd, mvQ, mv_ws, is_vec, s, e, qAp, ldQAP, ops);
  }
}

extern "C" void OPS_HIP_Set(struct OPS_* ops) {
  if (gcge_hip_init(-1) != 0) {
    fprintf(stderr, "OPS_HIP_Set: HIP back-end unavailable (no GPU): aborting — there is no CPU fallback\n");
    abort();
  }
  ops->Printf                   = DefaultPrintf;
  ops->GetWtime                 = DefaultGetWtime;
  ops->GetOptionFromCommandLine = DefaultGetOptionFromCommandLine;
  ops->MultiVecCreateByMat      = HIP_MultiVecCreateByMat;
  ops->MultiVecCreateByMultiVec = HIP_MultiVecCreateByMultiVec;
  ops->MultiVecDestroy          = HIP_MultiVecDestroy;
  ops->MultiVecView             = HIP_MultiVecView;
  ops->MultiVecLocalInnerProd   = HIP_MultiVecLocalInnerProd;
  ops->MultiVecInnerProd        = HIP_MultiVecInnerProd;   // installed HERE, not left to OPS_Setup: see above
  ops->MultiVecSetRandomValue   = HIP_MultiVecSetRandomValue;
  ops->MultiVecAxpby            = HIP_MultiVecAxpby;
  ops->MultiVecLinearComb       = HIP_MultiVecLinearComb;
  ops->MatDotMultiVec           = HIP_MatDotMultiVec;
  GCGE_SetResidualHook(HIP_ResidualSq, (void*)HIP_MatDotMultiVec);   /* used by our GCG driver for this table only */
  // panel updates work row by row on the row-major blocks (lincomb_mfma.hip: a block / wave reads only the rows it
  // writes, and writes them after its last read): one panel of <= 128 output columns may be updated in place
  GCGE_SetInplaceLinearComb((void*)HIP_MultiVecLinearComb, 128);
  {   // K7 on the device for the projected matrices where the host solver dominates an outer iteration (eig_device.hip)
    GCGE_SetSymEigHook(gcge_hip_symeig, 192, (void*)HIP_MultiVecLinearComb);
  }
  ops->MatTransDotMultiVec      = HIP_MatTransDotMultiVec;
  ops->MultiVecQtAP             = HIP_MultiVecQtAP;
  // the hierarchy behind BlockAMG (src/ops.h:134-139; multigrid.hip) and the fused device CG as its smoother for THIS table
  ops->MultiGridCreate          = gcge_hip_multigrid_create;
  ops->MultiGridDestroy         = gcge_hip_multigrid_destroy;
  GCGE_SetBlockAMGSmoother(gcge_hip_amg_smoother_setup, gcge_hip_amg_smoother_residual, (void*)HIP_MatDotMultiVec);
  GCGE_SetBlockAMGFusions(HIP_AmgResidual, HIP_AmgProlongAdd, (void*)HIP_MatDotMultiVec);   // r = b - A x and x += P e as one sweep each
  GCGE_SetBlockAMGFormRhs(HIP_AmgFormRhs, (void*)HIP_MatDotMultiVec);                         // b = x diag(scale) in one sweep
}
