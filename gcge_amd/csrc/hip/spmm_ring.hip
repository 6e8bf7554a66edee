// K1 / K3 (pattern path, read-only passes) — the chain + line-exchange sweep of spmm_pattern.hip with every X row
// arriving through an LDS ring filled by LDS-DMA (global_load_lds_dwordx4), several grid planes ahead.
//
// Why: spmm_pattern_chain2_kernel keeps ONE plane of rows in flight per wave (the "+S" row is requested one
// iteration before it is reduced, 16 waves = one block per CU), so a read-only pass moves 16 KB per CU per HBM round
// trip: 256 iterations x 2 rounds of blocks in 602 us = 1.18 us per iteration, the memory latency itself
// (profiles/r01_bench: first CG pass 3.6 TB/s on the bytes it needs).  More rows in flight cost 4 VGPRs per row and
// plane there (profiles/r02_cg: the register ring spilled / lost occupancy).  Here the rows in flight live in LDS:
//   * plane k of the ring holds, per wave (= grid line) of the block, the 8 x 128 B of X rows the wave reduces in
//     iteration k, plus the +-L rows outside the block (lowest / highest wave) and the 2 fringe rows (x - 1 of the
//     slice's first row, x + 1 of its last) — 22 KB per plane at 16 waves;
//   * the wave asks for plane j + DP + 1 while it reduces plane j: DP + 1 planes in flight, no VGPR spent on them;
//   * all vector-memory operations of the loop are LDS-DMA issued from inline asm, so the ONLY vmcnt waits are the
//     counted ones written here (hipcc would drain the queue with vmcnt(0) at the first use of an ordinary load);
//     the pattern ids of a slice (8 x 16 bit) travel the same way, 2 DP + 2 iterations ahead, into a small id ring
//     (through the scalar cache they cost one exposed miss per iteration: lgkmcnt cannot be counted past an s_load);
//   * the -S / centre rows of a lane stay in registers (chain), +S, +-L and +-1 are LDS reads: the +-1 rows of the
//     7-point stencil are the centre rows of the lanes 8 to the left / right, so no load at all for 6 of 8 rows.
// Scope: tables of 7 slots laid out as [-S, 0, +S, -L, +L, -1, +1] (app_hip.hip build_patterns: pat_near), the modes
// that store nothing (2: first pass of the block CG; 4: residual norms).  Everything else keeps spmm_pattern.hip.
// Reference operation: app/app_ccs.c:50-139 (MatDotMultiVec) inside src/ops_lin_sol.c:256-405 (BlockPCG).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "gcge_hip_internal.h"

extern "C" double* gcge_hip_partial_ws(size_t len);
extern "C" void gcge_hip_reduce_partials16(const double* d_partial, int nblocks, long slab_stride, int ncols, double* d_out,
                                           void* stream);

namespace gcge_ring {

typedef double v2d __attribute__((ext_vector_type(2)));
struct PatEntry { double val; long off; };
constexpr int LT = 7;

template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// one LDS-DMA piece: 64 lanes x 16 B, lane l lands at lds_dst + 16 l (lds_dst wave-uniform); counted by vmcnt
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// 4 lanes x 4 B: the 8 ids of a slice into LDS (lane l lands at lds_dst + 4 l)
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The pieces of one batch in ONE statement (M0 saved once, the partial pieces under an exec mask set here), addressed as
// scalar base + 32-bit lane offset: sb = X + (first row of the slice) - 2 GiB, sq = the slice's ids; the lane offsets
// (2 GiB + row / column / stencil offset) change only when the lane's pattern does, so a batch costs no vector ALU work:
//   rows (64 lanes x 16 B) -> lm;  ROLE != 1: the -L / +L row -> le;  lanes 0-15: the two fringe rows -> lf;
//   lanes 0-3: the 8 ids of a later slice (4 B each) -> lq
template <int ROLE>
__device__ __forceinline__ void batch_pieces(const char* sb, const char* sq, unsigned vm, unsigned ve, unsigned vf, unsigned vq,
                                             unsigned lm, unsigned le, unsigned lf, unsigned lq) {
  unsigned keep; unsigned long sv;
  if (ROLE == 1)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lm]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vm], %[sb]\n\t"
                 "s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %[lf]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vf], %[sb]\n\t"
                 "s_mov_b64 exec, 0xf\n\ts_mov_b32 m0, %[lq]\n\ts_nop 0\n\tglobal_load_lds_dword %[vq], %[sq]\n\t"
                 "s_mov_b64 exec, %[sv]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [sv] "=&s"(sv)
                 : [sb] "s"(sb), [sq] "s"(sq), [vm] "v"(vm), [vf] "v"(vf), [vq] "v"(vq), [lm] "s"(lm), [lf] "s"(lf), [lq] "s"(lq)
                 : "memory");
  else
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lm]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vm], %[sb]\n\t"
                 "s_mov_b32 m0, %[le]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[ve], %[sb]\n\t"
                 "s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %[lf]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vf], %[sb]\n\t"
                 "s_mov_b64 exec, 0xf\n\ts_mov_b32 m0, %[lq]\n\ts_nop 0\n\tglobal_load_lds_dword %[vq], %[sq]\n\t"
                 "s_mov_b64 exec, %[sv]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [sv] "=&s"(sv)
                 : [sb] "s"(sb), [sq] "s"(sq), [vm] "v"(vm), [ve] "v"(ve), [vf] "v"(vf), [vq] "v"(vq), [lm] "s"(lm), [le] "s"(le),
                   [lf] "s"(lf), [lq] "s"(lq)
                 : "memory");
}

// The same with 64-bit lane addresses (tables whose byte offsets do not fit the 32-bit lane offset: WIDE):
//   rows (64 lanes x 16 B) -> lm;  ROLE != 1: the -L / +L row -> le;  lanes 0-15: the two fringe rows -> lf;
//   lanes 0-3: the 8 ids of a later slice (4 B each) -> lq
template <int ROLE>
__device__ __forceinline__ void batch_pieces64(const void* am, const void* ae, const void* af, const void* aq,
                                             unsigned lm, unsigned le, unsigned lf, unsigned lq) {
  unsigned keep; unsigned long sv;
  if (ROLE == 1)
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lm]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[am], off\n\t"
                 "s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %[lf]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[af], off\n\t"
                 "s_mov_b64 exec, 0xf\n\ts_mov_b32 m0, %[lq]\n\ts_nop 0\n\tglobal_load_lds_dword %[aq], off\n\t"
                 "s_mov_b64 exec, %[sv]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [sv] "=&s"(sv)
                 : [am] "v"(am), [af] "v"(af), [aq] "v"(aq), [lm] "s"(lm), [lf] "s"(lf), [lq] "s"(lq) : "memory");
  else
    asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[lm]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[am], off\n\t"
                 "s_mov_b32 m0, %[le]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[ae], off\n\t"
                 "s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %[lf]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[af], off\n\t"
                 "s_mov_b64 exec, 0xf\n\ts_mov_b32 m0, %[lq]\n\ts_nop 0\n\tglobal_load_lds_dword %[aq], off\n\t"
                 "s_mov_b64 exec, %[sv]\n\ts_mov_b32 m0, %[k]"
                 : [k] "=&s"(keep), [sv] "=&s"(sv)
                 : [am] "v"(am), [ae] "v"(ae), [af] "v"(af), [aq] "v"(aq), [lm] "s"(lm), [le] "s"(le), [lf] "s"(lf), [lq] "s"(lq)
                 : "memory");
}

// LDS copy of the table, 128 B per pattern: 7 values (+ pad), then the 7 offsets as BYTE distances in X (off * ldx * 8)
struct RPat { double val[8]; long offb[8]; };

// bytes per plane: NW + 2 lines of 1 KB, NW fringes of 16 x 16 B; behind the planes a ring of RI x 16 B per wave with the
// pattern ids of its slices (8 x 16 bit each)
template <int NW> struct Plane { static constexpr unsigned FR = (NW + 2) * 1024, PB = FR + NW * 256; };
constexpr int RI = 16;

// MODE 2 / 4 as in spmm_pattern.hip.  ROLE: 0 lowest wave, 1 inner, 2 highest (they differ in the pieces per plane).
// Geometry (host-checked): tile t of the block = tile b0 + t G of the sweep, and consecutive tiles of a block lie
// `step_rows` rows apart (one grid plane; 0 when every block has a single tile).
template <int MODE, int NW, int DP, int ROLE, bool WIDE>
__device__ __forceinline__ void ring_body(
    long nrows, const unsigned short* __restrict__ pid, const RPat* s_pat, char* ring, const double* __restrict__ xblk,
    const double* __restrict__ xl, size_t ldx, bool act, int i, int g, int wave, int lane, int ntiles, long line, long step_rows, int b0,
    double& d0, double& d1, double& e0, double& e1, const v2d* s_cf, double* __restrict__ y, size_t ldy) {
  constexpr int R = DP + 3;                                  // planes in the ring
  constexpr unsigned PB = Plane<NW>::PB;
  constexpr int K = (ROLE == 1) ? 3 : 4;                     // LDS-DMA pieces per batch
  constexpr int S = (MODE <= 1) ? 1 : 0;                     // stores per iteration (they share vmcnt with the pieces, in order)
  static_assert(2 * DP + 5 <= RI, "id ring");
  const unsigned ring0 = (unsigned)(uintptr_t)ring;          // LDS byte address (low half of the flat address)
  const int G = gridDim.x;
  const int cnt = __builtin_amdgcn_readfirstlane((ntiles - b0 + G - 1) / G);
  const long ldxb = (long)ldx * 8;
  // rows of this wave: slice at rb0 + k * step_rows, k = 0 .. cnt - 1; the slices beyond the matrix (k > kv) and the
  // surplus batches (k >= cnt) re-read the last valid slice (weights 0): every address stays inside the block of vectors
  const int asl = (int)(line / 8), q0 = b0 / asl;
  long rb0 = ((long)NW * q0 + wave) * line + 8 * (b0 - q0 * asl);
  int kv = -1;
  if (rb0 <= nrows - 8) kv = step_rows > 0 ? (int)((nrows - 8 - rb0) / step_rows) : 0x7fffffff;
  else rb0 = nrows - 8;
  kv = __builtin_amdgcn_readfirstlane(kv);
  const int kmax = kv < 0 ? 0 : (kv < cnt - 1 ? kv : cnt - 1);   // last k that moves the walkers
  const long stepB = step_rows * ldxb, stepI = step_rows * 2;
  long rowsB = rb0 * ldxb, idsB = rb0 * 2;                   // byte offsets: rows being requested (in X), ids being requested (in pid)
  int ki = 0, kq = 0;
  // per-lane byte offsets inside a plane
  const unsigned o_own = (1 + wave) * 1024 + lane * 16, o_lo = wave * 1024 + lane * 16, o_hi = (2 + wave) * 1024 + lane * 16;
  const unsigned o_fr = Plane<NW>::FR + wave * 256;
  const unsigned o_m1 = (g >= 1) ? o_own - 128 : o_fr + i * 16;
  const unsigned o_p1 = (g <= 6) ? o_own + 128 : o_fr + 128 + i * 16;
  const unsigned idr = R * PB + wave * (RI * 16);            // this wave's id ring
  const unsigned o_idg = idr + g * 2, o_idf = idr + (lane < 8 ? 0 : 14);
  const int fslot = lane < 8 ? 5 : 6;
  const char* xg = reinterpret_cast<const char*>(xl + (size_t)g * ldx);                       // row g of a slice at row 0
  const char* xf = reinterpret_cast<const char*>(xl + (size_t)(lane < 8 ? 0 : 7) * ldx);      // fringe lanes: its first / last row
  const char* pidl = reinterpret_cast<const char*>(pid) + 4 * (lane & 3);
  auto id_at = [&](unsigned o) { return (int)*reinterpret_cast<const unsigned short*>(ring + o); };

  // ---- prologue: ids of iterations 0 .. DP + 1; planes -1 (slot R-1, own line only) and 0; the batches of 0 .. DP
  int sq = 0;   // id-ring entry being requested (byte offset)
#pragma unroll
  for (int d = 0; d <= DP + 1; ++d) {
    if (lane < 4) glds4(pidl + idsB, ring0 + idr + sq);
    sq = (sq + 16) & (RI * 16 - 1);
    if (kq < kmax) { idsB += stepI; } ++kq;
  }
  vm_wait<0>();
  {
    const RPat* e = s_pat + id_at(o_idg);
    glds16(xg + rowsB + e->offb[0], ring0 + (R - 1) * PB + (1 + wave) * 1024);
    glds16(xg + rowsB + e->offb[1], ring0 + (1 + wave) * 1024);
  }
  unsigned si = 0, ei = 0;   // plane slot / id entry of the batch being issued (byte offsets)
  // The byte offsets a batch needs (slot 2 of the row's pattern, slot 3 / 4 for the outer waves, slot 5 / 6 of the slice's
  // first / last row for the fringe lanes) stay in registers and are looked up again only by the lanes whose pattern
  // changed: along a sweep (one (x, y) position, plane after plane) a lane's pattern changes at the first and last plane
  // only, so the look-ups (LDS bandwidth is what bounds this kernel) all but disappear.
  int pb_c = -1, pbf_c = -1;
  unsigned vo_s = 0, vo_e = 0, vo_f = 0;                      // lane offsets of the row, edge and fringe pieces (see batch_pieces)
  long ob_s = 0, ob_e = 0, ob_f = 0;                          // WIDE: the byte offsets themselves
  const long lane0 = (1L << 31) + (xl - xblk) * 8;            // 2 GiB bias + this lane's column pair
  const unsigned vo_g = (unsigned)(lane0 + g * ldxb), vo_x = (unsigned)(lane0 + (lane < 8 ? 0 : 7) * ldxb), vo_q = 4 * (lane & 3);
  const char* xb = reinterpret_cast<const char*>(xblk) - (1L << 31);
  auto batch = [&](int pb, int pbf) {
    if (pb != pb_c) {
      const RPat* e = s_pat + pb;
      ob_s = e->offb[2]; if (ROLE != 1) ob_e = e->offb[ROLE == 0 ? 3 : 4];
      vo_s = vo_g + (unsigned)ob_s; vo_e = vo_g + (unsigned)ob_e;
      pb_c = pb;
    }
    if (pbf != pbf_c) { ob_f = s_pat[pbf].offb[fslot]; vo_f = vo_x + (unsigned)ob_f; pbf_c = pbf; }
    const unsigned si1 = si + PB == R * PB ? 0 : si + PB;
    const unsigned lm = ring0 + si1 + (1 + wave) * 1024, le = ring0 + si + (ROLE == 0 ? 0 : (NW + 1) * 1024);
    if (WIDE) batch_pieces64<ROLE>(xg + rowsB + ob_s, xg + rowsB + ob_e, xf + rowsB + ob_f, pidl + idsB, lm, le, ring0 + si + o_fr, ring0 + idr + sq);
    else batch_pieces<ROLE>(xb + rowsB, reinterpret_cast<const char*>(pid) + idsB, vo_s, vo_e, vo_f, vo_q, lm, le, ring0 + si + o_fr, ring0 + idr + sq);
    si = si1; ei = (ei + 16) & (RI * 16 - 1); sq = (sq + 16) & (RI * 16 - 1);
    if (ki < kmax) { rowsB += stepB; } ++ki;
    if (kq < kmax) { idsB += stepI; } ++kq;
  };
#pragma unroll
  for (int d = 0; d <= DP; ++d) {
    const int pb = id_at(o_idg + ei), pbf = id_at(o_idf + ei);
    batch(pb, pbf);
  }
  // outstanding: 2 + (DP + 1) K pieces
  vm_wait<DP * K>();                     // planes -1, 0 and batch 0 (plane 1, fringe / edge 0, ids DP + 2) have landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  v2d a = *reinterpret_cast<const v2d*>(ring + (R - 1) * PB + o_own);
  v2d b = *reinterpret_cast<const v2d*>(ring + o_own);
  unsigned sf = 0, ef = 0;               // plane slot / id entry of the iteration being reduced (byte offsets)
  int pfin = id_at(o_idg), pb = id_at(o_idg + ei), pbf = id_at(o_idf + ei);   // patterns: rows reduced now / requested now
  const double wact = act ? 1.0 : 0.0;
  // MODE 0 / 1: this lane's place in Y (the host only takes these modes when every slice of the sweep is inside the
  // matrix: each wave then issues exactly one store per iteration, which the counted waits below rely on)
  char* yl = MODE <= 1 ? reinterpret_cast<char*>(y + (size_t)(rb0 + g) * ldy + 2 * i) : nullptr;
  const long stepY = step_rows * (long)ldy * 8;
  int pfin_c = -1;            // the pattern whose values sit in val[] (same caching as in batch)
  double val[LT];
#pragma unroll
  for (int t = 0; t < LT; ++t) val[t] = 0.0;
  for (int j = 0; j < cnt; ++j) {
    // ---- everything iteration j reads from LDS, up front: the rows, the table entries, the ids of the NEXT iteration
    const unsigned sf1 = sf + PB == R * PB ? 0 : sf + PB;
    const char* pj = ring + sf;
    const v2d c  = *reinterpret_cast<const v2d*>(ring + sf1 + o_own);
    const v2d vm = *reinterpret_cast<const v2d*>(pj + o_lo);
    const v2d vp = *reinterpret_cast<const v2d*>(pj + o_hi);
    const v2d m1 = *reinterpret_cast<const v2d*>(pj + o_m1);
    const v2d p1 = *reinterpret_cast<const v2d*>(pj + o_p1);
    if (pfin != pfin_c) {
      const RPat* e = s_pat + pfin;
#pragma unroll
      for (int t = 0; t < LT; ++t) val[t] = e->val[t];
      pfin_c = pfin;
    }
    ef = (ef + 16) & (RI * 16 - 1);
    const int pfin_n = id_at(o_idg + ef);
    double a0 = val[0] * a.x, a1 = val[0] * a.y;
    a0 = fma(val[1], b.x, a0);  a1 = fma(val[1], b.y, a1);
    a0 = fma(val[2], c.x, a0);  a1 = fma(val[2], c.y, a1);
    a0 = fma(val[3], vm.x, a0); a1 = fma(val[3], vm.y, a1);
    a0 = fma(val[4], vp.x, a0); a1 = fma(val[4], vp.y, a1);
    a0 = fma(val[5], m1.x, a0); a1 = fma(val[5], m1.y, a1);
    a0 = fma(val[6], p1.x, a0); a1 = fma(val[6], p1.y, a1);
    const double wgt = j <= kv ? wact : 0.0;                  // j < cnt by the loop bound
    if (MODE <= 1) {
      if (act) { v2d o = {a0, a1}; __builtin_nontemporal_store(o, reinterpret_cast<v2d*>(yl)); }
      yl += stepY;
    }
    if (MODE == 1 || MODE == 2) {
      d0 = fma(a0 * wgt, b.x, d0); d1 = fma(a1 * wgt, b.y, d1);
      e0 = fma(a0 * wgt, a0, e0);  e1 = fma(a1 * wgt, a1, e1);
    } else if (MODE == 4) {
      const v2d lam = s_cf[i];
      const double q0 = fma(-lam.x, b.x, a0), q1 = fma(-lam.y, b.y, a1);
      d0 = fma(q0 * wgt, q0, d0); d1 = fma(q1 * wgt, q1, d1);
    }
    a = b; b = c; sf = sf1; pfin = pfin_n;
    // ---- request the batch of iteration j + 1 + DP (plane j + DP + 2 = the slot plane j - 1 leaves) and the ids of
    //      iteration j + 2 DP + 3 (read one iteration before their batch goes out)
    batch(pb, pbf);
    pb = id_at(o_idg + ei); pbf = id_at(o_idf + ei);
    // everything up to the batch of iteration j + 1 has landed: behind it in the queue are DP batches and, once the loop
    // has run DP times, DP x S stores (before that fewer: the stricter count is the safe one)
    if (S == 0 || j < DP) vm_wait<DP * K>(); else vm_wait<DP * (K + S)>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS reads above are done before anybody overwrites them
    __builtin_amdgcn_s_barrier();
  }
  vm_wait<0>();               // no LDS-DMA may land after the block has released its LDS
}

template <int MODE, int NW, int DP, bool WIDE>
__global__ __launch_bounds__(64 * NW) void spmm_ring_kernel(
    long nrows, const unsigned short* __restrict__ pid, const PatEntry* __restrict__ tab, int ntab,
    const double* __restrict__ x, size_t ldx, int m, int ntiles, long line, long step_rows, int xcd_runs,
    double* __restrict__ dot_partial, long yy_offset, const double* __restrict__ lambda, double* __restrict__ y, size_t ldy) {
  static_assert(MODE == 0 || MODE == 1 || MODE == 2 || MODE == 4, "product (with sums) and the passes that store nothing");
  // gridDim.y > 1: the 16-column passes of one operation in one launch (spmm_pattern.hip, g_pass_merge_blocks): pass blockIdx.y
  // works on columns [16 y, 16 y + 16) of X / Y / lambda and on its own slab of partial sums, m = all the columns
  if (gridDim.y > 1) {
    const int c0 = 16 * (int)blockIdx.y;
    x += c0; if (y != nullptr) y += c0; if (lambda != nullptr) lambda += c0;
    if (dot_partial != nullptr) dot_partial += (size_t)blockIdx.y * gridDim.x * 16;
    m = min(m - c0, 16);
  }
  constexpr int R = DP + 3;
  constexpr unsigned PB = Plane<NW>::PB;
  extern __shared__ __align__(16) unsigned char smem_raw[];   // ONE LDS object: ring | table | coefficients | reduction
  char* ring = reinterpret_cast<char*>(smem_raw);
  RPat* s_pat = reinterpret_cast<RPat*>(smem_raw + R * PB + NW * RI * 16);
  v2d* s_cf = reinterpret_cast<v2d*>(smem_raw + R * PB + NW * RI * 16 + (size_t)(ntab / LT) * sizeof(RPat));
  double (*sred)[32] = reinterpret_cast<double (*)[32]>(s_cf + 8);
  for (int e = threadIdx.x; e < ntab; e += 64 * NW) {
    const int p = e / LT, t = e - p * LT;
    s_pat[p].val[t] = tab[e].val; s_pat[p].offb[t] = tab[e].off * (long)ldx * 8;
  }
  if (MODE == 4 && threadIdx.x < 8)
    s_cf[threadIdx.x] = (2 * (int)threadIdx.x < m) ? v2d{lambda[2 * threadIdx.x], lambda[2 * threadIdx.x + 1]} : v2d{0.0, 0.0};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 3, i = lane & 7;
  const bool act = 2 * i < m;
  const double* __restrict__ xl = x + (act ? 2 * i : 0);
  double d0 = 0.0, d1 = 0.0, e0 = 0.0, e1 = 0.0;
  // Blocks are dealt round-robin to the 8 XCDs.  xcd_runs: XCD k takes the k-th eighth of the sweep's tiles, so the
  // slices that share fringe rows (neighbours along a grid line) and the line groups that share +-L rows meet in ONE L2
  const int G = gridDim.x;
  const int bx = blockIdx.x;
  const int b0 = (xcd_runs == 2 && G % 32 == 0) ? ((bx & ~31) | ((bx & 7) << 2) | ((bx >> 3) & 3))   // runs of 4 tiles, see spmm_pattern.hip
               : (xcd_runs == 1 && G % 8 == 0) ? (bx & 7) * (G >> 3) + (bx >> 3) : bx;
  if (b0 < ntiles) {   // block-uniform
    if (wave == 0) ring_body<MODE, NW, DP, 0, WIDE>(nrows, pid, s_pat, ring, x, xl, ldx, act, i, g, wave, lane, ntiles, line, step_rows, b0, d0, d1, e0, e1, s_cf, y, ldy);
    else if (wave == NW - 1) ring_body<MODE, NW, DP, 2, WIDE>(nrows, pid, s_pat, ring, x, xl, ldx, act, i, g, wave, lane, ntiles, line, step_rows, b0, d0, d1, e0, e1, s_cf, y, ldy);
    else ring_body<MODE, NW, DP, 1, WIDE>(nrows, pid, s_pat, ring, x, xl, ldx, act, i, g, wave, lane, ntiles, line, step_rows, b0, d0, d1, e0, e1, s_cf, y, ldy);
  }
  if (MODE == 0) return;   // no column sums
  auto sx = [](double v, int mask) {
    int lo = __shfl_xor(__double2loint(v), mask, 64), hi = __shfl_xor(__double2hiint(v), mask, 64);
    return __hiloint2double(hi, lo);
  };
  d0 += sx(d0, 8);  d1 += sx(d1, 8);  e0 += sx(e0, 8);  e1 += sx(e1, 8);
  d0 += sx(d0, 16); d1 += sx(d1, 16); e0 += sx(e0, 16); e1 += sx(e1, 16);
  d0 += sx(d0, 32); d1 += sx(d1, 32); e0 += sx(e0, 32); e1 += sx(e1, 32);
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

}  // namespace gcge_ring

using namespace gcge_ring;

static int g_ring_on = 1;      // 0: keep spmm_pattern_chain2_kernel for the read-only passes (tuning / A-B measurements)
static int g_ring_depth = 3;   // planes requested ahead (2 or 3)
// Y = A X through the ring: off by default — measured 3.44 ms against 3.43-3.46 ms of the chain2 kernel at 256^3 x 64 (the
// product is bound by its read + write traffic, 18.2 GB at 5.3 TB/s, not by rows in flight); kept, and tested, because it
// shows the counted waits with stores in the queue (what a ring form of the second CG pass would need)
static int g_ring_product = 0;
extern "C" void gcge_hip_spmm_ring_product(int on) { g_ring_product = on; }
static int g_ring_wide = 0;    // 1: always 64-bit lane addresses (tests)
extern "C" void gcge_hip_spmm_ring_wide(int on) { g_ring_wide = on; }
static int g_ring_xcd = 2;     // 2: runs of 4 neighbouring tiles per XCD (1.75 -> 1.71 ms); 1: one contiguous eighth per XCD; 0: block order
extern "C" void gcge_hip_spmm_ring_xcd(int on) { g_ring_xcd = on; }
extern "C" void gcge_hip_spmm_ring_tune(int on, int depth) { g_ring_on = on; if (depth == 2 || depth == 3) g_ring_depth = depth; }

template <int MODE, int NW, int DP, bool WIDE>
static int ring_launch(long nb, hipStream_t st, long nrows, const unsigned short* pid, const void* tab, int ntab, const double* x,
                       size_t ldx, int m, long ntl, long line, double* part, long yyo, const double* lambda, double* y, size_t ldy, int gy) {
  constexpr unsigned PB = Plane<NW>::PB;
  if (ntl > 0x7fffffffL) return -1;
  // tiles of a block: b0, b0 + nb, ... — one grid plane apart when nb is a whole number of line groups
  const long asl = line / 8;
  long step_rows = 0;
  if (nb < ntl) { if (nb % asl) return -1; step_rows = nb / asl * NW * line; }
  const size_t lds = (size_t)(DP + 3) * PB + (size_t)NW * RI * 16 + (size_t)(ntab / LT) * sizeof(RPat) + 8 * sizeof(v2d) + (size_t)NW * 32 * sizeof(double);
  if (lds > 160 * 1024) return -1;
  static size_t granted = 0;   // per instantiation
  if (lds > granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_ring_kernel<MODE, NW, DP, WIDE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return -1; }
    granted = lds;
  }
  hipLaunchKernelGGL((spmm_ring_kernel<MODE, NW, DP, WIDE>), dim3((unsigned)nb, (unsigned)gy), dim3(64 * NW), lds, st, nrows, pid,
                     (const PatEntry*)tab, ntab, x, ldx, m, (int)ntl, line, step_rows, g_ring_xcd, part, yyo, lambda, y, ldy);
  return 0;
}

static long g_ring_launches = 0;
extern "C" long gcge_hip_spmm_ring_launches(void) { return g_ring_launches; }

template <int MODE, int NW, bool WIDE>
static int ring_pass_mode(long nb, hipStream_t st, long nrows, const unsigned short* pid, const void* tab, int ntab, const double* x,
                          size_t ldx, int m, long ntl, long L, double* part, long yyo, const double* lambda, double* y, size_t ldy, int gy) {
  int rc = -1;
  if (g_ring_depth == 3) rc = ring_launch<MODE, NW, 3, WIDE>(nb, st, nrows, pid, tab, ntab, x, ldx, m, ntl, L, part, yyo, lambda, y, ldy, gy);
  if (rc != 0) rc = ring_launch<MODE, NW, 2, WIDE>(nb, st, nrows, pid, tab, ntab, x, ldx, m, ntl, L, part, yyo, lambda, y, ldy, gy);
  return rc;
}
template <int NW, bool WIDE>
static int ring_pass_nw(int mode, long nb, hipStream_t st, long nrows, const unsigned short* pid, const void* tab, int ntab,
                        const double* x, size_t ldx, int m, long ntl, long L, double* part, long yyo, const double* lambda,
                        double* y, size_t ldy, int gy) {
  switch (mode) {
    case 0: return ring_pass_mode<0, NW, WIDE>(nb, st, nrows, pid, tab, ntab, x, ldx, m, ntl, L, nullptr, 0, nullptr, y, ldy, gy);
    case 1: return ring_pass_mode<1, NW, WIDE>(nb, st, nrows, pid, tab, ntab, x, ldx, m, ntl, L, part, yyo, nullptr, y, ldy, gy);
    case 2: return ring_pass_mode<2, NW, WIDE>(nb, st, nrows, pid, tab, ntab, x, ldx, m, ntl, L, part, yyo, nullptr, nullptr, 0, gy);
    case 4: return ring_pass_mode<4, NW, WIDE>(nb, st, nrows, pid, tab, ntab, x, ldx, m, ntl, L, part, yyo, lambda, nullptr, 0, gy);
  }
  return -1;
}

// One 16-column launch of the ring sweep on a [-S, 0, +S, -L, +L, -1, +1] table; same geometry (nb blocks of nw waves,
// lines of L rows) and the same partial-sum workspace as the chain2 kernel.  mode 2 / 4: the passes that store nothing
// (gcge_hip_pattern_cg); mode 0 / 1: Y = A X (with the column sums) — only when every slice of the sweep lies inside the
// matrix (the counted waits rely on one store per wave and iteration).  -1: not applicable, the caller goes on.
// maxoff: the largest |column offset| (rows) in the table: decides between 32-bit lane offsets and 64-bit addresses.
extern "C" int gcge_hip_ring_pass(int mode, int nrows, const unsigned short* d_pid, const void* d_tab, int npat, long L, int nw,
                                  long nb, const double* d_x, long ldx, int m, double* part, long yyo,
                                  const double* d_lambda, void* stream, long maxoff, double* d_y, long ldy, int gy) {
  if (gy < 1) gy = 1;
  if (!g_ring_on || (mode != 0 && mode != 1 && mode != 2 && mode != 4) || (nrows & 7) || nrows < 8 || ((uintptr_t)d_pid & 15) || L % 8 || L < 8) return -1;
  if (((uintptr_t)d_x & 15) || (ldx & 1)) return -1;
  const long nlines = ((long)nrows + L - 1) / L, ntl = (nlines + nw - 1) / nw * (L / 8);
  if (mode <= 1) {
    if (!g_ring_product || d_y == nullptr || ((uintptr_t)d_y & 15) || (ldy & 1) || ntl * nw * 8 != (long)nrows) return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ntab = npat * LT;
  int rc = -1;
  // lane offset = 2 GiB + (row in slice, column pair) + stencil offset, as an unsigned 32-bit number
  const bool wide = g_ring_wide || (double)(maxoff + 16) * (double)ldx * 8.0 >= 2147483648.0 - 4096.0;
#define GCGE_RING_NW(N) (wide ? ring_pass_nw<N, true>(mode, nb, st, nrows, d_pid, d_tab, ntab, d_x, (size_t)ldx, m, ntl, L, part, yyo, d_lambda, d_y, (size_t)ldy, gy) \
                              : ring_pass_nw<N, false>(mode, nb, st, nrows, d_pid, d_tab, ntab, d_x, (size_t)ldx, m, ntl, L, part, yyo, d_lambda, d_y, (size_t)ldy, gy))
  if (nw == 16) rc = GCGE_RING_NW(16);
  else if (nw == 8) rc = GCGE_RING_NW(8);
  else if (nw == 4) rc = GCGE_RING_NW(4);
#undef GCGE_RING_NW
  if (rc == 0) g_ring_launches += gy;   // column passes that took the ring (a merged launch carries gy of them)
  return rc;
}
