// K5 — block conjugate gradients with the vector work fused on the device.
//
// Same recurrence, stopping rules and per-column retirement as the reference's
// BlockPCG (src/ops_lin_sol.c:140-437) — every right-hand side has its own alpha,
// beta, rho and leaves the iteration on its own — but instead of ~3 b single-column
// MultiVecAxpby calls + 2 inner products per iteration an iteration
// is two launches on whole blocks (one-pass scheme, the default):
//     w = A p ; pTw_j = p_j . w_j ; wTw_j = w_j . w_j     (K1 SpMM, both column sums fused in its epilogue)
//     r -= alpha_j w ; x += alpha_j p ; p = r + beta_j p ; rho_j = r_j . r_j
//                                                          (cg_update_all: 4 reads + 3 writes)
// with beta_j = rho_pred_j / rho_j, rho_pred = alpha^2 wTw - rho (what r_new . r_new is in exact arithmetic for CG
// directions); the measured rho_j of the sweep drives the next alpha and the stopping test.  7 block streams + the
// SpMM instead of the 13 + SpMM of the unfused recurrence.  When memory allows, the x update is taken out of the
// sweep as well: the directions go into a ring of up to 16 blocks (cg_update_rp: 3 reads + 2 writes) and x is
// brought up to date every 15 iterations (cg_accum_x): 6.1 block streams per iteration + the SpMM's two.
// Recompute form (pattern matrices in chain + line-exchange layout, no shift, ring available): w is never stored.
//     pass 1: pTw_j, wTw_j from one read of p                       (spmm_pattern.hip MODE 2, gcge_hip_cg_pass1_mv)
//     pass 2: w rebuilt in the SpMM kernel's registers, r -= alpha_j w ; p' = r + beta_j p ; rho_j
//             (reads p, r; writes r, p' into the next ring slot)    (MODE 3, gcge_hip_cg_pass2_mv)
// 1 + 4 + 1.1 = 6.1 block streams per iteration in total; the start r = b - A x, p0 = r, rho is one sweep too (MODE 5).
// The two-sweep form — the fallback when a block cannot be walked with 16-byte lanes (odd widths, odd column
// origins) — needs no prediction:
//     x += alpha'_j p ; p = r + beta_j p       (cg_update_xp: the x update of the PREVIOUS step is
//                                               deferred into this pass: 3 reads + 2 writes)
//     w = A p ; pTw_j = p_j . w_j
//     r -= alpha_j w ; rho_j = r_j . r_j       (cg_update_r: 2 reads + 1 write)
// The b scalars per iteration stay on the host exactly as in the reference (two tiny
// device->host reads per iteration, which is also where the cross-rank all-reduce of
// ops_lin_sol.c:317,365 happens); retired columns get alpha = 0 / keep-flag so their
// x, r, p are bit-for-bit untouched, as if they had been skipped.
//
// A shift published by the caller (GCGE_SetLinearSolverShift: our GCG does it for -gcge_compW_cg_shift, the
// reference leaves sigma to a user-defined solver, ops_eig_sol_gcg.c:584-618) turns the operator into
// A + sigma B (second SpMM + axpy per application; B == NULL: + sigma I).
//
// Installed as ops->MultiLinearSolver by gcge_hip_bpcg_setup(); GCG calls it through
// the reference's user_defined_multi_linear_solver = 1 hook (ops_eig_sol_gcg.c:584-618).
#include <hip/hip_runtime.h>
#include <assert.h>
#include <math.h>
#include <cmath>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "gcge_hip.h"
#include "gcge_hip_internal.h"

extern "C" double* gcge_hip_partial_ws(size_t len);
extern "C" void gcge_hip_reduce_partials(const double* d_partial, int nblocks, int len, double* d_out, void* stream);
extern "C" void gcge_hip_spmm_dot_mv(void* mat, void** x, void** y, int* start, int* end, double* host_dots, struct OPS_* ops);
extern "C" void gcge_hip_spmm_dot2_mv(void* mat, void** x, void** y, int* start, int* end, double* host_dots, double* host_yy,
                                      struct OPS_* ops);
extern "C" void gcge_hip_local_inner_prod(char nsd, void** x, void** y, int* start, int* end, double* ip, int ldIP, struct OPS_* ops);
extern "C" int gcge_hip_cg_fusable(void* mat, void** p, int ncols);
extern "C" int gcge_hip_cg_recompute_pays(void* mat);
extern "C" int gcge_hip_cg_pass2i_dev(void* mat, void** p, void** pprev, void** pnew, int c0, int m, const double* d_alpha,
                                      const double* d_beta, const int* d_flag, const double* d_betaprev, double* d_rho);
extern "C" int gcge_hip_cg_start_scaled_mv(void* mat, void** x, int xc0, const double* host_scale, void** r, void** p0, int rc0,
                                           int m, double* host_rho);
extern "C" int gcge_hip_cg_start_mv(void* mat, void** x, int xc0, void** b, int bc0, void** r, void** p0, int rc0, int m,
                                    double* host_rho);
extern "C" int gcge_hip_cg_pass1_mv(void* mat, void** p, int c0, int m, double* host_pw, double* host_ww);
extern "C" int gcge_hip_cg_pass1_dev(void* mat, void** p, int c0, int m, double* d_out);
extern "C" int gcge_hip_spmm_dot2_dev(void* mat, void** x, void** y, int cx, int cy, int m, double* d_out);
extern "C" int gcge_hip_spmm_dot2_dev_ok(void* mat, void** x, void** y, int cx, int cy, int m);
extern "C" int gcge_hip_cg_pass2_dev(void* mat, void** p, void** r, void** pnew, int c0, int m, const double* d_alpha,
                                     const double* d_beta, const int* d_flag, double* d_rho);
extern "C" int gcge_hip_cg_pass2_mv(void* mat, void** p, void** r, void** pnew, int c0, int m, const double* d_alpha,
                                    const double* d_beta, const int* d_flag, double* host_rho);

namespace gcge {

// x_j += aprev_j p_j (flag bit 2: deferred x update of the previous step), then
// p_j = r_j (flag bits 0-1 == 2, first step) or p_j = r_j + beta_j p_j (== 1); 0: p untouched
__global__ __launch_bounds__(256) void cg_update_xp(long nrows, const double* __restrict__ r, long ldr,
    double* __restrict__ p, long ldp, double* __restrict__ x, long ldx, int m, const double* __restrict__ beta,
    const double* __restrict__ aprev, const int* __restrict__ flag) {
  const long total = nrows * (long)m;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; idx < total; idx += stride) {
    const long row = idx / m;
    const int j = (int)(idx - row * m);
    const int f = flag[j];
    if (f == 0) continue;
    double* pp = p + row * ldp + j;
    double pv = 0.0;
    if ((f & 4) || (f & 3) == 1) pv = *pp;
    if (f & 4) { double* px = x + row * ldx + j; *px = fma(aprev[j], pv, *px); }
    if ((f & 3) == 2) *pp = r[row * ldr + j];
    else if ((f & 3) == 1) *pp = fma(beta[j], pv, r[row * ldr + j]);
  }
}

// r_j -= alpha_j w_j (flagged columns) ; partial[b*m + j] = sum_rows r_j^2
__global__ __launch_bounds__(256) void cg_update_r(long nrows, const double* __restrict__ w, long ldw,
    double* __restrict__ r, long ldr, int m, const double* __restrict__ alpha, const int* __restrict__ flag,
    double* __restrict__ partial, long rows_per_block) {
  __shared__ double red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(nrows, r0 + rows_per_block);
  for (int c0 = 0; c0 < m; c0 += 64) {
    const int j = c0 + tx;
    double s = 0.0;
    if (j < m && flag[j]) {
      const double a = alpha[j];
      for (long row = r0 + ty; row < r1; row += 4) {
        const double rv = fma(-a, w[row * ldw + j], r[row * ldr + j]);
        r[row * ldr + j] = rv;
        s = fma(rv, rv, s);
      }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && j < m)
      partial[(long)blockIdx.x * m + j] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    __syncthreads();
  }
}

typedef double v2d __attribute__((ext_vector_type(2)));

// ---- one-pass step -------------------------------------------------------------------------------------------
// r -= alpha w ; x += alpha p ; p = r_new + beta p ; partial[b*m + j] = sum over the block's rows of r_new^2
// in ONE sweep (4 reads + 3 writes instead of the 5 + 3 streams of the two kernels above).  beta = rho_new / rho
// is needed before r_new exists, so the caller predicts rho_new = alpha^2 (w.w) - rho from the column sums the
// SpMM kernel delivers for free (exact in exact arithmetic: r_new.r_new = rho - 2 alpha r.w + alpha^2 w.w and
// r.w = p.w = rho / alpha for CG directions); the TRUE rho_new comes back from this sweep and is what the next
// alpha and the stopping test use.  Retired columns: alpha = 0, (cr, cb) = (0, 1): r, x, p stay bit-identical.
template <int UNR>
__global__ __launch_bounds__(256) void cg_update_all(long nrows, const double* __restrict__ w, long ldw,
    double* __restrict__ r, long ldr, double* __restrict__ p, long ldp, double* __restrict__ x, long ldx, int m,
    const double* __restrict__ alpha, const double* __restrict__ beta, const int* __restrict__ flag,
    double* __restrict__ partial, int tpr) {
  __shared__ double red[256][2];
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, rpb = 256 / tpr;
  const int j = 2 * tx;
  double s0 = 0.0, s1 = 0.0;
  const bool mine = j < m;
  const int f0 = mine ? flag[j] : 0, f1 = mine ? flag[j + 1] : 0;
  if (mine && (f0 | f1)) {
    const double a0 = f0 ? alpha[j] : 0.0, a1 = f1 ? alpha[j + 1] : 0.0;
    const double cr0 = f0 ? 1.0 : 0.0, cr1 = f1 ? 1.0 : 0.0;
    const double cb0 = f0 ? beta[j] : 1.0, cb1 = f1 ? beta[j + 1] : 1.0;
    const long step = rpb, group = (long)rpb * UNR;
    const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + group - 1) / group * group;
    const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
    auto one = [&](long rr, v2d wv, v2d rv, v2d pv, v2d xv) {
      v2d rn = {fma(-a0, wv.x, rv.x), fma(-a1, wv.y, rv.y)};
      v2d xn = {fma(a0, pv.x, xv.x), fma(a1, pv.y, xv.y)};
      v2d pn = {fma(cb0, pv.x, cr0 * rn.x), fma(cb1, pv.y, cr1 * rn.y)};
      __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(r + rr * ldr + j));
      __builtin_nontemporal_store(xn, reinterpret_cast<v2d*>(x + rr * ldx + j));
      __builtin_nontemporal_store(pn, reinterpret_cast<v2d*>(p + rr * ldp + j));
      s0 = fma(cr0 * rn.x, rn.x, s0); s1 = fma(cr1 * rn.y, rn.y, s1);
    };
    long row = (long)blockIdx.x * slab + ty;
    for (; row + (UNR - 1) * step < rend; row += step * UNR) {
      v2d wv[UNR], rv[UNR], pv[UNR], xv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = row + u * step;
        wv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(w + rr * ldw + j));
        rv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(r + rr * ldr + j));
        pv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + rr * ldp + j));
        xv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(x + rr * ldx + j));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; ++u) one(row + u * step, wv[u], rv[u], pv[u], xv[u]);
    }
    for (; row < rend; row += step)
      one(row, *reinterpret_cast<const v2d*>(w + row * ldw + j), *reinterpret_cast<const v2d*>(r + row * ldr + j),
          *reinterpret_cast<const v2d*>(p + row * ldp + j), *reinterpret_cast<const v2d*>(x + row * ldx + j));
  }
  red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1;
  __syncthreads();
  if (ty == 0 && mine) {
    for (int q = 1; q < rpb; ++q) { s0 += red[q * tpr + tx][0]; s1 += red[q * tpr + tx][1]; }
    partial[(long)blockIdx.x * m + j] = s0;
    partial[(long)blockIdx.x * m + j + 1] = s1;
  }
}

// ---- p-ring variant of the one-pass step ------------------------------------------------------------------------
// x is only needed when the solve ends, but x += alpha p costs a read and a write of a whole block in every
// iteration (2 of the 7 streams).  With memory to spare the directions are kept instead: iteration k reads p_k from
// ring slot k and writes p_{k+1} to slot k+1 (5 streams: w, r, p_k in; r, p_{k+1} out), and every J iterations one
// sweep adds the J pending terms to x (J + 2 streams).  J = 15: 6.1 streams per iteration instead of 7.
template <int UNR>
__global__ __launch_bounds__(256) void cg_update_rp(long nrows, const double* __restrict__ w, long ldw,
    double* __restrict__ r, long ldr, const double* __restrict__ pold, double* __restrict__ pnew, long ldp, int m,
    const double* __restrict__ alpha, const double* __restrict__ beta, const int* __restrict__ flag,
    double* __restrict__ partial, int tpr) {
  __shared__ double red[256][2];
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, rpb = 256 / tpr;
  const int j = 2 * tx;
  double s0 = 0.0, s1 = 0.0;
  const bool mine = j < m;
  const int f0 = mine ? flag[j] : 0, f1 = mine ? flag[j + 1] : 0;
  if (mine) {   // retired columns are still COPIED to the next slot (alpha = 0, cr = 0, cb = 1): the ring must stay complete
    const double a0 = f0 ? alpha[j] : 0.0, a1 = f1 ? alpha[j + 1] : 0.0;
    const double cr0 = f0 ? 1.0 : 0.0, cr1 = f1 ? 1.0 : 0.0;
    const double cb0 = f0 ? beta[j] : 1.0, cb1 = f1 ? beta[j + 1] : 1.0;
    const long step = rpb, group = (long)rpb * UNR;
    const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + group - 1) / group * group;
    const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
    auto one = [&](long rr, v2d wv, v2d rv, v2d pv) {
      v2d rn = {fma(-a0, wv.x, rv.x), fma(-a1, wv.y, rv.y)};
      v2d pn = {fma(cb0, pv.x, cr0 * rn.x), fma(cb1, pv.y, cr1 * rn.y)};
      __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(r + rr * ldr + j));
      __builtin_nontemporal_store(pn, reinterpret_cast<v2d*>(pnew + rr * ldp + j));
      s0 = fma(cr0 * rn.x, rn.x, s0); s1 = fma(cr1 * rn.y, rn.y, s1);
    };
    long row = (long)blockIdx.x * slab + ty;
    for (; row + (UNR - 1) * step < rend; row += step * UNR) {
      v2d wv[UNR], rv[UNR], pv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = row + u * step;
        wv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(w + rr * ldw + j));
        rv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(r + rr * ldr + j));
        pv[u] = *reinterpret_cast<const v2d*>(pold + rr * ldp + j);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; ++u) one(row + u * step, wv[u], rv[u], pv[u]);
    }
    for (; row < rend; row += step)
      one(row, *reinterpret_cast<const v2d*>(w + row * ldw + j), *reinterpret_cast<const v2d*>(r + row * ldr + j),
          *reinterpret_cast<const v2d*>(pold + row * ldp + j));
  }
  red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1;
  __syncthreads();
  if (ty == 0 && mine) {
    for (int q = 1; q < rpb; ++q) { s0 += red[q * tpr + tx][0]; s1 += red[q * tpr + tx][1]; }
    partial[(long)blockIdx.x * m + j] = s0;
    partial[(long)blockIdx.x * m + j + 1] = s1;
  }
}

// The same step without a stored residual (ring of >= 3 slots): r_k = p_k - beta_{k-1} p_{k-1} is rebuilt from the two directions
// the ring holds, so the sweep reads w, p_k, p_{k-1} and writes p_{k+1} — 4 streams instead of 5.  bprev = 0 in the first
// iteration and after a restart (p_k = r_k); retired columns are copied to the next slot.  The rebuilt residual carries an error
// of eps |p| instead of eps |r|: only where the stopping rule is loose (HIP_BlockPCG_run: rate >= 1e-4, <= 100 iterations).
template <int UNR>
__global__ __launch_bounds__(256) void cg_update_p_implicit(long nrows, const double* __restrict__ w, long ldw,
    const double* __restrict__ pprev, const double* __restrict__ pold, double* __restrict__ pnew, long ldp, int m,
    const double* __restrict__ alpha, const double* __restrict__ beta, const double* __restrict__ bprev, const int* __restrict__ flag,
    double* __restrict__ partial, int tpr) {
  __shared__ double red[256][2];
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, rpb = 256 / tpr;
  const int j = 2 * tx;
  double s0 = 0.0, s1 = 0.0;
  const bool mine = j < m;
  const int f0 = mine ? flag[j] : 0, f1 = mine ? flag[j + 1] : 0;
  if (mine) {
    const double a0 = f0 ? alpha[j] : 0.0, a1 = f1 ? alpha[j + 1] : 0.0;
    const double cr0 = f0 ? 1.0 : 0.0, cr1 = f1 ? 1.0 : 0.0;
    const double cb0 = f0 ? beta[j] : 1.0, cb1 = f1 ? beta[j + 1] : 1.0;
    const double bp0 = f0 ? bprev[j] : 0.0, bp1 = f1 ? bprev[j + 1] : 0.0;
    const long step = rpb, group = (long)rpb * UNR;
    const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + group - 1) / group * group;
    const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
    auto one = [&](long rr, v2d wv, v2d qv, v2d pv) {
      v2d rn = {fma(-a0, wv.x, fma(-bp0, qv.x, pv.x)), fma(-a1, wv.y, fma(-bp1, qv.y, pv.y))};
      v2d pn = {fma(cb0, pv.x, cr0 * rn.x), fma(cb1, pv.y, cr1 * rn.y)};
      __builtin_nontemporal_store(pn, reinterpret_cast<v2d*>(pnew + rr * ldp + j));
      s0 = fma(cr0 * rn.x, rn.x, s0); s1 = fma(cr1 * rn.y, rn.y, s1);
    };
    long row = (long)blockIdx.x * slab + ty;
    for (; row + (UNR - 1) * step < rend; row += step * UNR) {
      v2d wv[UNR], qv[UNR], pv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = row + u * step;
        wv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(w + rr * ldw + j));
        qv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(pprev + rr * ldp + j));
        pv[u] = *reinterpret_cast<const v2d*>(pold + rr * ldp + j);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; ++u) one(row + u * step, wv[u], qv[u], pv[u]);
    }
    for (; row < rend; row += step)
      one(row, *reinterpret_cast<const v2d*>(w + row * ldw + j), *reinterpret_cast<const v2d*>(pprev + row * ldp + j),
          *reinterpret_cast<const v2d*>(pold + row * ldp + j));
  }
  red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1;
  __syncthreads();
  if (ty == 0 && mine) {
    for (int q = 1; q < rpb; ++q) { s0 += red[q * tpr + tx][0]; s1 += red[q * tpr + tx][1]; }
    partial[(long)blockIdx.x * m + j] = s0;
    partial[(long)blockIdx.x * m + j + 1] = s1;
  }
}

// Start of a solve whose right-hand sides are b = x diag(scale) (GCGE_SetLinearSolverRhsScale) on a matrix whose product is stored:
// after w = A x, ONE sweep forms r = x diag(scale) - w, writes it as r and as p0 and sums r.r per column — instead of copying x to b,
// scaling b, r = b - w, the column sums and the copy p0 = r (10 block streams; here 4).
template <int UNR>
__global__ __launch_bounds__(256) void cg_start_scaled_stored(long nrows, const double* __restrict__ x, long ldx, const double* __restrict__ w,
    long ldw, double* __restrict__ r, long ldr, double* __restrict__ p0, long ldp, int m, const double* __restrict__ scale,
    double* __restrict__ partial, int tpr) {
  __shared__ double red[256][2];
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, rpb = 256 / tpr;
  const int j = 2 * tx;
  double s0 = 0.0, s1 = 0.0;
  const bool mine = j < m;
  if (mine) {
    const double c0 = scale[j], c1 = scale[j + 1];
    const long step = rpb, group = (long)rpb * UNR;
    const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + group - 1) / group * group;
    const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
    auto one = [&](long rr, v2d xv, v2d wv) {
#pragma clang fp contract(off)
      v2d rn = {c0 * xv.x - wv.x, c1 * xv.y - wv.y};                   // rounded like the column scaling followed by the subtraction
      __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(r + rr * ldr + j));
      __builtin_nontemporal_store(rn, reinterpret_cast<v2d*>(p0 + rr * ldp + j));
      s0 = fma(rn.x, rn.x, s0); s1 = fma(rn.y, rn.y, s1);
    };
    long row = (long)blockIdx.x * slab + ty;
    for (; row + (UNR - 1) * step < rend; row += step * UNR) {
      v2d xv[UNR], wv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long rr = row + u * step;
        xv[u] = *reinterpret_cast<const v2d*>(x + rr * ldx + j);
        wv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(w + rr * ldw + j));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; ++u) one(row + u * step, xv[u], wv[u]);
    }
    for (; row < rend; row += step)
      one(row, *reinterpret_cast<const v2d*>(x + row * ldx + j), *reinterpret_cast<const v2d*>(w + row * ldw + j));
  }
  red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1;
  __syncthreads();
  if (ty == 0 && mine) {
    for (int q = 1; q < rpb; ++q) { s0 += red[q * tpr + tx][0]; s1 += red[q * tpr + tx][1]; }
    partial[(long)blockIdx.x * m + j] = s0;
    partial[(long)blockIdx.x * m + j + 1] = s1;
  }
}

struct RingPtrs { const double* p[16]; };
// x[:, j] += sum_{q < cnt} coef[q * m + j] * ring[q][:, j]      (cnt <= 16)
__global__ __launch_bounds__(256) void cg_accum_x(long nrows, RingPtrs ring, int cnt, long ldp, double* __restrict__ x,
    long ldx, int m, const double* __restrict__ coef, int tpr) {
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr, rpb = 256 / tpr;
  const int j = 2 * tx;
  if (j >= m) return;
  const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + rpb - 1) / rpb * rpb;
  const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
  for (long row = (long)blockIdx.x * slab + ty; row < rend; row += rpb) {
    v2d xv = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(x + row * ldx + j));
#pragma unroll 4
    for (int q = 0; q < cnt; ++q) {
      const v2d pv = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(ring.p[q] + row * ldp + j));
      xv.x = fma(coef[q * m + j], pv.x, xv.x); xv.y = fma(coef[q * m + j + 1], pv.y, xv.y);
    }
    __builtin_nontemporal_store(xv, reinterpret_cast<v2d*>(x + row * ldx + j));
  }
}

// ---- the scalars of an iteration on the device (recompute form) ---------------------------------------------------
// alpha_j = rho_j / (p_j . A p_j); beta_j = rho_pred_j / rho_j with rho_pred = alpha^2 |A p|^2 - rho (clamped at 0: the
// column restarts from r); retired columns get flag 0 / alpha 0 so the passes leave them bit-for-bit untouched.
// ahist: this iteration's row of the pending-x-update coefficients (cg_accum_x).
__global__ __launch_bounds__(256) void cg_scalars_a(int m, const double* __restrict__ rho2, const double* __restrict__ pw_ww,
    const int* __restrict__ active, double* __restrict__ alpha, double* __restrict__ beta, int* __restrict__ flag,
    double* __restrict__ ahist) {
  for (int j = threadIdx.x; j < m; j += 256) {
    double al = 0.0, be = 0.0; int fl = 0;
    if (active[j]) {
      al = rho2[j] / pw_ww[j];
      double pred = al * al * pw_ww[m + j] - rho2[j];
      if (!(pred > 0.0) || !isfinite(pred)) pred = 0.0;
      be = pred / rho2[j]; fl = 1;
    }
    alpha[j] = al; beta[j] = be; flag[j] = fl; ahist[j] = al;
  }
}
// rho_j <- measured r_j . r_j of the sweep, stopping test of src/ops_lin_sol.c:372-380 per column; the number of columns
// still active goes to nact_out (host-mapped memory: the host reads it one iteration later)
__global__ __launch_bounds__(256) void cg_scalars_b(int m, const double* __restrict__ newrho, double rate, double tol,
    const double* __restrict__ norm_b, const double* __restrict__ init_res, double* __restrict__ rho2, int* __restrict__ active,
    double* __restrict__ last_res, int* __restrict__ nact_out) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int mine = 0;
  for (int j = threadIdx.x; j < m; j += 256) {
    if (!active[j]) continue;
    const double rv = newrho[j];
    rho2[j] = rv;
    const double lr = sqrt(rv);
    last_res[j] = lr;
    const int a = (lr > rate * init_res[j]) && (lr > tol * norm_b[j]);
    active[j] = a; mine += a;
  }
  if (mine) atomicAdd(&cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) { *nact_out = cnt; __threadfence_system(); }
}
}  // namespace gcge

using namespace gcge;

static bool cg_vec_ok(int m, std::initializer_list<const double*> ptrs, std::initializer_list<long> lds) {
  if (m & 1) return false;
  for (const double* q : ptrs) if ((uintptr_t)q & 15) return false;
  for (long l : lds) if (l & 1) return false;
  return true;
}
static int cg_tpr(int m) { int t = 1; while (t < m / 2) t *= 2; return t; }   // threads per row (m <= 512)

static void launch_update_xp(long n, const double* r, long ldr, double* p, long ldp, double* x, long ldx, int mw,
                             const double* beta, const double* aprev, const int* flag, hipStream_t st) {
  long total = n * mw, g = (total + 255) / 256; if (g > 8192) g = 8192;
  hipLaunchKernelGGL(cg_update_xp, dim3((unsigned)g), dim3(256), 0, st, n, r, ldr, p, ldp, x, ldx, mw, beta, aprev, flag);
}

struct HipBpcg {
  int max_iter; double rate, tol; char tol_type[8];
  void** mv_ws[4];       // r, p, w (+ B p scratch when a shift with B != NULL is active), created lazily
  void** ring[17]; int ring_len;   // p-ring of the one-pass scheme (ring[0] aliases mv_ws[1])
  int ws_cols, ws_rows;
  int niter; double residual;
  long spmm_calls, spmm_cols;   // statistics for bench.py
  double* d_coef; int* d_flag; double* h_pin; int cap;
  long recompute_iters;         // iterations run in the two-pass form with the product recomputed (pattern matrices)
  long col_iters, active_col_iters;   // columns streamed per iteration, summed / of which still active (one-pass scheme)
  long total_iters; double total_seconds;   // CG iterations and host wall time over all calls (bench.py: ms per CG iteration)
  double* d_sc; int* d_sci; int* h_nact; int sc_cap; hipEvent_t ev_it[2];   // device-side scalars of the recompute form
  long dev_scalar_iters;
  long implicit_r_iters;    // iterations of the device-scalar loop that rebuilt r from two directions (no stored residual)
  long stored_dev_iters;    // iterations of the device-scalar loop with the product stored (matrices without a pattern form)
  long fused_starts;        // solves started by product + one sweep (cg_start_scaled_stored)
  long surplus_iters;       // iterations enqueued after the last column had retired (no-ops on the data, but they stream)
};
static HipBpcg g_bpcg = {30, 1e-2, 1e-14, "abs", {nullptr, nullptr, nullptr, nullptr}, {nullptr}, 0, 0, 0, 0, -1.0, 0, 0, nullptr, nullptr, nullptr, 0};

static int g_residual_form = 0;   // 0 automatic, 1 never stored (rebuilt from two directions), 2 always stored
extern "C" void gcge_hip_bpcg_residual_form(int form) { g_residual_form = form; }

static void reduce_over_ranks(double* v, int n) {
  GCGE_COMM* c = GCGE_GetComm();
  if (c != nullptr && n > 0) c->allreduce_sum(v, n, c->ctx);
}

// r, p, w (and the shift's scratch block on demand) for this problem shape.  A solver that alternates between systems of different
// size — the levels of a multigrid cycle (BlockAMG smooths every level with this CG) — must not rebuild its blocks and ring on
// every call: the set of the shape that is left is PARKED (up to 8 row counts) and taken out again when that shape comes back.
// A change of the column count within a shape drops that shape's ring, as before.
extern "C" unsigned gcge_hip_mv_row_order_id(void** mv);
struct BpcgParked { void** mv_ws[4]; void** ring[17]; int ring_len, ws_cols, ws_rows; unsigned order; };
static unsigned g_ws_order = 0;      // row order (mat_upload.hip "row orders") of the CURRENT set's blocks: they were created like the x block of that call
static BpcgParked g_parked[8]; static int g_nparked = 0;
static void bpcg_destroy_set(BpcgParked* q, struct OPS_* ops) {
  for (int i = 1; i < q->ring_len; ++i) if (q->ring[i]) ops->MultiVecDestroy(&q->ring[i], q->ws_cols, ops);
  for (int i = 0; i < 4; ++i) if (q->mv_ws[i]) ops->MultiVecDestroy(&q->mv_ws[i], q->ws_cols, ops);
  memset(q, 0, sizeof *q);
}
static void bpcg_shape(HipBpcg* s, int n, int nrhs, void** mv_x, struct OPS_* ops) {
  const unsigned order = gcge_hip_mv_row_order_id(mv_x);
  if (s->ws_cols >= nrhs && s->ws_rows == n && g_ws_order == order) return;
  if (s->ws_rows == n && g_ws_order != order && s->mv_ws[0] != nullptr) {   // same size, another row order: the blocks belong to the other matrix
    BpcgParked q; memcpy(q.mv_ws, s->mv_ws, sizeof q.mv_ws); memcpy(q.ring, s->ring, sizeof q.ring); q.ring_len = s->ring_len; q.ws_cols = s->ws_cols;
    bpcg_destroy_set(&q, ops);
    memset(s->mv_ws, 0, sizeof s->mv_ws); memset(s->ring, 0, sizeof s->ring); s->ring_len = 0; s->ws_cols = 0; s->ws_rows = 0;
  }
  if (s->ws_rows != n) {
    if (s->ws_rows > 0 && s->mv_ws[0] != nullptr) {          // park the current set
      if (g_nparked == 8) { bpcg_destroy_set(&g_parked[0], ops); memmove(&g_parked[0], &g_parked[1], 7 * sizeof(BpcgParked)); --g_nparked; }
      BpcgParked* q = &g_parked[g_nparked++];
      memcpy(q->mv_ws, s->mv_ws, sizeof q->mv_ws); memcpy(q->ring, s->ring, sizeof q->ring);
      q->ring_len = s->ring_len; q->ws_cols = s->ws_cols; q->ws_rows = s->ws_rows; q->order = g_ws_order;
      memset(s->mv_ws, 0, sizeof s->mv_ws); memset(s->ring, 0, sizeof s->ring); s->ring_len = 0; s->ws_cols = 0; s->ws_rows = 0;
    }
    for (int i = 0; i < g_nparked; ++i) if (g_parked[i].ws_rows == n && g_parked[i].order != order) {   // parked for another row order: gone
      bpcg_destroy_set(&g_parked[i], ops);
      memmove(&g_parked[i], &g_parked[i + 1], (size_t)(g_nparked - 1 - i) * sizeof(BpcgParked)); --g_nparked; --i;
    }
    for (int i = 0; i < g_nparked; ++i) if (g_parked[i].ws_rows == n) {     // a parked set of this row count comes back
      memcpy(s->mv_ws, g_parked[i].mv_ws, sizeof s->mv_ws); memcpy(s->ring, g_parked[i].ring, sizeof s->ring);
      s->ring_len = g_parked[i].ring_len; s->ws_cols = g_parked[i].ws_cols; s->ws_rows = n; g_ws_order = order;
      memmove(&g_parked[i], &g_parked[i + 1], (size_t)(g_nparked - 1 - i) * sizeof(BpcgParked)); --g_nparked;
      break;
    }
    if (s->ws_cols >= nrhs && s->ws_rows == n) return;
  }
  for (int i = 1; i < s->ring_len; ++i) if (s->ring[i]) ops->MultiVecDestroy(&s->ring[i], s->ws_cols, ops);
  s->ring_len = 0;
  for (int i = 0; i < 4; ++i) {
    if (s->mv_ws[i]) ops->MultiVecDestroy(&s->mv_ws[i], s->ws_cols, ops);
    if (i < 3) ops->MultiVecCreateByMultiVec(&s->mv_ws[i], nrhs, mv_x, ops);
  }
  s->ws_cols = nrhs; s->ws_rows = n; g_ws_order = order;
}
static void bpcg_ring(HipBpcg* s, void* mat, void** mv_x, double sigma, struct OPS_* ops) {
  long ldp = 0;
  (void)gcge_hip_mv_device_ptr(s->mv_ws[1], &ldp);
  // p-ring (see cg_update_rp): as many slots as memory allows, at most 16; fewer than 4 pending terms do not pay
  const int ring_max = getenv("GCGE_CG_RING") ? atoi(getenv("GCGE_CG_RING")) : 16;   // (read when a ring is created)
  // (a ring that was created for a solver with fewer iterations — the smoothing runs of a multigrid cycle — grows when a later
  //  set-up asks for more)
  const int have = s->ring_len > 0 ? s->ring_len - 1 : 0;
  if (s->max_iter >= 3 && have < 15 && have < s->max_iter) {   // (every rank gets here in the same call: the vote below is collective)
    size_t fr = 0, tot = 0;
    GCGE_HIP_CHECK(hipMemGetInfo(&fr, &tot));
    fr += gcge_hip_pool_cached_bytes();   // blocks parked in the back-end's pool are available to MultiVecCreate*
    const size_t slot = (size_t)gcge_hip_mv_nrows(s->mv_ws[1]) * (size_t)ldp * sizeof(double) + ((size_t)64 << 20);
    const size_t keep = (size_t)12 << 30;   // leave room for staging, partial sums and the caller
    long want = fr > keep ? (long)((fr - keep) / slot) : 0;
    if (want > ring_max - 1) want = ring_max - 1;
    // smallest ring worth having: with the product recomputed (1 + 4 streams per iteration) even two extra slots pay —
    // x then costs (2 + 2) / 2 = 2 streams per iteration, 7 in all against 8.1 of the stored-w form; that is the case of
    // BASELINE config 4's shape, where 244 of 288 GB are taken by the solver's own blocks
    // (with the product stored — generic matrices, shifts — a ring only pays from 4 extra slots on: 2 + 5 + (J + 2) / J
    // streams against the 9 of the ring-less sweep)
    const int min_ring = (sigma == 0.0 && gcge_hip_cg_recompute_pays(mat)) ? 2 : 4;
    if (want < min_ring) want = 0;
    want += have;                                  // (slots this shape already holds are not "free memory" any more)
    if (want > ring_max - 1) want = ring_max - 1;
    if (want > s->max_iter) want = s->max_iter;
    // Row-partitioned runs: the ring length decides the column window [alo, ahi) and with it the LENGTH of the two
    // all-reduces of an iteration, so every rank must use the same one — the minimum over the ranks (free memory
    // differs between slabs cut by non-zeros and between ranks sharing a device).  GCGE_COMM only sums: rank r's
    // "can hold at least q+1 extra slots" indicators are summed, the agreed length is the count of unanimous entries.
    if (GCGE_COMM* c = GCGE_GetComm()) {
      double vote[16];
      for (int q = 0; q < 16; ++q) vote[q] = q < want ? 1.0 : 0.0;
      c->allreduce_sum(vote, 16, c->ctx);
      long agreed = 0;
      while (agreed < 16 && vote[agreed] > (double)c->size - 0.5) ++agreed;
      want = agreed;
    }
    if (s->ring_len == 0) { s->ring[0] = s->mv_ws[1]; s->ring_len = 1; }
    for (long i = have; i < want; ++i) { ops->MultiVecCreateByMultiVec(&s->ring[s->ring_len], s->ws_cols, mv_x, ops); ++s->ring_len; }
  }
}

static void HIP_BlockPCG_run(void* mat, void** mv_b, void** mv_x, int* start_bx, int* end_bx, struct OPS_* ops);
static void HIP_BlockPCG(void* mat, void** mv_b, void** mv_x, int* start_bx, int* end_bx, struct OPS_* ops) {
  HipBpcg* s = (HipBpcg*)ops->multi_linear_solver_workspace;
  const double t0 = ops->GetWtime ? ops->GetWtime() : 0.0;
  HIP_BlockPCG_run(mat, mv_b, mv_x, start_bx, end_bx, ops);
  s->total_iters += s->niter;
  if (ops->GetWtime) s->total_seconds += ops->GetWtime() - t0;
  static const bool trace = getenv("GCGE_CG_TRACE") != nullptr;   // one line per call on stderr (tuning aid)
  if (trace && ops->GetWtime)
    fprintf(stderr, "HIP_BlockPCG: %d columns (b at %d, x at %d), %d iterations, ring of %d, %.1f ms\n", end_bx[0] - start_bx[0], start_bx[0],
            start_bx[1], s->niter, s->ring_len, 1e3 * (ops->GetWtime() - t0));
}
static void HIP_BlockPCG_run(void* mat, void** mv_b, void** mv_x, int* start_bx, int* end_bx, struct OPS_* ops) {
  HipBpcg* s = (HipBpcg*)ops->multi_linear_solver_workspace;
  hipStream_t st = (hipStream_t)gcge_hip_stream();
  const int nrhs = end_bx[0] - start_bx[0];
  assert(nrhs == end_bx[1] - start_bx[1]);
  if (nrhs <= 0) { s->niter = 0; return; }
  const int n = gcge_hip_mv_nrows(mv_x);
  if (gcge_hip_mv_nrows(mv_b) != n) { fprintf(stderr, "HIP_BlockPCG: b and x have different row counts\n"); abort(); }
  bpcg_shape(s, n, nrhs, mv_x, ops);
  // operator: A, or A + sigma B when the caller published a shift (GCGE_SetLinearSolverShift)
  double sigma = 0.0; void* matB = nullptr;
  GCGE_GetLinearSolverShift(&sigma, &matB);
  if (sigma != 0.0 && matB != nullptr && s->mv_ws[3] == nullptr) ops->MultiVecCreateByMultiVec(&s->mv_ws[3], s->ws_cols, mv_x, ops);
  // y[:, ys:ys+k) = (A + sigma B) x[:, xs:xs+k); dots != NULL: dots[j] = x_j . y_j (local part)
  auto apply = [&](void** xin, int xs, void** yout, int ys, int k, double* dots, double* yy = nullptr) {
    int a2[2] = {xs, ys}, b2[2] = {xs + k, ys + k};
    if (sigma == 0.0) {
      if (dots) gcge_hip_spmm_dot2_mv(mat, xin, yout, a2, b2, dots, yy, ops);
      else ops->MatDotMultiVec(mat, xin, yout, a2, b2, ops);
      return;
    }
    ops->MatDotMultiVec(mat, xin, yout, a2, b2, ops);
    if (matB != nullptr) {
      int a3[2] = {xs, 0}, b3[2] = {xs + k, k};
      ops->MatDotMultiVec(matB, xin, s->mv_ws[3], a3, b3, ops);
      int a4[2] = {0, ys}, b4[2] = {k, ys + k};
      ops->MultiVecAxpby(sigma, s->mv_ws[3], 1.0, yout, a4, b4, ops);
    } else {
      ops->MultiVecAxpby(sigma, xin, 1.0, yout, a2, b2, ops);
    }
    if (dots) gcge_hip_local_inner_prod('D', xin, yout, a2, b2, dots, 1, ops);
    if (yy) { int a5[2] = {ys, ys}, b5[2] = {ys + k, ys + k}; gcge_hip_local_inner_prod('D', yout, yout, a5, b5, yy, 1, ops); }
  };
  if (s->cap < nrhs) {
    if (s->d_coef) { hipFree(s->d_coef); hipFree(s->d_flag); hipHostFree(s->h_pin); }
    s->cap = nrhs + 64;
    GCGE_HIP_CHECK(hipMalloc(&s->d_coef, 2 * s->cap * sizeof(double)));
    GCGE_HIP_CHECK(hipMalloc(&s->d_flag, s->cap * sizeof(int)));
    GCGE_HIP_CHECK(hipHostMalloc(&s->h_pin, 3 * s->cap * sizeof(double)));
  }
  long ldb, ldx, ldr, ldp, ldw;
  double* db = gcge_hip_mv_device_ptr(mv_b, &ldb) + start_bx[0];
  double* dx = gcge_hip_mv_device_ptr(mv_x, &ldx) + start_bx[1];
  double* dr = gcge_hip_mv_device_ptr(s->mv_ws[0], &ldr);
  double* dp = gcge_hip_mv_device_ptr(s->mv_ws[1], &ldp);
  double* dw = gcge_hip_mv_device_ptr(s->mv_ws[2], &ldw);
  (void)db; (void)dw;

  std::vector<double> norm_b(nrhs), rho1(nrhs), rho2(nrhs), pTw(nrhs), init_res(nrhs), last_res(nrhs), coef(nrhs);
  std::vector<int> active(nrhs), flag(nrhs);
  int st2[2], en2[2];
  // The caller may have declared b = x diag(scale) without forming it (GCGE_SetLinearSolverRhsScale: our GCG driver's
  // systems A w = (lambda + sigma) x, started from w = x).  The one-sweep start below then takes the scale factors
  // and neither reads nor needs b; every other route forms b first, here, on the device.
  const double* rhs_scale = GCGE_GetLinearSolverRhsScale();
  bool p0_done = false;
  // Will the iteration rebuild its residual from the directions (device-scalar loop of the recompute form, MODE 7 below)?  Then the
  // block r is never read, and the one-sweep starts store p0 alone (r and p0 name the same block: single store, 3 block streams
  // instead of 4 — the two smoothing calls of a V-cycle start this way).  Same conditions as where the loop is chosen further down;
  // the ring is sized first (bpcg_ring is collective: every rank gets here in the same call as before, only earlier).
  bool start_without_r = false;
  if (nrhs <= 512 && cg_vec_ok(nrhs, {dw, dr, dp, dx}, {ldw, ldr, ldp, ldx}) && sigma == 0.0 && gcge_hip_cg_recompute_pays(mat) &&
      getenv("GCGE_CG_START_STORES_R") == nullptr) {
    bpcg_ring(s, mat, mv_x, sigma, ops);
    const int R0p = s->ring_len >= 2 ? s->ring_len : 1;
    const bool rec_p = R0p > 1 && gcge_hip_cg_fusable(mat, s->ring[0], nrhs) && !(nrhs & 1);
    const int Rp = R0p + ((rec_p && R0p < 16 && getenv("GCGE_CG_NO_W_SLOT") == nullptr) ? 1 : 0);   // (idle blocks can only add slots)
    GCGE_COMM* comm_p = GCGE_GetComm();
    start_without_r = rec_p && Rp >= 3 && getenv("GCGE_CG_HOST_SCALARS") == nullptr && s->max_iter <= 4000 &&
                      (comm_p == nullptr || gcge_hip_comm_is_native(comm_p)) &&
                      (g_residual_form == 1 || (g_residual_form == 0 && s->rate >= 1e-4 && s->max_iter <= 100));
  }
  void** start_r = start_without_r ? s->mv_ws[1] : s->mv_ws[0];
  if (rhs_scale != nullptr) {
    if (0 != strcmp(s->tol_type, "rel") && sigma == 0.0 && gcge_hip_cg_recompute_pays(mat) &&   // ("rel" needs |b|: b is formed)
        gcge_hip_cg_start_scaled_mv(mat, mv_x, start_bx[1], rhs_scale, start_r, s->mv_ws[1], 0, nrhs, rho2.data()) == 0) {
      reduce_over_ranks(rho2.data(), nrhs);
      p0_done = true;
    } else if (0 != strcmp(s->tol_type, "rel") && sigma == 0.0 && mat != nullptr && nrhs <= 512 && getenv("GCGE_CG_NO_FUSED_START") == nullptr &&
               cg_vec_ok(nrhs, {dw, dr, dp, dx}, {ldw, ldr, ldp, ldx})) {
      // matrices whose product is stored (no pattern form): w = A x, then one sweep for r = x diag(scale) - w, p0 = r, r.r
      // (cg_start_scaled_stored: 4 block streams behind the product instead of 10)
      apply(mv_x, start_bx[1], s->mv_ws[2], 0, nrhs, nullptr);
      long nb0 = ((long)n + 255) / 256; if (nb0 > 2048) nb0 = 2048;
      const long rpb0 = (((long)n + nb0 - 1) / nb0 + 3) / 4 * 4;
      nb0 = ((long)n + rpb0 - 1) / rpb0;
      hipStream_t st0 = (hipStream_t)gcge_hip_stream();
      double* part = gcge_hip_partial_ws((size_t)nb0 * nrhs + nrhs);
      GCGE_HIP_CHECK(hipStreamSynchronize(st0));          // the pinned staging may still feed an earlier upload
      memcpy(s->h_pin, rhs_scale, nrhs * sizeof(double));
      GCGE_HIP_CHECK(hipMemcpyAsync(s->d_coef, s->h_pin, nrhs * sizeof(double), hipMemcpyHostToDevice, st0));
      hipLaunchKernelGGL(cg_start_scaled_stored<4>, dim3((unsigned)nb0), dim3(256), 0, st0, (long)n, (const double*)dx, ldx, (const double*)dw, ldw,
                         dr, ldr, dp, ldp, nrhs, (const double*)s->d_coef, part, cg_tpr(nrhs));
      gcge_hip_reduce_partials(part, (int)nb0, nrhs, part + (size_t)nb0 * nrhs, st0);
      GCGE_HIP_CHECK(hipMemcpyAsync(s->h_pin, part + (size_t)nb0 * nrhs, nrhs * sizeof(double), hipMemcpyDeviceToHost, st0));
      GCGE_HIP_CHECK(hipStreamSynchronize(st0));
      memcpy(rho2.data(), s->h_pin, nrhs * sizeof(double));
      reduce_over_ranks(rho2.data(), nrhs);
      p0_done = true;
      ++s->fused_starts;
    } else {
      st2[0] = start_bx[1]; en2[0] = end_bx[1]; st2[1] = start_bx[0]; en2[1] = end_bx[0];
      ops->MultiVecAxpby(1.0, mv_x, 0.0, mv_b, st2, en2, ops);
      std::vector<double> sc(rhs_scale, rhs_scale + nrhs);
      ops->MultiVecLinearComb(NULL, mv_b, 0, st2, en2, NULL, 0, sc.data(), 1, ops);
      rhs_scale = nullptr;
    }
  }
  if (0 == strcmp(s->tol_type, "rel")) {
    st2[0] = start_bx[0]; en2[0] = end_bx[0]; st2[1] = start_bx[0]; en2[1] = end_bx[0];
    ops->MultiVecInnerProd('D', mv_b, mv_b, 0, st2, en2, norm_b.data(), 1, ops);
    for (int i = 0; i < nrhs; ++i) norm_b[i] = sqrt(norm_b[i]);
  } else if (0 == strcmp(s->tol_type, "user")) {
    // src/ops_lin_sol.c:186-192: the caller's scales, |lambda_j + sigma| from the GCG driver (GCGE_SetLinearSolverUserScale)
    int nsc = 0;
    const double* usc = GCGE_GetLinearSolverUserScale(&nsc);
    if (usc == nullptr || nsc < nrhs) {
      fprintf(stderr, "HIP_BlockPCG: tol_type \"user\" but the caller published %d scales for %d right-hand sides "
                      "(GCGE_SetLinearSolverUserScale, include/gcge_ops.h)\n", nsc, nrhs);
      abort();
    }
    for (int i = 0; i < nrhs; ++i) norm_b[i] = fabs(usc[i]);
  } else {
    for (int i = 0; i < nrhs; ++i) norm_b[i] = 1.0;   // "abs" (gcge_hip_bpcg_setup refuses anything else)
  }
  // r = b - A x ; rho2 = diag(r^T r) ; p0 = r.  On pattern matrices in one sweep (kernel MODE 5) when the operands
  // allow it; otherwise product, axpby, column dots (and the copy p0 = r further down)
  if (p0_done) {
    // started from the scale factors above
  } else if (sigma == 0.0 && gcge_hip_cg_recompute_pays(mat) &&
      gcge_hip_cg_start_mv(mat, mv_x, start_bx[1], mv_b, start_bx[0], start_r, s->mv_ws[1], 0, nrhs, rho2.data()) == 0) {
    reduce_over_ranks(rho2.data(), nrhs);
    p0_done = true;
  } else {
    apply(mv_x, start_bx[1], s->mv_ws[0], 0, nrhs, nullptr);
    st2[0] = start_bx[0]; en2[0] = end_bx[0]; st2[1] = 0; en2[1] = nrhs;
    ops->MultiVecAxpby(1.0, mv_b, -1.0, s->mv_ws[0], st2, en2, ops);
    st2[0] = 0; en2[0] = nrhs; st2[1] = 0; en2[1] = nrhs;
    ops->MultiVecInnerProd('D', s->mv_ws[0], s->mv_ws[0], 0, st2, en2, rho2.data(), 1, ops);
  }
  s->spmm_calls++; s->spmm_cols += nrhs;
  int nact = 0;
  for (int i = 0; i < nrhs; ++i) {
    init_res[i] = sqrt(rho2[i]); last_res[i] = init_res[i];
    active[i] = init_res[i] > s->tol * norm_b[i];
    nact += active[i];
  }
  long nb = ((long)n + 255) / 256; if (nb > 2048) nb = 2048;
  const long rpb = (((long)n + nb - 1) / nb + 3) / 4 * 4;
  nb = ((long)n + rpb - 1) / rpb;

  // d_coef holds [beta | aprev/alpha] (2*cap doubles), d_flag the per-column flags
  std::vector<double> aprev(nrhs, 0.0);
  std::vector<int> pend(nrhs, 0);           // x update of the last step still to be applied
  auto upload = [&](int lo, int mw, const double* c0v, const double* c1v, const int* fl) {
    GCGE_HIP_CHECK(hipStreamSynchronize(st));
    memcpy(s->h_pin, c0v + lo, mw * sizeof(double));
    memcpy(s->h_pin + s->cap, c1v + lo, mw * sizeof(double));
    memcpy(s->h_pin + 2 * s->cap, fl + lo, mw * sizeof(int));
    GCGE_HIP_CHECK(hipMemcpyAsync(s->d_coef, s->h_pin, mw * sizeof(double), hipMemcpyHostToDevice, st));
    GCGE_HIP_CHECK(hipMemcpyAsync(s->d_coef + s->cap, s->h_pin + s->cap, mw * sizeof(double), hipMemcpyHostToDevice, st));
    GCGE_HIP_CHECK(hipMemcpyAsync(s->d_flag, s->h_pin + 2 * s->cap, mw * sizeof(int), hipMemcpyHostToDevice, st));
  };
  int niter = 0;
  // ---- one-pass scheme (default whenever all four blocks can be walked with 16-byte lanes) ----
  if (nrhs <= 512 && cg_vec_ok(nrhs, {dw, dr, dp, dx}, {ldw, ldr, ldp, ldx})) {
    std::vector<double> wTw(nrhs), bet(nrhs);
    bpcg_ring(s, mat, mv_x, sigma, ops);
    // Pattern matrices (stencils) with a ring: the product is formed twice and never stored — pass 1 reads p for
    // p.w and w.w, pass 2 reads p again and applies the r / p update with w rebuilt in registers (app_hip.hip:
    // gcge_hip_cg_pass1_mv / pass2_mv): 1 + 4 block streams per iteration instead of 2 + 5.  Same recurrences,
    // same operands, so alpha, beta and the iterates agree with the stored-w form to rounding of the sums.
    const int R0 = s->ring_len >= 2 ? s->ring_len : 1;
    const bool recompute = R0 > 1 && sigma == 0.0 && gcge_hip_cg_recompute_pays(mat) && gcge_hip_cg_fusable(mat, s->ring[0], nrhs) && !(nrhs & 1);
    // the slots of this call: the ring, plus the w block — which the recompute form never writes — as one more slot
    // (matters where memory is short: at BASELINE config 4's shape a ring of 3 becomes one of 4, x costs 5/3 instead of
    // 2 block streams per iteration)
    void** slots[17]; int R = R0;                            // ring slots; 1: x is updated in every iteration
    for (int q = 0; q < R0; ++q) slots[q] = R0 > 1 ? s->ring[q] : s->mv_ws[1];
    if (recompute && R < 16 && getenv("GCGE_CG_NO_W_SLOT") == nullptr) slots[R++] = s->mv_ws[2];
    // ... and the blocks the caller declared idle for the duration of the solve (GCGE_SetLinearSolverIdleBlocks), where
    // they have the shape of the ring's own blocks (same rows, same leading dimension, 16-byte aligned, room for the halo)
    if (recompute && getenv("GCGE_CG_NO_IDLE_SLOTS") == nullptr) {
      int nidle = 0; void*** idle = GCGE_GetLinearSolverIdleBlocks(&nidle);
      for (int q = 0; q < nidle && R < 16; ++q) {
        void** hb = idle[q];
        if (hb == nullptr || hb == mv_x || hb == mv_b || gcge_hip_mv_nrows(hb) != n || gcge_hip_mv_ncols(hb) < nrhs) continue;
        bool dup = false;
        for (int t = 0; t < R; ++t) dup |= slots[t] == hb;
        for (int t = 0; t < 4; ++t) dup |= s->mv_ws[t] == hb;
        long ldq = 0;
        if (dup || gcge_hip_mv_device_ptr(hb, &ldq) == nullptr || ldq != ldp || !gcge_hip_cg_fusable(mat, hb, nrhs)) continue;
        slots[R++] = hb;
      }
    }
    const int J = R - 1;                                     // pending directions before x is brought up to date
    std::vector<double> ahist((size_t)(J > 0 ? J : 1) * nrhs, 0.0);
    int npend = 0, first_slot = 0, cur = 0;                  // p_k lives in ring[cur]; pending: slots first_slot .. (npend of them)
    double* d_ahist = nullptr;
    auto flush_x = [&]() {
      if (npend == 0) return;
      RingPtrs rp;
      for (int q = 0; q < 16; ++q) { long ldq; rp.p[q] = gcge_hip_mv_device_ptr(slots[(first_slot + (q < npend ? q : 0)) % R], &ldq); }
      d_ahist = gcge_hip_partial_ws((size_t)J * nrhs);
      GCGE_HIP_CHECK(hipMemcpyAsync(d_ahist, ahist.data(), (size_t)npend * nrhs * sizeof(double), hipMemcpyHostToDevice, st));
      const int tpr = cg_tpr(nrhs);
      long g = ((long)n + (256 / tpr) * 8 - 1) / ((256 / tpr) * 8); if (g > 8192) g = 8192; if (g < 1) g = 1;
      hipLaunchKernelGGL(cg_accum_x, dim3((unsigned)g), dim3(256), 0, st, (long)n, rp, npend, ldp, dx, ldx, nrhs, d_ahist, tpr);
      GCGE_HIP_CHECK(hipStreamSynchronize(st));   // ahist (pageable) and the partial workspace are reused right away
      first_slot = (first_slot + npend) % R; npend = 0;
    };
    if (nact > 0 && !p0_done) {   // p0 = r0  (ring[0] is mv_ws[1])
      st2[0] = 0; en2[0] = nrhs; st2[1] = 0; en2[1] = nrhs;
      ops->MultiVecAxpby(1.0, s->mv_ws[0], 0.0, s->ring_len ? s->ring[0] : s->mv_ws[1], st2, en2, ops);
    }
    // ---- recompute form with the scalars on the device: no host round trip inside an iteration ----------------------
    // pass 1 -> [all-reduce] -> cg_scalars_a -> pass 2 -> [all-reduce] -> cg_scalars_b, all on the back-end's stream; the
    // host only learns, one iteration late, how many columns are still active.  The iteration it enqueued in the
    // meantime is a no-op on the data when that number was 0 (all flags 0: r untouched, p copied, zero coefficients for
    // the pending x update).  Over ranks the sums are reduced on the device by RCCL (gcge_hip_comm_allreduce_device); a
    // callback transport (torch.distributed rehearsals) keeps the host-scalar loop below.  GCGE_CG_HOST_SCALARS=1: off.
    GCGE_COMM* comm_now = GCGE_GetComm();
    const bool host_scalars = getenv("GCGE_CG_HOST_SCALARS") != nullptr;
    // The stored-product form (matrices without a pattern form: the plane sweep, dense blocks, pad-8; no shift) takes the same loop
    // from round 4 on: product + sums left on the device (gcge_hip_spmm_dot2_dev), the direction update reading alpha / beta /
    // flags where cg_scalars_a put them.  Before, every iteration made two host round trips (sums down, coefficients up, new
    // rho down): 0.6 of 6.2 ms per iteration on the SiO2-like matrix.  GCGE_CG_HOST_SCALARS=1: the host loop below.
    // (only where gcge_hip_spmm_dot2_dev will take these operands — a row slab whose halo buffers are narrower than the
    //  block, or a slot without room for the halo rows, keeps the host-scalar loop, which chunks the columns)
    const bool stored_dev = !recompute && R >= 2 && sigma == 0.0 && mat != nullptr && getenv("GCGE_CG_STORED_HOST") == nullptr &&
                            gcge_hip_spmm_dot2_dev_ok(mat, slots[0], s->mv_ws[2], 0, 0, nrhs);
    if ((recompute || stored_dev) && !host_scalars && nact > 0 && s->max_iter <= 4000 && (comm_now == nullptr || gcge_hip_comm_is_native(comm_now))) {
      if (s->sc_cap < nrhs) {
        GCGE_HIP_CHECK(hipStreamSynchronize(st));
        if (s->d_sc) { hipFree(s->d_sc); hipFree(s->d_sci); hipHostFree(s->h_nact); }
        else { GCGE_HIP_CHECK(hipEventCreateWithFlags(&s->ev_it[0], hipEventDisableTiming)); GCGE_HIP_CHECK(hipEventCreateWithFlags(&s->ev_it[1], hipEventDisableTiming)); }
        s->sc_cap = nrhs + 64;
        GCGE_HIP_CHECK(hipMalloc(&s->d_sc, (size_t)(30 + 16) * s->sc_cap * sizeof(double)));
        GCGE_HIP_CHECK(hipMalloc(&s->d_sci, (size_t)2 * s->sc_cap * sizeof(int)));
        GCGE_HIP_CHECK(hipHostMalloc(&s->h_nact, 4096 * sizeof(int)));
      }
      const size_t cap = (size_t)s->sc_cap;
      double *d_rho2 = s->d_sc, *d_init = d_rho2 + cap, *d_normb = d_init + cap, *d_last = d_normb + cap, *d_alpha = d_last + cap,
             *d_beta = d_alpha + cap, *d_sums = d_beta + cap /* 6 cap */, *d_newrho = d_sums + 6 * cap /* 6 cap */,
             *d_ahist2 = d_newrho + 6 * cap /* 16 cap, starts at 18 cap: (30 + 16) cap in total */;
      int *d_active = s->d_sci, *d_flag2 = d_active + cap;
      // The residual is not stored (gcge_hip_bpcg_residual_form(2): it is): r_k = p_k - beta_{k-1} p_{k-1} is rebuilt in the second
      // pass from the previous direction, which the ring still holds (3 slots suffice: p_{k-1}, p_k, p_{k+1}) — the pass
      // then reads p_k, p_{k-1} and writes p_{k+1}: 3 block streams instead of 4, 5.1 instead of 6.1 per iteration.  Same
      // recurrence in exact arithmetic; in floating point the rebuilt r_k carries a rounding error of eps |p_k| instead
      // of eps |r_k| — immaterial for systems solved to a relative 1e-2 (kernel MODE 7, spmm_pattern.hip).
      // Taken where the systems are solved to a moderate reduction (rate >= 1e-4 within <= 100 iterations: the GCG driver's
      // 1e-2 in 30); a caller asking for more keeps the stored residual — gcge_hip_bpcg_residual_form overrides
      // either way, tests/test_hip_parity.py::test_fused_cg_tight_tolerances compares the two forms down to 1e-13.
      const bool implicit_r = R >= 3 && (g_residual_form == 1 || (g_residual_form == 0 && s->rate >= 1e-4 && s->max_iter <= 100));
      if (start_without_r && p0_done && !(recompute && implicit_r)) { fprintf(stderr, "HIP_BlockPCG: the start sweep left no residual but the iteration needs one\n"); abort(); }
      double* d_betaB = d_ahist2 + 16 * cap;   // second beta buffer: beta_k and beta_{k-1} alternate between the two
      if (implicit_r) {
        GCGE_HIP_CHECK(hipMemsetAsync(d_beta, 0, cap * sizeof(double), st));
        GCGE_HIP_CHECK(hipMemsetAsync(d_betaB, 0, cap * sizeof(double), st));
      }
      {   // start values (rho, initial residuals, scales, active flags were computed on the host above)
        std::vector<double> up(3 * (size_t)nrhs);
        memcpy(up.data(), rho2.data(), nrhs * sizeof(double)); memcpy(up.data() + nrhs, init_res.data(), nrhs * sizeof(double));
        memcpy(up.data() + 2 * nrhs, norm_b.data(), nrhs * sizeof(double));
        GCGE_HIP_CHECK(hipMemcpyAsync(d_rho2, up.data(), nrhs * sizeof(double), hipMemcpyHostToDevice, st));
        GCGE_HIP_CHECK(hipMemcpyAsync(d_init, up.data() + nrhs, nrhs * sizeof(double), hipMemcpyHostToDevice, st));
        GCGE_HIP_CHECK(hipMemcpyAsync(d_normb, up.data() + 2 * nrhs, nrhs * sizeof(double), hipMemcpyHostToDevice, st));
        GCGE_HIP_CHECK(hipMemcpyAsync(d_last, up.data() + nrhs, nrhs * sizeof(double), hipMemcpyHostToDevice, st));
        GCGE_HIP_CHECK(hipMemcpyAsync(d_active, active.data(), nrhs * sizeof(int), hipMemcpyHostToDevice, st));
        GCGE_HIP_CHECK(hipStreamSynchronize(st));   // `up` leaves scope
      }
      const bool reduce = comm_now != nullptr;
      auto flush_x_dev = [&]() {
        if (npend == 0) return;
        RingPtrs rp;
        for (int q = 0; q < 16; ++q) { long ldq; rp.p[q] = gcge_hip_mv_device_ptr(slots[(first_slot + (q < npend ? q : 0)) % R], &ldq); }
        const int tpr = cg_tpr(nrhs);
        long g = ((long)n + (256 / tpr) * 8 - 1) / ((256 / tpr) * 8); if (g > 8192) g = 8192; if (g < 1) g = 1;
        hipLaunchKernelGGL(cg_accum_x, dim3((unsigned)g), dim3(256), 0, st, (long)n, rp, npend, ldp, dx, ldx, nrhs, d_ahist2, tpr);
        first_slot = (first_slot + npend) % R; npend = 0;
      };
      int enq = 0;            // iterations enqueued
      int done = -1;          // index of the last iteration whose active count the host has seen
      int stop_at = -1;       // first iteration that found no active column at its start
      while (enq < s->max_iter && enq < 4096) {
        void** pcur = slots[cur];
        if (recompute) {
          if (gcge_hip_cg_pass1_dev(mat, pcur, 0, nrhs, d_sums) != 0) { fprintf(stderr, "HIP_BlockPCG: first CG pass refused operands it had accepted\n"); abort(); }
        } else if (gcge_hip_spmm_dot2_dev(mat, pcur, s->mv_ws[2], 0, 0, nrhs, d_sums) != 0) {
          fprintf(stderr, "HIP_BlockPCG: product with column sums refused operands the one-pass scheme had accepted\n"); abort();
        }
        if (reduce) gcge_hip_comm_allreduce_device(d_sums, 2 * nrhs);
        double* bcur = (implicit_r && (enq & 1)) ? d_betaB : d_beta;
        const double* bprev = (enq & 1) ? d_beta : d_betaB;
        hipLaunchKernelGGL(cg_scalars_a, dim3(1), dim3(256), 0, st, nrhs, d_rho2, d_sums, d_active, d_alpha, bcur, d_flag2,
                           d_ahist2 + (size_t)npend * nrhs);
        int rc2 = 0;
        if (!recompute) {     // w is stored: one sweep over w, the directions (and r where it is stored) — cg_update_p_implicit / cg_update_rp
          long ldq;
          const double* pold = gcge_hip_mv_device_ptr(slots[cur], &ldq);
          double* pnew = gcge_hip_mv_device_ptr(slots[(cur + 1) % R], &ldq);
          double* part = gcge_hip_partial_ws((size_t)nb * nrhs);
          if (implicit_r) {
            const double* pprev = gcge_hip_mv_device_ptr(slots[(cur + R - 1) % R], &ldq);
            hipLaunchKernelGGL(cg_update_p_implicit<4>, dim3((unsigned)nb), dim3(256), 0, st, (long)n, (const double*)dw, ldw, enq == 0 ? pold : pprev,
                               pold, pnew, ldp, nrhs, (const double*)d_alpha, (const double*)bcur, bprev, (const int*)d_flag2, part, cg_tpr(nrhs));
            ++s->implicit_r_iters;
          } else
            hipLaunchKernelGGL(cg_update_rp<4>, dim3((unsigned)nb), dim3(256), 0, st, (long)n, (const double*)dw, ldw, dr, ldr, pold, pnew, ldp, nrhs,
                               (const double*)d_alpha, (const double*)d_beta, (const int*)d_flag2, part, cg_tpr(nrhs));
          gcge_hip_reduce_partials(part, (int)nb, nrhs, d_newrho, st);
          ++s->stored_dev_iters;
        } else if (implicit_r) {
          rc2 = gcge_hip_cg_pass2i_dev(mat, pcur, enq == 0 ? pcur : slots[(cur + R - 1) % R], slots[(cur + 1) % R], 0, nrhs, d_alpha, bcur,
                                       d_flag2, bprev, d_newrho);
          ++s->implicit_r_iters;
        } else rc2 = gcge_hip_cg_pass2_dev(mat, pcur, s->mv_ws[0], slots[(cur + 1) % R], 0, nrhs, d_alpha, d_beta, d_flag2, d_newrho);
        if (rc2 != 0) { fprintf(stderr, "HIP_BlockPCG: second CG pass refused operands it had accepted\n"); abort(); }
        if (reduce) gcge_hip_comm_allreduce_device(d_newrho, nrhs);
        hipLaunchKernelGGL(cg_scalars_b, dim3(1), dim3(256), 0, st, nrhs, d_newrho, s->rate, s->tol, d_normb, d_init, d_rho2, d_active,
                           d_last, s->h_nact + enq);
        GCGE_HIP_CHECK(hipEventRecord(s->ev_it[enq & 1], st));
        ++npend; cur = (cur + 1) % R; if (recompute) ++s->recompute_iters; ++s->dev_scalar_iters; s->spmm_calls++; s->spmm_cols += nrhs;
        ++enq;
        if (npend == J) flush_x_dev();
        if (enq >= 2) {   // what did iteration enq - 2 leave?  (iteration enq - 1 is already in the queue)
          GCGE_HIP_CHECK(hipEventSynchronize(s->ev_it[(enq - 2) & 1]));
          done = enq - 2;
          if (s->h_nact[done] == 0) { stop_at = done + 1; break; }
        }
      }
      GCGE_HIP_CHECK(hipStreamSynchronize(st));
      if (stop_at < 0) {      // the queue ran dry at max_iter (or the last iterations have not been looked at yet)
        stop_at = enq;
        for (int q = done + 1; q < enq; ++q) if (s->h_nact[q] == 0) { stop_at = q + 1; break; }
      }
      niter = stop_at;        // iterations that started with at least one active column, as the reference counts them
      flush_x_dev();
      GCGE_HIP_CHECK(hipMemcpyAsync(s->h_pin, d_last, sizeof(double), hipMemcpyDeviceToHost, st));
      GCGE_HIP_CHECK(hipStreamSynchronize(st));
      // columns streamed: every enqueued iteration walks all nrhs columns (the ring must stay complete), including the up to
      // two iterations enqueued before the host saw that nothing was active any more; columns still active at the START of
      // iteration q: the count iteration q - 1 left in h_nact
      s->col_iters += (long)nrhs * enq; s->surplus_iters += enq - niter;
      { long act = nact; for (int q = 1; q < niter; ++q) act += s->h_nact[q - 1]; s->active_col_iters += act; }
      s->niter = niter;
      s->residual = s->h_pin[0];
      return;
    }
    if (start_without_r && p0_done && nact > 0) { fprintf(stderr, "HIP_BlockPCG: the start sweep left no residual but the host loop needs one\n"); abort(); }
    // stored-product form without a stored residual: same rule as the device-scalar loop above
    const bool host_implicit_r = !recompute && R >= 3 && (g_residual_form == 1 || (g_residual_form == 0 && s->rate >= 1e-4 && s->max_iter <= 100));
    std::vector<double> bprev_host(nrhs, 0.0);
    while (niter < s->max_iter && nact > 0) {
      s->col_iters += nrhs; s->active_col_iters += nact;
      int alo = 0, ahi = nrhs;
      if (R == 1) {   // (with a ring every column takes part in every step: the slots must stay complete)
        while (alo < nrhs && !active[alo]) ++alo;
        while (ahi > alo && !active[ahi - 1]) --ahi;
      }
      int aw = ahi - alo;
      if (((alo & 1) || (aw & 1))) { alo &= ~1; ahi = (ahi + 1) & ~1; aw = ahi - alo; }   // keep 16-byte column pairs
      void** pcur = slots[cur];
      if (recompute) {
        if (gcge_hip_cg_pass1_mv(mat, pcur, alo, aw, pTw.data() + alo, wTw.data() + alo) != 0) {
          fprintf(stderr, "HIP_BlockPCG: first CG pass refused operands it had accepted\n"); abort();
        }
      } else apply(pcur, alo, s->mv_ws[2], alo, aw, pTw.data() + alo, wTw.data() + alo);
      s->spmm_calls++; s->spmm_cols += aw;
      {   // one all-reduce for both sums
        std::vector<double> both(2 * (size_t)aw);
        memcpy(both.data(), pTw.data() + alo, aw * sizeof(double)); memcpy(both.data() + aw, wTw.data() + alo, aw * sizeof(double));
        reduce_over_ranks(both.data(), 2 * aw);
        memcpy(pTw.data() + alo, both.data(), aw * sizeof(double)); memcpy(wTw.data() + alo, both.data() + aw, aw * sizeof(double));
      }
      for (int j = alo; j < ahi; ++j) {
        flag[j] = active[j]; coef[j] = 0.0; bet[j] = 0.0;
        if (!active[j]) continue;
        const double al = rho2[j] / pTw[j];
        double rho_pred = al * al * wTw[j] - rho2[j];
        if (!(rho_pred > 0.0) || !std::isfinite(rho_pred)) rho_pred = 0.0;   // cancellation: restart this column from r
        coef[j] = al; bet[j] = rho_pred / rho2[j];
      }
      upload(alo, aw, bet.data(), coef.data(), flag.data());   // d_coef = [beta | alpha]
      std::vector<double> newrho(aw);
      if (recompute) {
        if (gcge_hip_cg_pass2_mv(mat, pcur, s->mv_ws[0], slots[(cur + 1) % R], alo, aw, s->d_coef + s->cap, s->d_coef,
                                 s->d_flag, newrho.data()) != 0) {
          fprintf(stderr, "HIP_BlockPCG: second CG pass refused operands it had accepted\n"); abort();
        }
        for (int j = 0; j < nrhs; ++j) ahist[(size_t)npend * nrhs + j] = (j >= alo && j < ahi && active[j]) ? coef[j] : 0.0;
        ++npend; cur = (cur + 1) % R; ++s->recompute_iters;
      } else {
      double* part = gcge_hip_partial_ws((size_t)nb * aw + aw + (size_t)(J > 0 ? J : 0) * nrhs);
      if (R > 1) {
        long ldq;
        const double* pold = gcge_hip_mv_device_ptr(slots[cur], &ldq);
        double* pnew = gcge_hip_mv_device_ptr(slots[(cur + 1) % R], &ldq);
        if (host_implicit_r) {   // r is not stored: rebuilt from p_k and p_{k-1} (beta_{k-1} of the previous step, 0 at the start)
          const double* pprev = gcge_hip_mv_device_ptr(slots[(cur + R - 1) % R], &ldq);
          double* d_bprev = part + (size_t)nb * aw + aw;   // (behind the partial sums and their total; J * nrhs doubles are reserved there)
          GCGE_HIP_CHECK(hipStreamSynchronize(st));        // the pinned staging below may still feed the upload() above
          memcpy(s->h_pin, bprev_host.data() + alo, aw * sizeof(double));
          GCGE_HIP_CHECK(hipMemcpyAsync(d_bprev, s->h_pin, aw * sizeof(double), hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(cg_update_p_implicit<4>, dim3((unsigned)nb), dim3(256), 0, st, (long)n, dw + alo, ldw,
                             (niter == 0 ? pold : pprev) + alo, pold + alo, pnew + alo, ldp, aw, s->d_coef + s->cap, s->d_coef,
                             (const double*)d_bprev, s->d_flag, part, cg_tpr(aw));
          for (int j = alo; j < ahi; ++j) bprev_host[j] = active[j] ? bet[j] : 0.0;
          ++s->implicit_r_iters;
        } else
        hipLaunchKernelGGL(cg_update_rp<4>, dim3((unsigned)nb), dim3(256), 0, st, (long)n, dw + alo, ldw, dr + alo, ldr,
                           pold + alo, pnew + alo, ldp, aw, s->d_coef + s->cap, s->d_coef, s->d_flag, part, cg_tpr(aw));
        for (int j = 0; j < nrhs; ++j) ahist[(size_t)npend * nrhs + j] = (j >= alo && j < ahi && active[j]) ? coef[j] : 0.0;
        ++npend; cur = (cur + 1) % R;
      } else {
        hipLaunchKernelGGL(cg_update_all<4>, dim3((unsigned)nb), dim3(256), 0, st, (long)n, dw + alo, ldw, dr + alo, ldr,
                           dp + alo, ldp, dx + alo, ldx, aw, s->d_coef + s->cap, s->d_coef, s->d_flag, part, cg_tpr(aw));
      }
      gcge_hip_reduce_partials(part, (int)nb, aw, part + (size_t)nb * aw, st);
      GCGE_HIP_CHECK(hipMemcpyAsync(s->h_pin, part + (size_t)nb * aw, aw * sizeof(double), hipMemcpyDeviceToHost, st));
      GCGE_HIP_CHECK(hipStreamSynchronize(st));
      memcpy(newrho.data(), s->h_pin, aw * sizeof(double));
      }
      reduce_over_ranks(newrho.data(), aw);
      nact = 0;
      for (int j = alo; j < ahi; ++j) {
        if (!active[j]) continue;
        rho1[j] = rho2[j]; rho2[j] = newrho[j - alo];
        last_res[j] = sqrt(rho2[j]);
        active[j] = (last_res[j] > s->rate * init_res[j]) && (last_res[j] > s->tol * norm_b[j]);
        nact += active[j];
      }
      ++niter;
      if (npend == J && J > 0) flush_x();
    }
    flush_x();
    s->niter = niter;
    s->residual = last_res[0];
    return;
  }
  while (niter < s->max_iter && nact > 0) {
    // contiguous column window covering every active column and every pending x update
    int lo = 0, hi = nrhs;
    while (lo < nrhs && !active[lo] && !pend[lo]) ++lo;
    while (hi > lo && !active[hi - 1] && !pend[hi - 1]) --hi;
    int mw = hi - lo;
    // x += alpha' p (deferred) ; p = r + beta p
    for (int j = lo; j < hi; ++j) {
      flag[j] = (active[j] ? (niter == 0 ? 2 : 1) : 0) | (pend[j] ? 4 : 0);
      coef[j] = (active[j] && niter > 0) ? rho2[j] / rho1[j] : 0.0;
    }
    upload(lo, mw, coef.data(), aprev.data(), flag.data());
    launch_update_xp((long)n, dr + lo, ldr, dp + lo, ldp, dx + lo, ldx, mw, s->d_coef, s->d_coef + s->cap, s->d_flag, st);
    for (int j = lo; j < hi; ++j) pend[j] = 0;
    // w = A p and pTw on the window of ACTIVE columns
    int alo = lo, ahi = hi;
    while (alo < hi && !active[alo]) ++alo;
    while (ahi > alo && !active[ahi - 1]) --ahi;
    const int aw = ahi - alo;
    apply(s->mv_ws[1], alo, s->mv_ws[2], alo, aw, pTw.data() + alo);
    s->spmm_calls++; s->spmm_cols += aw;
    reduce_over_ranks(pTw.data() + alo, aw);
    // r -= alpha w ; rho2 = diag(r^T r)
    for (int j = alo; j < ahi; ++j) {
      rho1[j] = rho2[j]; coef[j] = active[j] ? rho2[j] / pTw[j] : 0.0; flag[j] = active[j];
      if (active[j]) { aprev[j] = coef[j]; pend[j] = 1; }
    }
    upload(alo, aw, coef.data(), coef.data(), flag.data());
    double* part = gcge_hip_partial_ws((size_t)nb * aw + aw);
    hipLaunchKernelGGL(cg_update_r, dim3((unsigned)nb), dim3(256), 0, st, (long)n, dw + alo, ldw, dr + alo, ldr, aw,
                       s->d_coef, s->d_flag, part, rpb);
    gcge_hip_reduce_partials(part, (int)nb, aw, part + (size_t)nb * aw, st);
    GCGE_HIP_CHECK(hipMemcpyAsync(s->h_pin, part + (size_t)nb * aw, aw * sizeof(double), hipMemcpyDeviceToHost, st));
    GCGE_HIP_CHECK(hipStreamSynchronize(st));
    std::vector<double> newrho(s->h_pin, s->h_pin + aw);
    reduce_over_ranks(newrho.data(), aw);
    nact = 0;
    for (int j = alo; j < ahi; ++j) {
      if (!active[j]) continue;
      rho2[j] = newrho[j - alo];
      last_res[j] = sqrt(rho2[j]);
      active[j] = (last_res[j] > s->rate * init_res[j]) && (last_res[j] > s->tol * norm_b[j]);
      nact += active[j];
    }
    ++niter;
  }
  {   // flush the x updates still pending
    int lo = 0, hi = nrhs;
    while (lo < nrhs && !pend[lo]) ++lo;
    while (hi > lo && !pend[hi - 1]) --hi;
    if (hi > lo) {
      const int mw = hi - lo;
      for (int j = lo; j < hi; ++j) { flag[j] = pend[j] ? 4 : 0; coef[j] = 0.0; }
      upload(lo, mw, coef.data(), aprev.data(), flag.data());
      launch_update_xp((long)n, dr + lo, ldr, dp + lo, ldp, dx + lo, ldx, mw, s->d_coef, s->d_coef + s->cap, s->d_flag, st);
    }
  }
  s->niter = niter;
  s->residual = last_res[0];
}

// C-ABI: install the fused solver (GCG: pass flag = 1 to the harness / -gcge_user_defined_multi_lin_sol 1)
extern "C" void gcge_hip_bpcg_setup(struct OPS_* ops, int max_iter, double rate, double tol, const char* tol_type) {
  // "user": the column scales BlockPCG finds in its scalar scratch (src/ops_lin_sol.c:186-192) are read from
  // GCGE_GetLinearSolverUserScale at solve time (the GCG driver publishes lambda_j + sigma there); anything else is refused
  if (tol_type != nullptr && strcmp(tol_type, "abs") != 0 && strcmp(tol_type, "rel") != 0 && strcmp(tol_type, "user") != 0) {
    fprintf(stderr, "gcge_hip_bpcg_setup: tol_type \"%s\" is not one of \"abs\", \"rel\", \"user\"\n", tol_type);
    abort();
  }
  g_bpcg.max_iter = max_iter; g_bpcg.rate = rate; g_bpcg.tol = tol;
  strncpy(g_bpcg.tol_type, tol_type ? tol_type : "abs", 7); g_bpcg.tol_type[7] = 0;
  ops->multi_linear_solver_workspace = (void*)&g_bpcg;
  ops->MultiLinearSolver = HIP_BlockPCG;
  GCGE_SetRhsScaleCapability((void*)HIP_BlockPCG);   // b = x diag(scale) need not be formed (see HIP_BlockPCG_run)
}
// Optional: create the solver's blocks (r, p, w and the ring of direction slots) NOW, for systems with the rows of `mv_like` and up
// to `ncols` right-hand sides, instead of inside the first solve.  The counterpart of the reference's EigenSolverCreateWorkspace /
// MultiLinearSolverSetup_BlockPCG (src/ops_lin_sol.c:439-465), which take the CG's blocks from the caller before any timer
// starts: a ring of 15 slots at BASELINE config 2's shape is 129 GB of fresh device memory, which the driver clears on first
// use (4 s in a new process — otherwise spent inside the first call).  Collective when a communicator exists.
extern "C" int gcge_hip_bpcg_prepare(struct OPS_* ops, void* mat, void** mv_like, int ncols) {
  if (ops == nullptr || ops->MultiLinearSolver != HIP_BlockPCG || mv_like == nullptr || ncols <= 0) return -1;
  HipBpcg* s = (HipBpcg*)ops->multi_linear_solver_workspace;
  double sigma = 0.0; void* matB = nullptr;
  GCGE_GetLinearSolverShift(&sigma, &matB);
  bpcg_shape(s, gcge_hip_mv_nrows(mv_like), ncols, mv_like, ops);
  bpcg_ring(s, mat, mv_like, sigma, ops);
  GCGE_HIP_CHECK(hipStreamSynchronize((hipStream_t)gcge_hip_stream()));
  return s->ring_len;
}
extern "C" void gcge_hip_bpcg_stats(long* spmm_calls, long* spmm_cols, int* last_niter) {
  if (spmm_calls) *spmm_calls = g_bpcg.spmm_calls;
  if (spmm_cols) *spmm_cols = g_bpcg.spmm_cols;
  if (last_niter) *last_niter = g_bpcg.niter;
}
extern "C" long gcge_hip_bpcg_recompute_iters(void) { return g_bpcg.recompute_iters; }
extern "C" long gcge_hip_bpcg_device_scalar_iters(void) { return g_bpcg.dev_scalar_iters; }
extern "C" long gcge_hip_bpcg_implicit_r_iters(void) { return g_bpcg.implicit_r_iters; }
extern "C" long gcge_hip_bpcg_stored_dev_iters(void) { return g_bpcg.stored_dev_iters; }
extern "C" long gcge_hip_bpcg_fused_starts(void) { return g_bpcg.fused_starts; }
extern "C" void gcge_hip_bpcg_time_stats(long* iters, double* seconds, int reset) {
  if (iters) *iters = g_bpcg.total_iters;
  if (seconds) *seconds = g_bpcg.total_seconds;
  if (reset) { g_bpcg.total_iters = 0; g_bpcg.total_seconds = 0.0; }
}
extern "C" void gcge_hip_bpcg_column_stats(long* col_iters, long* active_col_iters) {
  if (col_iters) *col_iters = g_bpcg.col_iters;
  if (active_col_iters) *active_col_iters = g_bpcg.active_col_iters;
}
extern "C" long gcge_hip_bpcg_surplus_iters(void) { return g_bpcg.surplus_iters; }
// the fused CG as the smoother of BlockAMG for the HIP table (GCGE_SetBlockAMGSmoother, registered by OPS_HIP_Set): same stopping
// rules as MultiLinearSolverSetup_BlockPCG, its own blocks per level (parked sets above)
extern "C" void gcge_hip_amg_smoother_setup(int max_iter, double rate, double tol, const char* tol_type, struct OPS_* ops) {
  gcge_hip_bpcg_setup(ops, max_iter, rate, tol, tol_type);
}
extern "C" double gcge_hip_amg_smoother_residual(struct OPS_* ops) { (void)ops; return g_bpcg.residual; }
extern "C" void gcge_hip_bpcg_release(struct OPS_* ops) {
  for (int i = 0; i < g_nparked; ++i) bpcg_destroy_set(&g_parked[i], ops);
  g_nparked = 0;
  for (int i = 1; i < g_bpcg.ring_len; ++i) if (g_bpcg.ring[i]) ops->MultiVecDestroy(&g_bpcg.ring[i], g_bpcg.ws_cols, ops);
  g_bpcg.ring_len = 0;
  for (int i = 0; i < 4; ++i)
    if (g_bpcg.mv_ws[i]) ops->MultiVecDestroy(&g_bpcg.mv_ws[i], g_bpcg.ws_cols, ops);
  g_bpcg.ws_cols = 0; g_bpcg.ws_rows = 0;
}
