// Rate of the CG direction update without a stored residual (block_pcg.hip: cg_update_p_implicit — reads w, p_k, p_{k-1}, writes
// p_{k+1}, sums r.r per column) as a function of how the rows are dealt to the workgroups.  Measurement aid.
//   hipcc -O3 --offload-arch=gfx950 tools/update_bench.hip -o tools/_bin/update_bench && tools/_bin/update_bench [rows]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

// LAYOUT 0: every workgroup a contiguous slab of rows (what block_pcg.hip does); 1: workgroups interleaved, UNR x 8 rows at a time
template <int UNR, int LAYOUT, bool NT>
__global__ __launch_bounds__(256) void upd(long nrows, const double* __restrict__ w, const double* __restrict__ pprev,
                                           const double* __restrict__ pold, double* __restrict__ pnew, double* __restrict__ partial) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = 2 * tx;
  const double a0 = 0.3, a1 = 0.4, cb0 = 0.5, cb1 = 0.6, bp0 = 0.7, bp1 = 0.8;
  double s0 = 0.0, s1 = 0.0;
  auto ld = [&](const double* p) { return NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p)) : *reinterpret_cast<const v2d*>(p); };
  auto body = [&](long row) {
    v2d wv[UNR], qv[UNR], pv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long rr = row + u * 8;
      wv[u] = ld(w + rr * 64 + j); qv[u] = ld(pprev + rr * 64 + j); pv[u] = *reinterpret_cast<const v2d*>(pold + rr * 64 + j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long rr = row + u * 8;
      v2d rn = {fma(-a0, wv[u].x, fma(-bp0, qv[u].x, pv[u].x)), fma(-a1, wv[u].y, fma(-bp1, qv[u].y, pv[u].y))};
      v2d pn = {fma(cb0, pv[u].x, rn.x), fma(cb1, pv[u].y, rn.y)};
      __builtin_nontemporal_store(pn, reinterpret_cast<v2d*>(pnew + rr * 64 + j));
      s0 = fma(rn.x, rn.x, s0); s1 = fma(rn.y, rn.y, s1);
    }
  };
  const long group = 8L * UNR;
  if (LAYOUT == 0) {
    const long slab = (((nrows + gridDim.x - 1) / gridDim.x) + group - 1) / group * group;
    const long rend = min(nrows, ((long)blockIdx.x + 1) * slab);
    for (long row = (long)blockIdx.x * slab + ty; row + (UNR - 1) * 8 < rend; row += group) body(row);
  } else {
    for (long row = (long)blockIdx.x * group + ty; row + (UNR - 1) * 8 < nrows; row += group * gridDim.x) body(row);
  }
  __shared__ double red[256][2];
  red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1;
  __syncthreads();
  if (ty == 0) {
    for (int q = 1; q < 8; ++q) { s0 += red[q * 32 + tx][0]; s1 += red[q * 32 + tx][1]; }
    partial[(long)blockIdx.x * 64 + j] = s0; partial[(long)blockIdx.x * 64 + j + 1] = s1;
  }
}

template <int UNR, int LAYOUT, bool NT>
static void run(long n, const double* w, const double* q, const double* p, double* pn, double* part, int nb) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((upd<UNR, LAYOUT, NT>), dim3(nb), dim3(256), 0, 0, n, w, q, p, pn, part);
  CK(hipEventRecord(e0, 0));
  const int reps = 6;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((upd<UNR, LAYOUT, NT>), dim3(nb), dim3(256), 0, 0, n, w, q, p, pn, part);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("rows in flight %d, %s, %s loads, %5d workgroups: %.3f ms  %.2f TB/s\n", UNR, LAYOUT ? "interleaved" : "slabs      ", NT ? "nt   " : "plain", nb, ms,
         (double)n * 512 * 4 / ms * 1e-9);
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 5000211;
  double *w, *q, *p, *pn, *part;
  CK(hipMalloc(&w, n * 512)); CK(hipMalloc(&q, n * 512)); CK(hipMalloc(&p, n * 512)); CK(hipMalloc(&pn, n * 512)); CK(hipMalloc(&part, 16384 * 64 * 8));
  CK(hipMemset(w, 0, n * 512)); CK(hipMemset(q, 0, n * 512)); CK(hipMemset(p, 0, n * 512)); CK(hipMemset(pn, 0, n * 512));
  for (int nb : {1024, 2048, 4096, 8192}) {
    run<4, 0, true>(n, w, q, p, pn, part, nb); run<4, 1, true>(n, w, q, p, pn, part, nb);
    run<2, 0, true>(n, w, q, p, pn, part, nb); run<2, 1, true>(n, w, q, p, pn, part, nb);
    run<8, 0, true>(n, w, q, p, pn, part, nb); run<8, 1, true>(n, w, q, p, pn, part, nb);
    run<4, 0, false>(n, w, q, p, pn, part, nb); run<4, 1, false>(n, w, q, p, pn, part, nb);
  }
  return 0;
}
