// One translation unit per padded latent dim (compiled with -DBNPVI_DP=<8|16|32|64|128|256>) so
// the unrolled MFMA bodies build in parallel.  See estep_kernels.h.
#include "estep_kernels.h"

#ifndef BNPVI_DP
#error "compile with -DBNPVI_DP=<padded latent dim>"
#endif

namespace bnpvi {

template <int DP, int NT>
static int run_fused(const EstepArgs& a) {
  constexpr int Lp = 32 * NT;
  const int L = a.tree->L;
  const int64_t ntiles = ceil_div(a.N, 32);
  int64_t blocks = ceil_div(ntiles, kEWaves);
  const int64_t cap = (int64_t)a.ctx->num_cu * 2;
  if (blocks > cap) blocks = cap;
  double* eslab = nullptr;
  if (a.estats) {
    int rc = a.ctx->slabs.ensure((size_t)blocks * 2 * L * sizeof(double));
    if (rc) return rc;
    eslab = (double*)a.ctx->slabs.ptr;
  }
  size_t lds = (size_t)NT * (DP / 8) * 64 * 16 + (size_t)Lp * 4 + (size_t)Lp * 16;
  BNPVI_REQUIRE(lds <= (size_t)kMaxLdsBytes, "estep: LDS %zu too large", lds);
  auto k = a.estats ? estep_fused_kernel<DP, NT, true> : estep_fused_kernel<DP, NT, false>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const bool timed = a.ctx->profiling && a.ctx->timer[BNPVI_PROF_ESTEP].begin(a.stream);
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kEWaves * 64), lds, a.stream, a.Z, a.ldz, a.vid,
                     a.N, L, a.apack, a.cvec, a.prior, a.half_sz, a.phi, a.logphi, a.assign, eslab);
  if (timed) a.ctx->timer[BNPVI_PROF_ESTEP].end(a.stream);
  BNPVI_LAUNCH_CHECK();
  if (a.estats) return estep_reduce_slabs(a, eslab, (int)blocks);
  return BNPVI_OK;
}

template <int DP>
static int run_streamed(const EstepArgs& a) {
  constexpr int NT = kStreamNT;
  const int Lp = a.tree->Lp;
  const int tiles = Lp / 32;
  const int nchunk = (tiles + NT - 1) / NT;
  BNPVI_REQUIRE(nchunk <= 65535, "estep: too many leaf chunks (%d)", nchunk);
  int rc = a.ctx->part.ensure((size_t)a.N * nchunk * sizeof(float2));
  if (rc) return rc;
  float2* part = (float2*)a.ctx->part.ptr;
  const int64_t ntiles = ceil_div(a.N, 32);
  int64_t bx = ceil_div(ntiles, kEWaves);
  int64_t cap = (int64_t)a.ctx->num_cu * 8 / nchunk;
  if (cap < 1) cap = 1;
  if (bx > cap) bx = cap;
  dim3 grid((unsigned)bx, (unsigned)nchunk);
  const bool timed = a.ctx->profiling && a.ctx->timer[BNPVI_PROF_ESTEP].begin(a.stream);
  if (a.mode == BNPVI_ESTEP_HARD)
    hipLaunchKernelGGL((estep_logits_kernel<DP, NT, true>), grid, dim3(kEWaves * 64), 0, a.stream, a.Z,
                       a.ldz, a.vid, a.N, a.tree->L, Lp, a.apack, a.cvec, a.prior, a.half_sz, a.phi,
                       part, nchunk);
  else
    hipLaunchKernelGGL((estep_logits_kernel<DP, NT, false>), grid, dim3(kEWaves * 64), 0, a.stream, a.Z,
                       a.ldz, a.vid, a.N, a.tree->L, Lp, a.apack, a.cvec, a.prior, a.half_sz, a.phi,
                       part, nchunk);
  if (timed) a.ctx->timer[BNPVI_PROF_ESTEP].end(a.stream);
  BNPVI_LAUNCH_CHECK();
  return estep_finish_streamed(a, part, nchunk);
}

template <>
int estep_dispatch<BNPVI_DP>(const EstepArgs& a) {
  if (a.mode == BNPVI_ESTEP_SOFT) {
    switch (a.tree->Lp / 32) {
      case 1: return run_fused<BNPVI_DP, 1>(a);
      case 2: return run_fused<BNPVI_DP, 2>(a);
      case 3: return run_fused<BNPVI_DP, 3>(a);
      case 4: return run_fused<BNPVI_DP, 4>(a);
      default: break;
    }
  }
  return run_streamed<BNPVI_DP>(a);
}

}  // namespace bnpvi
