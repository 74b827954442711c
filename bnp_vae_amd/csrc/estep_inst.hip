// One translation unit per padded latent dim (compiled with -DBNPVI_DP=<32|64|128|256>) so the
// unrolled MFMA bodies build in parallel.  See estep_kernels.h.
#include <stdlib.h>
#include <string.h>

#include "estep_rows1.h"
#include "estep_stream2.h"

#ifndef BNPVI_DP
#error "compile with -DBNPVI_DP=<padded latent dim>"
#endif

#ifndef BNPVI_FUSED_PF_SMALL
#define BNPVI_FUSED_PF_SMALL 1
#endif

namespace bnpvi {

template <int DP, int NT>
static size_t fused_lds_bytes() {
  return (size_t)NT * (DP / 32) * 2 * 64 * 16 + (size_t)kLT * NT * 4 + (size_t)kLT * NT * 16;
}

// workgroups of kEWaves waves that fit a CU at once: one wave per SIMD each, so the count is
// the waves/SIMD the register allocation allows (512 registers per lane per SIMD, granule 8),
// capped by the LDS and by 8 (MI355X_MICROARCH.md, register files)
static int resident_blocks(const void* kern, size_t lds, int* out) {
  hipFuncAttributes fa;
  BNPVI_HIP(hipFuncGetAttributes(&fa, kern));
  int regs = fa.numRegs > 0 ? fa.numRegs : 128;
  regs = (regs + 7) / 8 * 8;
  int b = 512 / regs;
  if (b > 8) b = 8;
  if (b < 1) b = 1;
  const size_t per = lds + (size_t)fa.sharedSizeBytes;
  if (per > 0) {
    const int bl = (int)((size_t)kMaxLdsBytes / per);
    if (bl < b) b = bl;
  }
  *out = b < 1 ? 1 : b;
  return BNPVI_OK;
}

template <int DP, int NT, bool HARD, bool STATS>
static int launch_fused(const EstepArgs& a) {
  // Z tiles in flight per wave.  At DP = 64, NT <= 4 a second tile (16 registers) measured 5 % faster
  // until the prior entries were requested ahead of the MFMA block (another 16 registers held
  // across it): with both the kernel spills at 3 waves per SIMD; priors early + one tile wins
  // (106.6 us against 109-110 us; both at 2 waves per SIMD: 114 us).
  constexpr int PF = (DP <= 64 && NT <= 4) ? BNPVI_FUSED_PF_SMALL : 1;
  const int L = a.tree->L, Lp = a.tree->Lp;
  const size_t lds = fused_lds_bytes<DP, NT>();
  auto k = estep_fused_kernel<DP, NT, PF, HARD, STATS>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 1;
  int rc = resident_blocks((const void*)k, lds, &per_cu);
  if (rc) return rc;
  const int64_t ntiles = ceil_div(a.N, kFT);
  int64_t blocks = ceil_div(ntiles, kEWaves);
  const int64_t cap = (int64_t)a.ctx->num_cu * per_cu;
  if (blocks > cap) blocks = cap;
  double* eslab = nullptr;
  if (STATS) {
    rc = a.ctx->slabs.ensure((size_t)blocks * 2 * L * sizeof(double));
    if (rc) return rc;
    eslab = (double*)a.ctx->slabs.ptr;
  }
  const bool timed = a.ctx->profiling && a.ctx->timer[BNPVI_PROF_ESTEP].begin(a.stream);
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kEWaves * 64), lds, a.stream, a.Z, a.ldz, a.vid,
                     a.N, L, Lp, a.apack, a.aavec, a.a32, a.scal, a.prior, a.half_sz, a.phi, a.logphi,
                     a.assign, eslab);
  if (timed) a.ctx->timer[BNPVI_PROF_ESTEP].end(a.stream);
  BNPVI_LAUNCH_CHECK();
  if (STATS) return estep_reduce_slabs(a, eslab, (int)blocks);
  return BNPVI_OK;
}

template <int DP, int NT>
static int run_fused(const EstepArgs& a) {
  if (a.mode == BNPVI_ESTEP_HARD) return launch_fused<DP, NT, true, false>(a);
  if (a.estats) return launch_fused<DP, NT, false, true>(a);
  return launch_fused<DP, NT, false, false>(a);
}

template <int DP>
static int run_streamed(const EstepArgs& a) {
  constexpr int NT = kStreamNT;
  const int Lp = a.tree->Lp;
  const int tiles = Lp / kLT;
  const int nchunk = (tiles + NT - 1) / NT;
  BNPVI_REQUIRE(nchunk <= 65535, "estep: too many leaf chunks (%d)", nchunk);
  // per (row, chunk) partials + one (max, log-sum-exp) per row for the column-window normalise pass
  int rc = a.ctx->part.ensure(((size_t)a.N * nchunk + (size_t)a.N) * sizeof(float2));
  if (rc) return rc;
  float2* part = (float2*)a.ctx->part.ptr;
  const int64_t ntiles = ceil_div(a.N, kFT);
  // operands of one chunk staged in LDS when they fit (8 waves share them), else read from L2
  constexpr int kStagedWaves = 8;
  const size_t lds = (size_t)NT * (DP / 32) * 2 * 64 * sizeof(f16x8) + (size_t)kLT * NT * sizeof(float);
  const char* env = dev_getenv("BNPVI_ESTEP_STREAM");                   // development switch: "l2"
  const bool staged = lds <= (size_t)kMaxLdsBytes && !(env && strcmp(env, "l2") == 0);
  const int waves = staged ? kStagedWaves : kEWaves;
  const int per_cu = staged ? (int)((size_t)kMaxLdsBytes / lds) : 8;
  int64_t bx = ceil_div(ntiles, waves);
  int64_t cap = (int64_t)a.ctx->num_cu * (per_cu > 8 ? 8 : per_cu) / nchunk;
  if (cap < 1) cap = 1;
  if (bx > cap) bx = cap;
  dim3 grid((unsigned)bx, (unsigned)nchunk);
  // soft mode with staged operands: frame fragments packed once per call, FT frame tiles per wave and unit
  // (estep_stream2.h); falls back to the kernel below when the fragment image cannot be allocated
  const char* env2 = dev_getenv("BNPVI_ESTEP_STREAM2");                 // development switch: "off"
  if (a.mode == BNPVI_ESTEP_SOFT && staged && !(env2 && strcmp(env2, "off") == 0)) {
    constexpr int FT = kStream2FT;
    const size_t frag_bytes = (size_t)ntiles * (DP / 32) * 2 * 64 * sizeof(f16x8);
    const size_t meta_off = (frag_bytes + 255) / 256 * 256;
    if (a.ctx->zpack.ensure(meta_off + (size_t)ntiles * kFT * sizeof(float2)) == BNPVI_OK) {
      f16x8* zpack = (f16x8*)a.ctx->zpack.ptr;
      float2* zmeta = (float2*)((char*)a.ctx->zpack.ptr + meta_off);
      const bool timed2 = a.ctx->profiling && a.ctx->timer[BNPVI_PROF_ESTEP].begin(a.stream);
      int64_t pb = ceil_div(ntiles, (int64_t)4);
      if (pb > (int64_t)a.ctx->num_cu * 32) pb = (int64_t)a.ctx->num_cu * 32;
      hipLaunchKernelGGL(pack_latents_kernel<DP>, dim3((unsigned)pb), dim3(256), 0, a.stream, a.Z, a.ldz, a.N, zpack,
                         zmeta);
      const int64_t nunits = ceil_div(ntiles, (int64_t)FT);
      int64_t ux = ceil_div(nunits, (int64_t)kS2Waves);
      if (ux > cap) ux = cap;
      auto k2 = estep_logits2_kernel<DP, NT, FT>;
      if (lds > 48 * 1024)
        BNPVI_HIP(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k2, dim3((unsigned)ux, (unsigned)nchunk), dim3(kS2Waves * 64), lds, a.stream, a.Z, a.ldz,
                         a.vid, a.N, a.tree->L, Lp, zpack, zmeta, a.apack, a.aavec, a.a32, a.scal, a.prior,
                         a.half_sz, a.phi, part, nchunk);
      if (timed2) a.ctx->timer[BNPVI_PROF_ESTEP].end(a.stream);
      BNPVI_LAUNCH_CHECK();
      return estep_finish_streamed(a, part, nchunk);
    }
  }
  const bool timed = a.ctx->profiling && a.ctx->timer[BNPVI_PROF_ESTEP].begin(a.stream);
#define BNPVI_LAUNCH_LOGITS(HARD_, WAVES_, LDS_)                                                           \
  do {                                                                                                    \
    auto k = estep_logits_kernel<DP, NT, HARD_, WAVES_>;                                                  \
    if ((LDS_) > 48 * 1024)                                                                               \
      BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_))); \
    hipLaunchKernelGGL(k, grid, dim3(((WAVES_) > 0 ? (WAVES_) : kEWaves) * 64), (LDS_), a.stream, a.Z, a.ldz, \
                       a.vid, a.N, a.tree->L, Lp, a.apack, a.aavec, a.a32, a.scal, a.prior, a.half_sz,    \
                       a.phi, part, nchunk);                                                              \
  } while (0)
  if (a.mode == BNPVI_ESTEP_HARD) {
    if (staged) BNPVI_LAUNCH_LOGITS(true, kStagedWaves, lds); else BNPVI_LAUNCH_LOGITS(true, 0, (size_t)0);
  } else {
    if (staged) BNPVI_LAUNCH_LOGITS(false, kStagedWaves, lds); else BNPVI_LAUNCH_LOGITS(false, 0, (size_t)0);
  }
#undef BNPVI_LAUNCH_LOGITS
  if (timed) a.ctx->timer[BNPVI_PROF_ESTEP].end(a.stream);
  BNPVI_LAUNCH_CHECK();
  return estep_finish_streamed(a, part, nchunk);
}

// Two-pass row kernel (estep_rows.h): soft mode, packed leaf operands resident in LDS.
template <int DP>
static bool rows_fits(const EstepArgs& a) {
  const int NTp = round_up(a.tree->Lp / kLT, kRChunk);
  return a.mode == BNPVI_ESTEP_SOFT && a.assign == nullptr &&
         RowsLds<DP>::bytes(NTp, a.estats != nullptr) <= (size_t)kMaxLdsBytes;
}

template <int DP, bool STATS, int NCH = 0>
static int launch_rows(const EstepArgs& a) {
  constexpr int kW = rows_waves(NCH);
  const int L = a.tree->L, Lp = a.tree->Lp;
  const int NTp = NCH > 0 ? NCH * kRChunk : round_up(Lp / kLT, kRChunk);
  const size_t lds = RowsLds<DP>::bytes(NTp, STATS);
  auto k = estep_rows_kernel<DP, STATS, NCH>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)((size_t)kMaxLdsBytes / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 32 / kW ? 32 / kW : per_cu);
  const int64_t ntiles = ceil_div(a.N, kFT);
  // contiguous tile ranges per workgroup; at most 512 tiles (8192 frames) so that the fp32
  // statistics table of a workgroup stays exact enough before its promotion to fp64
  int64_t tpw = ceil_div(ntiles, (int64_t)a.ctx->num_cu * per_cu);
  tpw = round_up((int)(tpw < kW ? kW : tpw), kW);
  if (tpw > 512 / kW * kW) tpw = 512 / kW * kW;
  const int64_t blocks = ceil_div(ntiles, tpw);
  double* eslab = nullptr;
  if (STATS) {
    int rc = a.ctx->slabs.ensure((size_t)blocks * 2 * L * sizeof(double));
    if (rc) return rc;
    eslab = (double*)a.ctx->slabs.ptr;
  }
  const bool timed = a.ctx->profiling && a.ctx->timer[BNPVI_PROF_ESTEP].begin(a.stream);
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kW * 64), lds, a.stream, a.Z, a.ldz, a.vid, a.N, L,
                     Lp, NTp, a.apack, a.aavec, a.a32, a.scal, a.prior, a.half_sz, a.phi, a.logphi, eslab,
                     (int)tpw);
  if (timed) a.ctx->timer[BNPVI_PROF_ESTEP].end(a.stream);
  BNPVI_LAUNCH_CHECK();
  if (STATS) return estep_reduce_slabs(a, eslab, (int)blocks);
  return BNPVI_OK;
}

// Row kernel selection: up to 256 leaves (D <= 64), when no statistics are asked for, the logits
// of a frame stay in registers and the kernel makes one pass (NCH = 2 or 4 chunks of 64 leaves);
// otherwise two passes.
template <int DP, bool STATS>
static int run_rows(const EstepArgs& a) {
  if constexpr (DP <= 64 && !STATS) {
    const char* e = dev_getenv("BNPVI_ROWS_PASSES");            // development switch: "2" forces two passes
    const int chunks = ceil_div(a.tree->Lp, kLT * kRChunk);
    const bool one = !(e && e[0] == '2');
    if (one && chunks <= 2 && RowsLds<DP>::bytes(2 * kRChunk, STATS) <= (size_t)kMaxLdsBytes)
      return launch_rows<DP, STATS, 2>(a);
    if (one && chunks <= 4 && RowsLds<DP>::bytes(4 * kRChunk, STATS) <= (size_t)kMaxLdsBytes)
      return launch_rows<DP, STATS, 4>(a);
  }
  return launch_rows<DP, STATS, 0>(a);
}

// One-pass row kernel (estep_rows1.h): soft mode, no log phi / arg-max, at most Rows1Lds<DP>::max_chunks()
// chunks of 64 leaves, per-leaf mass as the only statistic (S2 left 0: the caller asked for that with
// BNPVI_ESTEP_STATS_S0, or for no statistics at all).
template <int DP, int NCH>
static int launch_rows1(const EstepArgs& a) {
  const int L = a.tree->L, Lp = a.tree->Lp;
  constexpr size_t lds = Rows1Lds<DP>::bytes(NCH);
  auto k = estep_rows1_kernel<DP, NCH>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)((size_t)kMaxLdsBytes / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 16 / kR1Waves ? 16 / kR1Waves : per_cu);     // two waves per SIMD
  const int64_t ntiles = ceil_div(a.N, kFT);
  // contiguous tile ranges per workgroup, a multiple of the waves; at most 512 tiles (8192 frames) per
  // workgroup: its fp64 table takes one fp32 partial per (tile, leaf)
  int64_t tpw = ceil_div(ntiles, (int64_t)a.ctx->num_cu * per_cu);
  tpw = round_up((int)(tpw < kR1Waves ? kR1Waves : tpw), kR1Waves);
  if (tpw > 512) tpw = 512;
  const int64_t blocks = ceil_div(ntiles, tpw);
  double* eslab = nullptr;
  if (a.estats) {
    int rc = a.ctx->slabs.ensure((size_t)blocks * 2 * L * sizeof(double));
    if (rc) return rc;
    eslab = (double*)a.ctx->slabs.ptr;
  }
  const bool timed = a.ctx->profiling && a.ctx->timer[BNPVI_PROF_ESTEP].begin(a.stream);
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kR1Waves * 64), lds, a.stream, a.Z, a.ldz, a.vid, a.N, L, Lp,
                     a.apack, a.aavec, a.a32, a.scal, a.prior, a.half_sz, a.phi, eslab, (int)tpw);
  if (timed) a.ctx->timer[BNPVI_PROF_ESTEP].end(a.stream);
  BNPVI_LAUNCH_CHECK();
  if (a.estats) return estep_reduce_slabs(a, eslab, (int)blocks);
  return BNPVI_OK;
}

template <int DP, int NCH>
static int run_rows1(const EstepArgs& a, int chunks) {
  if constexpr (NCH < 2) {
    (void)chunks;
    set_error("estep: no one-pass row kernel for this shape");
    return BNPVI_EUNSUPPORTED;
  } else {
    if (chunks == NCH) return launch_rows1<DP, NCH>(a);
    return run_rows1<DP, NCH - 1>(a, chunks);
  }
}

template <int DP>
static bool rows1_fits(const EstepArgs& a) {
  const int chunks = (int)ceil_div(a.tree->Lp, kLT * kRChunk);
  return a.mode == BNPVI_ESTEP_SOFT && a.assign == nullptr && a.logphi == nullptr &&
         (a.estats == nullptr || a.s0_only) && chunks >= 2 && chunks <= Rows1Lds<DP>::max_chunks();
}

template <>
int estep_dispatch<BNPVI_DP>(const EstepArgs& a) {
  constexpr int DP = BNPVI_DP;
  const int Lp = a.tree->Lp;
  const bool hard_stats = a.mode == BNPVI_ESTEP_HARD && a.estats != nullptr;
  const char* force = dev_getenv("BNPVI_ESTEP_PATH");         // development switch: "rows" | "fused"
  const bool want_rows = force ? strcmp(force, "rows") == 0 : Lp > 64;
  if (Lp > 64 && !force && rows1_fits<DP>(a))
    return run_rows1<DP, Rows1Lds<DP>::max_chunks()>(a, (int)ceil_div(Lp, kLT * kRChunk));
  // 65 .. 128 padded leaves with S2 wanted in the kernel (the engine asks for it up to 100 leaves): the fused kernel
  // with eight leaf tiles keeps a frame's whole row of logits in registers, the two-pass row kernel computes every
  // product twice -- 2M frames, L = 81 / 100 / 121: D = 128 569 / 649 / 672 us against 1120 / 1230 / 1230,
  // D = 64 354 / 392 / 408 against 543 / 556 / 574 (tools/dev/g28.sh).  (S0-only and no statistics stay on the one-pass
  // row kernel above, which is faster still; log phi / arg-max requests and hard mode stay where they were.)
  // D = 256 (1M frames, L = 81): 1111 us against 2199.
  if (!force && Lp > 64 && Lp <= 128 && a.mode == BNPVI_ESTEP_SOFT && a.assign == nullptr && a.logphi == nullptr &&
      a.estats != nullptr && fused_lds_bytes<DP, 8>() <= (size_t)kMaxLdsBytes)
    return run_fused<DP, 8>(a);
  if (want_rows && !(force && strcmp(force, "fused") == 0) && rows_fits<DP>(a))
    return a.estats ? run_rows<DP, true>(a) : run_rows<DP, false>(a);
  if (!hard_stats) {
    // smallest instantiated tile count that covers the padded leaves and fits the LDS
    if (Lp <= 32 && fused_lds_bytes<DP, 2>() <= (size_t)kMaxLdsBytes) return run_fused<DP, 2>(a);
    if (Lp <= 64 && fused_lds_bytes<DP, 4>() <= (size_t)kMaxLdsBytes) return run_fused<DP, 4>(a);
    if (Lp <= 128 && fused_lds_bytes<DP, 8>() <= (size_t)kMaxLdsBytes) return run_fused<DP, 8>(a);
    if constexpr (DP <= 128) {
      if (Lp <= 256 && fused_lds_bytes<DP, 16>() <= (size_t)kMaxLdsBytes) return run_fused<DP, 16>(a);
    }
    if constexpr (DP <= 64) {
      if (Lp <= 512 && fused_lds_bytes<DP, 32>() <= (size_t)kMaxLdsBytes) return run_fused<DP, 32>(a);
    }
  }
  return run_streamed<DP>(a);
}

}  // namespace bnpvi
