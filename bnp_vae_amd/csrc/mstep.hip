// M-step data pass for gfx950: responsibility-weighted sufficient statistics
//   S0[l] = sum_n phi[n,l],  S1[l,:] = sum_n phi[n,l] z_n,  Q[l,:] = sum_n phi[n,l] z_n^2
//   m[v,l] = sum_{n in video v} phi[n,l]
// replacing the per-leaf NumPy passes of /root/reference/src/var_inf.py:265-297 and :400-403.
//
// R^T.Z is a GEMM whose contraction index is the FRAME.  It runs on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, 16x the f32-MFMA rate) at fp32 accuracy: every fp32 operand is
// split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits, by truncation:
// x = h + m + l with no rounding at all, and bf16 keeps the fp32 exponent range, so no scaling
// and no overflow cases), and the six products of total order <= 2,
//     phi.z = ph.zh + ph.zm + pm.zh + ph.zl + pl.zh + pm.zm        (dropped terms <= 2^-24)
// are accumulated in fp32 by the matrix core.  Every bf16 x bf16 product is exact in fp32.
//
// Operands.  A = phi^T (row i = leaf, k = frame), B = Z (k = frame, col j = dim): lane
// (i = lane & 31, h = lane >> 5) holds for 8 consecutive frames 8h .. 8h+7 of a 16-frame
// k-step the value of ITS column (leaf or dim), so every load instruction of the wave reads
// two whole 128-byte row segments straight from HBM -- no LDS staging, no transpose -- and
// both operands stream through registers with two k-steps in flight per wave.
// Partial sums are fp32 over at most kMaxChunk frames per wave; everything above that is
// accumulated in fp64 (LDS across the waves of a workgroup in fixed order, then a fixed-order
// slab reduction), so results are bit-reproducible run to run.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "mstep_common.h"

namespace bnpvi {

template <int TL, int TD, bool WANT_Q>
__global__ __launch_bounds__(kStatWaves * 64, 2) void leaf_stats_kernel(
    const float* __restrict__ Z, int ldz, const float* __restrict__ phi, int ldphi,
    const int32_t* __restrict__ vid, int64_t N, int D, int Dp, int L, int64_t chunk,
    double* __restrict__ slab, int W, double* __restrict__ mass, int gx, int gy, int gz) {
  constexpr int NQ = WANT_Q ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* acc_lds = reinterpret_cast<double*>(smem_raw);          // [TL*TD*NQ][16][64]
  double* s0_lds = acc_lds + TL * TD * NQ * 16 * 64;              // [TL][32]

  // Workgroups b and b + 8 share an XCD (round-robin dispatch): the (leaf block, dim block)
  // pairs of one frame chunk are walked back to back on ONE XCD, so the Z rows (and, with
  // several dim blocks, the phi rows) they share are L2 hits after the first reader.
  const unsigned inner = (unsigned)(gy * gz);
  const unsigned seq = blockIdx.x >> 3;
  const int cx = (int)((seq / inner) * 8 + (blockIdx.x & 7));
  if (cx >= gx) return;
  const int by = (int)(seq % inner) % gy;
  const int bz = (int)(seq % inner) / gy;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31;
  const int h = lane >> 5;
  const int ltile0 = by * TL;
  const int dtile0 = bz * TD;
  const bool lead_dim = (bz == 0);
  const bool do_mass = lead_dim && (mass != nullptr);

  const int64_t c = (int64_t)cx * kStatWaves + wave;
  const int64_t n0 = c * chunk < N ? c * chunk : N;
  const int64_t n1 = (n0 + chunk < N) ? n0 + chunk : N;

  f32x16 acc[TL][TD][NQ];
#pragma unroll
  for (int a = 0; a < TL; ++a)
#pragma unroll
    for (int b = 0; b < TD; ++b)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][q][r] = 0.f;

  // block-uniform tile bases; a tile past the padded width aliases the block's first tile (its
  // loads stay in bounds, its products are skipped and its sums are never written out)
  int lcol[TL], dcol[TD];
  bool lok[TL], dok[TD];
  const float* pA[TL];
  const float* pB[TD];
#pragma unroll
  for (int a = 0; a < TL; ++a) {
    lok[a] = (ltile0 + a) * kTile < ldphi;
    lcol[a] = (ltile0 + a) * kTile + i;
    pA[a] = phi + (lok[a] ? (ltile0 + a) * kTile : ltile0 * kTile);
  }
#pragma unroll
  for (int b = 0; b < TD; ++b) {
    dok[b] = (dtile0 + b) * kTile < Dp;
    dcol[b] = (dtile0 + b) * kTile + i;
    pB[b] = Z + (dok[b] ? (dtile0 + b) * kTile : dtile0 * kTile);
  }
  // per-lane element offsets inside a k-step (32-bit; the k-step base is wave-uniform)
  unsigned offA[8], offB[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    offA[e] = (unsigned)((8 * h + e) * ldphi + i);
    offB[e] = (unsigned)((8 * h + e) * ldz + i);
  }

  float s0[TL], msum[TL];
  S0Check s0chk[TL];
#pragma unroll
  for (int a = 0; a < TL; ++a) s0[a] = msum[a] = 0.f;
  int my_vid = -1;

  auto flush_mass = [&]() {
    if (my_vid >= 0) {
#pragma unroll
      for (int a = 0; a < TL; ++a)
        if (lcol[a] < L && msum[a] != 0.f) atomicAdd(&mass[(int64_t)my_vid * L + lcol[a]], (double)msum[a]);
    }
#pragma unroll
    for (int a = 0; a < TL; ++a) msum[a] = 0.f;
  };

  // loads of one k-step (16 frames from f0); TAIL: rows >= n1 read as zero
  auto load_step = [&](RawStep<TL, TD>& r, int64_t f0, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = !TAIL || f0 + 8 * h + e < n1;
      const unsigned oa = ok ? offA[e] : (unsigned)i;      // row f0 itself is always < n1
      const unsigned ob = ok ? offB[e] : (unsigned)i;
#pragma unroll
      for (int a = 0; a < TL; ++a) {
        const float x = (pA[a] + f0 * ldphi)[oa];
        r.a[a][e] = ok ? x : 0.f;
      }
#pragma unroll
      for (int b = 0; b < TD; ++b) {
        const float x = (pB[b] + f0 * ldz)[ob];
        r.b[b][e] = ok ? x : 0.f;
      }
    }
  };

  auto load_b = [&](RawStep<TL, TD>& r, int64_t f0) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int b = 0; b < TD; ++b) r.b[b][e] = (pB[b] + f0 * ldz)[offB[e]];
  };
  auto load_a = [&](RawStep<TL, TD>& r, int64_t f0, int a) {
#pragma unroll
    for (int e = 0; e < 8; ++e) r.a[a][e] = (pA[a] + f0 * ldphi)[offA[e]];
  };

  // everything a k-step does with its registers except the loads
  auto consume = [&](const RawStep<TL, TD>& r, int64_t f0, auto tail_tag, Frag (&bh)[TD][NQ],
                     Frag (&bm)[TD][NQ], Frag (&bl)[TD][NQ]) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    // videos of the lane's 8 frames: the 16 ids of a full k-step sit at a wave-uniform address
    int ve[8];
    if (do_mass) {
      const int32_t* vb = vid + f0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (TAIL)
          ve[e] = f0 + 8 * h + e < n1 ? vb[8 * h + e] : -2;
        else
          ve[e] = h ? vb[8 + e] : vb[e];
      }
    }
#pragma unroll
    for (int b = 0; b < TD; ++b) {
      split3(r.b[b], bh[b][0], bm[b][0], bl[b][0]);
      if (WANT_Q) {
        float sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sq[e] = r.b[b][e] * r.b[b][e];
        split3(sq, bh[b][NQ - 1], bm[b][NQ - 1], bl[b][NQ - 1]);
      }
    }
    if (lead_dim) {
      // S0: fixed tree sum of the lane's 8 frames, accumulated in straight-line code AHEAD of the
      // divergent video-boundary path (mstep_common.h, s0_accumulate: why)
#pragma unroll
      for (int a = 0; a < TL; ++a) {
        const float t = ((r.a[a][0] + r.a[a][1]) + (r.a[a][2] + r.a[a][3])) +
                        ((r.a[a][4] + r.a[a][5]) + (r.a[a][6] + r.a[a][7]));
        s0_accumulate(s0[a], t);
        s0chk[a].add(t);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (do_mass) {
          if (ve[e] != my_vid && ve[e] != -2) {   // video boundary: frames of a video are contiguous
            flush_mass();
            my_vid = ve[e];
          }
        }
#pragma unroll
        for (int a = 0; a < TL; ++a) msum[a] += r.a[a][e];
      }
    }
  };

  auto mfma_step = [&](const RawStep<TL, TD>& r, const Frag (&bh)[TD][NQ], const Frag (&bm)[TD][NQ],
                       const Frag (&bl)[TD][NQ]) {
#pragma unroll
    for (int a = 0; a < TL; ++a) {
      Frag ah, am, al;
      split3(r.a[a], ah, am, al);
#pragma unroll
      for (int b = 0; b < TD; ++b)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (lok[a] && dok[b]) mac6(ah, am, al, bh[b][q], bm[b][q], bl[b][q], acc[a][b][q]);
    }
  };

  // One k-step in registers per wave; each operand is re-loaded with the NEXT k-step's rows as
  // soon as it has been converted, so those loads are in flight during this step's MFMAs.
  const int64_t nfull = (n1 - n0) / kKStep;
  RawStep<TL, TD> r0;
  if (nfull > 0) load_step(r0, n0, std::false_type{});
  for (int64_t ks = 0; ks < nfull; ++ks) {
    const int64_t f0 = n0 + ks * kKStep;
    const bool more = ks + 1 < nfull;
    Frag bh[TD][NQ], bm[TD][NQ], bl[TD][NQ];
    consume(r0, f0, std::false_type{}, bh, bm, bl);
    if (more) load_b(r0, f0 + kKStep);
#pragma unroll
    for (int a = 0; a < TL; ++a) {
      Frag ah, am, al;
      split3(r0.a[a], ah, am, al);
      if (more) load_a(r0, f0 + kKStep, a);
#pragma unroll
      for (int b = 0; b < TD; ++b)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (lok[a] && dok[b]) mac6(ah, am, al, bh[b][q], bm[b][q], bl[b][q], acc[a][b][q]);
    }
  }
  if (n0 + nfull * kKStep < n1) {        // ragged last k-step of the data set
    load_step(r0, n0 + nfull * kKStep, std::true_type{});
    Frag bh[TD][NQ], bm[TD][NQ], bl[TD][NQ];
    consume(r0, n0 + nfull * kKStep, std::true_type{}, bh, bm, bl);
    mfma_step(r0, bh, bm, bl);
  }
  if (do_mass) flush_mass();

  // ---- combine the workgroup's waves in fp64 (wave order fixed => deterministic) ----
  for (int w = 0; w < kStatWaves; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < TL; ++a)
#pragma unroll
        for (int b = 0; b < TD; ++b)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              double* dst = &acc_lds[(((a * TD + b) * NQ + q) * 16 + r) * 64 + lane];
              const double v = (double)acc[a][b][q][r];
              *dst = (w == 0) ? v : (*dst + v);
            }
      if (lead_dim) {
#pragma unroll
        for (int a = 0; a < TL; ++a) {
          s0chk[a].verify(s0[a], "leaf_stats", a);
          float t = s0[a] + __shfl_xor(s0[a], 32);
          if (h == 0) {
            double* dst = &s0_lds[a * 32 + i];
            *dst = (w == 0) ? (double)t : (*dst + (double)t);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- write this workgroup's slab: slab[cx][leaf][W] ----
  double* out = slab + (int64_t)cx * L * W;
  for (int e = threadIdx.x; e < TL * TD * NQ * 16 * 64; e += blockDim.x) {
    const int ln = e & 63;
    const int r = (e >> 6) & 15;
    const int rest = e >> 10;
    const int q = rest % NQ;
    const int b = (rest / NQ) % TD;
    const int a = rest / (NQ * TD);
    const int leaf = (ltile0 + a) * kTile + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    const int dim = (dtile0 + b) * kTile + (ln & 31);
    if (leaf < L && dim < D) out[(int64_t)leaf * W + 1 + q * D + dim] = acc_lds[e];
  }
  if (lead_dim) {
    for (int e = threadIdx.x; e < TL * 32; e += blockDim.x) {
      const int leaf = (ltile0 + (e >> 5)) * kTile + (e & 31);
      if (leaf < L) out[(int64_t)leaf * W] = s0_lds[e];
    }
  }
}

template <int TL, int TD, bool Q>
static int run_stats(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                     const int32_t* vid, int64_t N, int D, int Dp, int L, int W, double* stats,
                     double* mass, hipStream_t s) {
  const int tilesL = ldphi / kTile, tilesD = (Dp + kTile - 1) / kTile;
  const int gy = (tilesL + TL - 1) / TL, gz = (tilesD + TD - 1) / TD;
  int64_t waves = (int64_t)ctx->num_cu * 8 / ((int64_t)gy * gz);      // 2 waves per SIMD resident
  if (waves < kStatWaves) waves = kStatWaves;
  int64_t chunk = ceil_div(N, waves);
  chunk = (chunk + kKStep - 1) / kKStep * kKStep;
  if (chunk < 4 * kKStep) chunk = 4 * kKStep;
  if (chunk > kMaxChunk) chunk = kMaxChunk;
  int64_t gx = ceil_div(N, chunk * kStatWaves);
  const int64_t cap = slab_cap(ctx, (size_t)L * W * sizeof(double));
  if (gx > cap && cap >= 1) {          // bounded workspace: fewer workgroups with longer chunks
    chunk = (ceil_div(N, cap * kStatWaves) + kKStep - 1) / kKStep * kKStep;
    if (chunk <= kMaxChunk) gx = ceil_div(N, chunk * kStatWaves);
  }
  const int64_t blocks = ceil_div(gx, 8) * 8 * gy * gz;
  BNPVI_REQUIRE(gx >= 1 && blocks < (int64_t(1) << 31), "leaf_stats: grid of %lld workgroups",
                (long long)blocks);
  const size_t slab_bytes = (size_t)gx * L * W * sizeof(double);
  double* slab_ws = nullptr;
  int rc = slab_workspace(ctx, slab_bytes, &slab_ws);
  if (rc) return rc;
  constexpr int NQ = Q ? 2 : 1;
  const size_t lds = (size_t)(TL * TD * NQ * 16 * 64 + TL * 32) * sizeof(double);
  auto kern = leaf_stats_kernel<TL, TD, Q>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const bool timed = ctx->profiling && ctx->timer[BNPVI_PROF_MSTEP].begin(s);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kStatWaves * 64), lds, s, Z, ldz, phi, ldphi, vid,
                     N, D, Dp, L, chunk, slab_ws, W, mass, (int)gx, gy, gz);
  if (timed) ctx->timer[BNPVI_PROF_MSTEP].end(s);
  BNPVI_LAUNCH_CHECK();
  const int64_t elems = (int64_t)L * W;
  return launch_slab_reduce(slab_ws, (int)gx, elems, stats, s);
}

int launch_leaf_stats(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                      const int32_t* vid, int64_t N, int D, int L, int V, int want_q,
                      double* stats, double* mass, hipStream_t s) {
  const int Dp = bnpvi_padded_dim(D);
  const int W = 1 + D * (want_q ? 2 : 1);
  // the mass rows are accumulated with atomics: they must start at zero (bnpvi_params_prior leaves
  // the buffer it consumed zeroed, so the steady-state iteration needs no memset)
  const bool known_zero = ctx->mass_clean == mass && ctx->mass_clean_elems == (size_t)V * L &&
                          ctx->mass_clean_stream == s;
  if (mass && !known_zero) BNPVI_HIP(hipMemsetAsync(mass, 0, (size_t)V * L * sizeof(double), s));
  if (mass) ctx->mass_clean = nullptr;
  if (N == 0) {
    BNPVI_HIP(hipMemsetAsync(stats, 0, (size_t)L * W * sizeof(double), s));
    return BNPVI_OK;
  }
  if (!want_q) {
    // default path: LDS-DMA ring (mstep_dma.hip); BNPVI_MSTEP_PATH=reg | dma force the register-streamed / per-wave ring kernels
    const char* force = dev_getenv("BNPVI_MSTEP_PATH");
    if (!(force && strcmp(force, "reg") == 0)) {
      if (!(force && strcmp(force, "dma") == 0)) {
        // large trees: workgroup-wide tiles (mstep_wide.hip)
        const int rc = launch_leaf_stats_wide(ctx, Z, ldz, phi, ldphi, vid, N, D, L, V, stats, mass, s);
        if (rc != BNPVI_EUNSUPPORTED) return rc;
      }
      const int rc = launch_leaf_stats_dma(ctx, Z, ldz, phi, ldphi, vid, N, D, L, V, stats, mass, s);
      if (rc != BNPVI_EUNSUPPORTED) return rc;
    }
  }
  const int tilesL = ldphi / kTile, tilesD = (Dp + kTile - 1) / kTile;
  const bool td2 = tilesD >= 2;
#define RUN(TL, TD, Q) \
  return run_stats<TL, TD, Q>(ctx, Z, ldz, phi, ldphi, vid, N, D, Dp, L, W, stats, mass, s)
  if (want_q) {
    // (2,2) with both S1 and Q accumulators does not fit the register file
    if (td2) RUN(1, 2, true);
    else if (tilesL >= 2) RUN(2, 1, true);
    else RUN(1, 1, true);
  } else {
    if (tilesL >= 2) { if (td2) RUN(2, 2, false); else RUN(2, 1, false); }
    else             { if (td2) RUN(1, 2, false); else RUN(1, 1, false); }
  }
#undef RUN
}

}  // namespace bnpvi
