// M-step data pass, LDS-DMA variant (gfx950): same statistics and the same arithmetic as
// mstep.hip (exact bf16x3 split, six products, v_mfma_f32_32x32x16_bf16, fp32 partial sums,
// fp64 above a chunk) -- what changes is how phi and Z reach the lanes.
//
// Each wave owns a RING of R slots in LDS.  A slot holds one k-step (16 frames): the phi block
// [16][32 TL] and the Z block [16][32 TD] of the wave's tile, row-major as in memory, plus the
// 16 video ids.  The wave fills slots with `global_load_lds_dwordx4` (LDS-DMA: 1 KiB per
// instruction, whole contiguous row segments, NO destination registers), R-1 k-steps ahead of
// the one it is converting, and only waits for the oldest outstanding slot (counted vmcnt).
// The MFMA operand layout (8 consecutive frames of ONE column per lane) is produced by the LDS
// reads: 32 lanes read 32 consecutive floats of a row -- conflict-free.  Nothing inside the
// loop is a register-returning vector load, so no s_waitcnt ever drains the DMA queue.
// HBM bytes in flight per CU: waves x (R-1) x slot bytes, independent of the register budget.
#include <stdlib.h>

#include "mstep_common.h"

namespace bnpvi {

constexpr int kDmaWaves = 4;

template <int TL, int TD>
struct DmaSlot {
  static constexpr int kA = kKStep * TL * 32;          // floats of the phi block
  static constexpr int kB = kKStep * TD * 32;          // floats of the Z block
  static constexpr int kFloats = kA + kB + 64;         // + 64 dwords for the video ids (16 used)
  static constexpr int kPieces = 2 * TL + 2 * TD + 1;  // DMA instructions per k-step
};

template <int TL, int TD, int R>
__global__ __launch_bounds__(kDmaWaves * 64, 1) void leaf_stats_dma_kernel(
    const float* __restrict__ Z, int ldz, const float* __restrict__ phi, int ldphi,
    const int32_t* __restrict__ vid, int64_t N, int D, int L, int64_t chunk, double* __restrict__ slab,
    int W, double* __restrict__ mass, int gx, int gy, int gz, int order, int rev) {
  using Slot = DmaSlot<TL, TD>;
  constexpr int P = Slot::kPieces;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* ring_all = reinterpret_cast<float*>(smem_raw);            // [waves][R][Slot::kFloats]

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
  float* ring = ring_all + (size_t)wave * R * Slot::kFloats;

  // Frame order.  order < 0: the wave owns the contiguous frames [c*chunk, (c+1)*chunk).
  // order = b >= 0: k-steps are dealt to the waves round-robin in blocks of 2^b, so the grid as a
  // whole sweeps the matrices in address order -- BACKWARDS when `rev` (the E-step sweeps
  // forwards: what it wrote last, and what this kernel reads last, is what the memory-side
  // cache still holds for the next kernel).  Every wave then runs chunk/16 k-steps; those past
  // the end of the data take the ragged path with every row masked.
  const int64_t c = (int64_t)cx * kDmaWaves + wave;
  const bool ilv = order >= 0;
  const int64_t nwaves = (int64_t)gx * kDmaWaves;
  const int64_t nblocks = ilv ? ((chunk / kKStep) >> order) * nwaves : 0;
  const int64_t n0 = c * chunk < N ? c * chunk : N;
  const int64_t n1 = ilv ? N : ((n0 + chunk < N) ? n0 + chunk : N);
  const int64_t nsteps = ilv ? chunk / kKStep : (n1 - n0 + kKStep - 1) / kKStep;   // the last one may be ragged
  auto first_frame = [&](int64_t ks) -> int64_t {                // wave-uniform
    if (!ilv) return n0 + ks * kKStep;
    int64_t blk = (ks >> order) * nwaves + c;
    if (rev) blk = nblocks - 1 - blk;
    return ((blk << order) + (ks & ((int64_t(1) << order) - 1))) * kKStep;
  };

  f32x16 acc[TL][TD];
#pragma unroll
  for (int a = 0; a < TL; ++a)
#pragma unroll
    for (int b = 0; b < TD; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // DMA lane geometry: a piece is 64 lanes x 16 B = 1 KiB = (64 / chunks-per-row) whole rows
  constexpr int CA = 8 * TL, CB = 8 * TD;
  constexpr int rowsA = 64 / CA, rowsB = 64 / CB;
  const int rA = lane / CA, cA = lane % CA;
  const int rB = lane / CB, cB = lane % CB;
  // fixed 32-bit lane offsets; everything that moves with the k-step is wave-uniform
  const unsigned offA = (unsigned)(rA * ldphi + 4 * cA);
  const unsigned offB = (unsigned)(rB * ldz + 4 * cB);
  const float* phi_t = phi + (int64_t)ltile0 * kTile;
  const float* z_t = Z + (int64_t)dtile0 * kTile;

  auto issue = [&](int64_t ks) {
    float* slot = ring + (size_t)(ks % R) * Slot::kFloats;
    const int64_t f0 = first_frame(ks);
    if (f0 + kKStep <= n1) {
      const float* ga = phi_t + f0 * ldphi;
      const float* gb = z_t + f0 * ldz;
#pragma unroll
      for (int q = 0; q < 2 * TL; ++q) dma16_nt(ga + (int64_t)(q * rowsA) * ldphi + offA, slot + q * 256);
#pragma unroll
      for (int q = 0; q < 2 * TD; ++q) dma16(gb + (int64_t)(q * rowsB) * ldz + offB, slot + Slot::kA + q * 256);
      dma4(vid + f0 + (lane & 15), slot + Slot::kA + Slot::kB);
    } else {
      // ragged last k-step of the data set: rows past n1 re-read the last valid row (their phi
      // is zeroed by the consumer, so they add nothing)
#pragma unroll
      for (int q = 0; q < 2 * TL; ++q) {
        int64_t row = f0 + q * rowsA + rA;
        row = row < n1 ? row : n1 - 1;
        dma16(phi_t + row * ldphi + 4 * cA, slot + q * 256);
      }
#pragma unroll
      for (int q = 0; q < 2 * TD; ++q) {
        int64_t row = f0 + q * rowsB + rB;
        row = row < n1 ? row : n1 - 1;
        dma16(z_t + row * ldz + 4 * cB, slot + Slot::kA + q * 256);
      }
      int64_t vr = f0 + (lane & 15);
      vr = vr < n1 ? vr : n1 - 1;
      dma4(vid + vr, slot + Slot::kA + Slot::kB);
    }
  };

  float s0[TL], msum[TL];
  S0Check s0chk[TL];
  int lcol[TL];
#pragma unroll
  for (int a = 0; a < TL; ++a) {
    s0[a] = msum[a] = 0.f;
    lcol[a] = (ltile0 + a) * kTile + i;
  }
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

  // prologue: R-1 k-steps in flight
#pragma unroll
  for (int k = 0; k < R - 1; ++k)
    if (k < nsteps) issue(k);
#ifdef BNPVI_MABLATE_NOLOAD
  // ablation build: every slot holds valid rows (video ids included), the loop loads nothing
  if (R - 1 < nsteps) issue(R - 1);
#endif

  for (int64_t ks = 0; ks < nsteps; ++ks) {
    // Refill first: the slot k-step ks+R-1 goes into was consumed a whole iteration ago (all its
    // LDS reads fed VALU work that has issued since).  Then the slot of k-step ks has landed
    // once at most the pieces of the R-1 younger k-steps are pending (in-order completion).
#ifdef BNPVI_MABLATE_NOLOAD
    wait_vmcnt<0>();
#else
    if (ks + R - 1 < nsteps) {
      // every LDS read of the slot about to be refilled has returned (they all fed VALU work of the
      // previous iteration); stated explicitly so the refill can never be scheduled above one
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(ks + R - 1);
      wait_vmcnt<(R - 1) * P>();
    } else {
      wait_vmcnt<0>();
    }
#endif
    const float* slot = ring + (size_t)(ks % R) * Slot::kFloats;
#ifdef BNPVI_MABLATE_NOCOMPUTE
    s0[0] += slot[lane] + slot[Slot::kA + lane];
    continue;
#endif
    const int64_t f0 = first_frame(ks);
    const bool ragged = f0 + kKStep > n1;                       // wave-uniform, last k-step only

    Frag bh[TD], bm[TD], bl[TD];
#pragma unroll
    for (int b = 0; b < TD; ++b) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = slot[Slot::kA + (8 * h + e) * (TD * 32) + 32 * b + i];
      split3(x, bh[b], bm[b], bl[b]);
    }
    float xa[TL][8];
#pragma unroll
    for (int a = 0; a < TL; ++a)
#pragma unroll
      for (int e = 0; e < 8; ++e) xa[a][e] = slot[(8 * h + e) * (TL * 32) + 32 * a + i];
    if (ragged) {
#pragma unroll
      for (int a = 0; a < TL; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (f0 + 8 * h + e >= n1) xa[a][e] = 0.f;
    }
    if (lead_dim) {
      const int* vslot = reinterpret_cast<const int*>(slot + Slot::kA + Slot::kB);
      // Common case (15 of 16 k-steps with 256-frame videos): all 16 frames belong to the video
      // every lane is already accumulating -- one LDS read and one wave-uniform branch instead
      // of a read-compare-branch chain per frame.
      bool same_video = !do_mass;
      if (do_mass && !ragged) same_video = __ballot(vslot[lane & 15] != my_vid) == 0ull;
      // S0 takes the same fixed tree sum on both paths (its rounding must not depend on whether
      // masses were asked for)
      float t[TL];
#pragma unroll
      for (int a = 0; a < TL; ++a) {
        t[a] = ((xa[a][0] + xa[a][1]) + (xa[a][2] + xa[a][3])) + ((xa[a][4] + xa[a][5]) + (xa[a][6] + xa[a][7]));
        s0_accumulate(s0[a], t[a]);            // mstep_common.h: why this is not `s0[a] += t[a]`
        s0chk[a].add(t[a]);
      }
      if (__builtin_expect(same_video, 1)) {
#pragma unroll
        for (int a = 0; a < TL; ++a) msum[a] += t[a];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ve = vslot[8 * h + e];
          if (ve != my_vid && (!ragged || f0 + 8 * h + e < n1)) {   // video boundary
            flush_mass();
            my_vid = ve;
          }
#pragma unroll
          for (int a = 0; a < TL; ++a) msum[a] += xa[a][e];
        }
      }
    }
#pragma unroll
    for (int a = 0; a < TL; ++a) {
      Frag ah, am, al;
      split3(xa[a], ah, am, al);
#pragma unroll
      for (int b = 0; b < TD; ++b) mac6(ah, am, al, bh[b], bm[b], bl[b], acc[a][b]);
    }
  }
  if (do_mass) flush_mass();
  if (lead_dim)
#pragma unroll
    for (int a = 0; a < TL; ++a) s0chk[a].verify(s0[a], "leaf_stats_dma", a);

  // ---- combine the workgroup's waves in fp64 (wave order fixed => deterministic); the ring
  //      memory is free now
  __syncthreads();
  double* acc_lds = reinterpret_cast<double*>(smem_raw);          // [TL*TD][16][64]
  double* s0_lds = acc_lds + TL * TD * 16 * 64;                   // [TL][32]
  for (int w = 0; w < kDmaWaves; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < TL; ++a)
#pragma unroll
        for (int b = 0; b < TD; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            double* dst = &acc_lds[((a * TD + b) * 16 + r) * 64 + lane];
            const double v = (double)acc[a][b][r];
            *dst = (w == 0) ? v : (*dst + v);
          }
      if (lead_dim) {
#pragma unroll
        for (int a = 0; a < TL; ++a) {
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
  double* out = slab + (int64_t)cx * L * W;
  for (int e = threadIdx.x; e < TL * TD * 16 * 64; e += blockDim.x) {
    const int ln = e & 63;
    const int r = (e >> 6) & 15;
    const int rest = e >> 10;
    const int b = rest % TD;
    const int a = rest / TD;
    const int leaf = (ltile0 + a) * kTile + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    const int dim = (dtile0 + b) * kTile + (ln & 31);
    if (leaf < L && dim < D) out[(int64_t)leaf * W + 1 + dim] = acc_lds[e];
  }
  if (lead_dim) {
    for (int e = threadIdx.x; e < TL * 32; e += blockDim.x) {
      const int leaf = (ltile0 + (e >> 5)) * kTile + (e & 31);
      if (leaf < L) out[(int64_t)leaf * W] = s0_lds[e];
    }
  }
}

template <int TL, int TD, int R>
static int run_dma(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi, const int32_t* vid,
                   int64_t N, int D, int L, int W, int tilesL, int tilesD, double* stats, double* mass,
                   hipStream_t s) {
  using Slot = DmaSlot<TL, TD>;
  const int gy = tilesL / TL, gz = tilesD / TD;
  size_t lds = (size_t)kDmaWaves * R * Slot::kFloats * sizeof(float);
  const size_t comb = (size_t)(TL * TD * 16 * 64 + TL * 32) * sizeof(double);
  if (lds < comb) lds = comb;
  int per_cu = (int)((size_t)kMaxLdsBytes / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
  int64_t waves = (int64_t)ctx->num_cu * per_cu * kDmaWaves / ((int64_t)gy * gz);
  if (waves < kDmaWaves) waves = kDmaWaves;
  // Frame order.  When phi + Z are within a few times the 256 MB memory-side cache, round-robin
  // blocks of 8 k-steps (128 frames) swept BACKWARDS: the E-step before this kernel swept
  // forwards, so the rows read first are the ones it touched last (measured inside the iteration
  // at N=1M, L=D=64: 109 us against 122 us for contiguous chunks, 130 us for a forward sweep).
  // Much larger or much smaller inputs keep one contiguous chunk per wave (4.5 GB: 1.08 vs
  // 1.11 ms; 15 MB: 11.5 vs 13.4 us).  BNPVI_MSTEP_ORDER=<b>[r] | c overrides (development):
  // blocks of 2^b k-steps, r = backwards; c = contiguous chunks.
  static const int env_order = [] {
    const char* e = dev_getenv("BNPVI_MSTEP_ORDER");
    if (!e || !e[0]) return -2;
    return (e[0] >= '0' && e[0] <= '6') ? e[0] - '0' : -1;
  }();
  static const int env_rev = [] {
    const char* e = dev_getenv("BNPVI_MSTEP_ORDER");
    return (e && e[0] && e[1] == 'r') ? 1 : 0;
  }();
  const double bytes = (double)N * (ldphi + ldz) * sizeof(float);
  const bool cache_sized = bytes >= 64.0 * (1 << 20) && bytes <= 1024.0 * (1 << 20);
  const int order = env_order != -2 ? env_order : (cache_sized ? 3 : -1);
  const int rev = env_order != -2 ? env_rev : 1;
  const int64_t unit = order >= 0 ? (int64_t)kKStep << order : kKStep;
  int64_t chunk = ceil_div(N, waves);
  chunk = (chunk + unit - 1) / unit * unit;
  if (chunk < 4 * kKStep) chunk = (4 * kKStep + unit - 1) / unit * unit;
  if (chunk > kMaxChunk) chunk = kMaxChunk / unit * unit;
  int64_t gx = ceil_div(N, chunk * kDmaWaves);
  const int64_t cap = slab_cap(ctx, (size_t)L * W * sizeof(double));
  if (gx > cap && cap >= 1) {          // bounded workspace: fewer workgroups with longer chunks
    const int64_t c2 = (ceil_div(N, cap * kDmaWaves) + unit - 1) / unit * unit;
    if (c2 <= kMaxChunk) {
      chunk = c2;
      gx = ceil_div(N, chunk * kDmaWaves);
    }
  }
  const int64_t blocks = ceil_div(gx, 8) * 8 * gy * gz;
  BNPVI_REQUIRE(gx >= 1 && blocks < (int64_t(1) << 31), "leaf_stats: grid of %lld workgroups", (long long)blocks);
  double* slab_ws = nullptr;
  int rc = slab_workspace(ctx, (size_t)gx * L * W * sizeof(double), &slab_ws);
  if (rc) return rc;
  auto kern = leaf_stats_dma_kernel<TL, TD, R>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const bool timed = ctx->profiling && ctx->timer[BNPVI_PROF_MSTEP].begin(s);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kDmaWaves * 64), lds, s, Z, ldz, phi, ldphi, vid, N, D, L,
                     chunk, slab_ws, W, mass, (int)gx, gy, gz, order, rev);
  if (timed) ctx->timer[BNPVI_PROF_MSTEP].end(s);
  BNPVI_LAUNCH_CHECK();
  return launch_slab_reduce(slab_ws, (int)gx, (int64_t)L * W, stats, s);
}

#ifndef BNPVI_DMA_RING
#define BNPVI_DMA_RING 2
#endif

int launch_leaf_stats_dma(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                          const int32_t* vid, int64_t N, int D, int L, int V, double* stats, double* mass,
                          hipStream_t s) {
  (void)V;
  const int Dp = bnpvi_padded_dim(D);
  const int W = 1 + D;
  const int tilesL = ldphi / kTile, tilesD = Dp / kTile;
  // whole 16-byte chunks per lane: row strides must keep every row 16-byte aligned
  if (N < kKStep || (ldphi % 4) || (ldz % 4) || vid == nullptr) return BNPVI_EUNSUPPORTED;
  constexpr int R = BNPVI_DMA_RING;
  const bool l2 = tilesL % 2 == 0, d2 = tilesD % 2 == 0;
#define RUN(TL, TD) \
  return run_dma<TL, TD, R>(ctx, Z, ldz, phi, ldphi, vid, N, D, L, W, tilesL, tilesD, stats, mass, s)
  // four dim tiles per wave where the padded dims allow (D = 128, 256): phi is then read once per 128 dims instead
  // of once per 64
  // (2M frames, 64 x 64 tiles -> 64 x 128: D = 128 L = 64 / 121 / 256 370 -> 343, 634 -> 570, 1141 -> 1010 us;
  //  D = 256 L = 64 / 121 670 -> 615, 1186 -> 1033 us; a third ring slot on top: no change)
  const char* td4 = dev_getenv("BNPVI_DMA_TD4");                        // development switch: "0" = 64-dim tiles
  if (tilesD % 4 == 0 && !(td4 && td4[0] == '0')) {
    if (l2) RUN(2, 4);
    RUN(1, 4);
  }
  const char* tl4 = dev_getenv("BNPVI_DMA_TL4");                        // development switch: "0" = 64-leaf tiles
  if (tilesL % 4 == 0 && tilesD == 2 && !(tl4 && tl4[0] == '0')) RUN(4, 2);   // D = 64: Z read once per 128 leaves
  if (l2 && d2) RUN(2, 2);
  if (l2) RUN(2, 1);
  if (d2) RUN(1, 2);
  RUN(1, 1);
#undef RUN
}

}  // namespace bnpvi
