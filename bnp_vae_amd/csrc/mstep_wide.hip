// M-step data pass for LARGE trees (gfx950): same statistics and arithmetic as mstep_dma.hip
// (exact bf16x3 split, six products on v_mfma_f32_32x32x16_bf16, fp32 partial sums over at most
// 8192 frames, fp64 slabs), with the k-step shared by a whole workgroup.
//
// With many (leaf block, dim block) pairs the per-wave ring of mstep_dma.hip re-reads every Z
// block once per leaf block and every phi block once per dim block through L2 (config 4: 48 GB
// of L2 traffic for 4 GB of HBM traffic -- L2-bandwidth bound).  Here a workgroup of 8 waves
// owns a tile of (64 LBW leaves) x (64 DBW dims) over one chunk of frames; the 16 frames of a
// k-step -- phi block [16][64 LBW], Z block [16][64 DBW], 16 video ids -- are brought into ONE
// LDS slot by all waves together (LDS-DMA, 1 KiB per instruction, whole row segments) and every
// wave takes its 64 x 64 sub-tile from it.  L2 traffic drops by LBW for Z and DBW for phi.
//
// Ring of 3 slots, ONE workgroup barrier per k-step:
//     wait until this wave's pieces of k-step ks have landed (counted vmcnt)
//     barrier            -> everybody's pieces of ks are in LDS, everybody is done reading ks-1
//     issue k-step ks+2  -> into the slot of ks-1
//     convert + MFMA on ks
// The waves of a workgroup work on the SAME frames and DIFFERENT outputs, so there is no
// cross-wave combine: every wave writes its accumulators straight into the chunk's fp64 slab.
#include "mstep_common.h"

namespace bnpvi {

constexpr int kWideWaves = 8;
constexpr int kWideRing = 3;

template <int LBW, int DBW>
struct WideSlot {
  static constexpr int kAL = 64 * LBW;                  // leaves per workgroup
  static constexpr int kBD = 64 * DBW;                  // dims per workgroup
  static constexpr int kA = kKStep * kAL;               // floats of the phi block
  static constexpr int kB = kKStep * kBD;               // floats of the Z block
  static constexpr int kFloats = kA + kB + 64;          // + the video ids (16 used)
  static constexpr int kPA = kA / 256;                  // 1 KiB pieces of the phi block
  static constexpr int kPT = (kA + kB) / 256;           // ... of both blocks
  static constexpr int kPW = kPT / kWideWaves;          // pieces per wave
  static_assert(LBW * DBW == kWideWaves, "8 waves per workgroup");
  static_assert(kPT % kWideWaves == 0, "pieces must divide evenly over the waves");
};

__device__ __forceinline__ void wide_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int LBW, int DBW>
__global__ __launch_bounds__(kWideWaves * 64) void leaf_stats_wide_kernel(
    const float* __restrict__ Z, int ldz, const float* __restrict__ phi, int ldphi,
    const int32_t* __restrict__ vid, int64_t N, int D, int L, int64_t chunk, double* __restrict__ slab,
    int W, double* __restrict__ mass, int gx, int gy, int gz) {
  using Slot = WideSlot<LBW, DBW>;
  constexpr int R = kWideRing;
  constexpr int AL = Slot::kAL, BD = Slot::kBD, PW = Slot::kPW;
  constexpr int TL = 2, TD = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* ring = reinterpret_cast<float*>(smem_raw);                // [R][Slot::kFloats]

  const unsigned inner = (unsigned)(gy * gz);
  const unsigned seq = blockIdx.x >> 3;
  const int cx = (int)((seq / inner) * 8 + (blockIdx.x & 7));
  if (cx >= gx) return;                                            // whole workgroup
  const int by = (int)(seq % inner) % gy;
  const int bz = (int)(seq % inner) / gy;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wl = wave % LBW, wd = wave / LBW;                      // sub-tile of this wave
  const int i = lane & 31;
  const int h = lane >> 5;
  const bool lead = (bz == 0) && (wd == 0);                        // S0 / masses of its leaves
  const bool do_mass = lead && (mass != nullptr);

  const int64_t n0 = (int64_t)cx * chunk < N ? (int64_t)cx * chunk : N;
  const int64_t n1 = (n0 + chunk < N) ? n0 + chunk : N;
  const int64_t nsteps = (n1 - n0 + kKStep - 1) / kKStep;          // same for all waves of the workgroup

  // ---- this wave's DMA pieces: piece p = wave + 8 q covers floats [256 p, 256 p + 256) of the slot
  const float* src_base[PW];
  int src_ld[PW];
  unsigned src_col[PW];
  int src_row[PW];
#pragma unroll
  for (int q = 0; q < PW; ++q) {
    const int p = wave + kWideWaves * q;
    if (p < Slot::kPA) {
      const int f = 256 * p + 4 * lane;
      int col = by * AL + f % AL;
      col = col + 4 <= ldphi ? col : ldphi - 4;                    // past the padded leaves: any valid column
      src_base[q] = phi;
      src_ld[q] = ldphi;
      src_col[q] = (unsigned)col;
      src_row[q] = f / AL;
    } else {
      const int f = 256 * (p - Slot::kPA) + 4 * lane;
      src_base[q] = Z;
      src_ld[q] = ldz;
      src_col[q] = (unsigned)(bz * BD + f % BD);                   // Dp is a multiple of BD (checked on the host)
      src_row[q] = f / BD;
    }
  }
  auto issue = [&](int64_t ks) {
    float* slot = ring + (size_t)(ks % R) * Slot::kFloats;
    const int64_t f0 = n0 + ks * kKStep;
    const bool whole = f0 + kKStep <= n1;                          // wave-uniform
#pragma unroll
    for (int q = 0; q < PW; ++q) {
      int64_t row = f0 + src_row[q];
      if (!whole) row = row < n1 ? row : n1 - 1;                   // ragged tail: re-read the last valid row
      dma16(src_base[q] + row * src_ld[q] + src_col[q], slot + 256 * (wave + kWideWaves * q));
    }
    int64_t vr = f0 + (lane & 15);
    vr = vr < n1 ? vr : n1 - 1;
    dma4(vid + vr, slot + Slot::kA + Slot::kB);                    // every wave: same 64 bytes, same place
  };
  constexpr int P = PW + 1;                                        // DMA instructions per wave per k-step

  f32x16 acc[TL][TD];
#pragma unroll
  for (int a = 0; a < TL; ++a)
#pragma unroll
    for (int b = 0; b < TD; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float s0[TL], msum[TL];
  int lcol[TL];
#pragma unroll
  for (int a = 0; a < TL; ++a) {
    s0[a] = msum[a] = 0.f;
    lcol[a] = by * AL + wl * 64 + 32 * a + i;
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

  if (nsteps > 0) issue(0);
  if (nsteps > 1) issue(1);

  for (int64_t ks = 0; ks < nsteps; ++ks) {
    if (ks + 1 < nsteps)
      wait_vmcnt<P>();          // only the pieces of ks+1 may still be in flight
    else
      wait_vmcnt<0>();
    wide_barrier();
    if (ks + 2 < nsteps) issue(ks + 2);

    const float* slot = ring + (size_t)(ks % R) * Slot::kFloats;
    const int64_t f0 = n0 + ks * kKStep;
    const bool ragged = f0 + kKStep > n1;

    Frag bh[TD], bm[TD], bl[TD];
#pragma unroll
    for (int b = 0; b < TD; ++b) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = slot[Slot::kA + (8 * h + e) * BD + wd * 64 + 32 * b + i];
      split3(x, bh[b], bm[b], bl[b]);
    }
    float xa[TL][8];
#pragma unroll
    for (int a = 0; a < TL; ++a)
#pragma unroll
      for (int e = 0; e < 8; ++e) xa[a][e] = slot[(8 * h + e) * AL + wl * 64 + 32 * a + i];
    if (ragged) {
#pragma unroll
      for (int a = 0; a < TL; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (f0 + 8 * h + e >= n1) xa[a][e] = 0.f;
    }
    if (lead) {
      const int* vslot = reinterpret_cast<const int*>(slot + Slot::kA + Slot::kB);
      bool same_video = !do_mass;
      if (do_mass && !ragged) same_video = __ballot(vslot[lane & 15] != my_vid) == 0ull;
      float t[TL];
#pragma unroll
      for (int a = 0; a < TL; ++a) {
        t[a] = ((xa[a][0] + xa[a][1]) + (xa[a][2] + xa[a][3])) + ((xa[a][4] + xa[a][5]) + (xa[a][6] + xa[a][7]));
        s0[a] += t[a];
        asm volatile("" : "+v"(s0[a]));        // see mstep_dma.hip: keep the S0 update ahead of the divergent path
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

  // ---- every wave owns its outputs: fp64 straight into the chunk's slab
  double* out = slab + (int64_t)cx * L * W;
#pragma unroll
  for (int a = 0; a < TL; ++a)
#pragma unroll
    for (int b = 0; b < TD; ++b) {
      const int dim = bz * BD + wd * 64 + 32 * b + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int leaf = by * AL + wl * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (leaf < L && dim < D) out[(int64_t)leaf * W + 1 + dim] = (double)acc[a][b][r];
      }
    }
  if (lead) {
#pragma unroll
    for (int a = 0; a < TL; ++a) {
      const float t = s0[a] + __shfl_xor(s0[a], 32);
      if (h == 0 && lcol[a] < L) out[(int64_t)lcol[a] * W] = (double)t;
    }
  }
}

template <int LBW, int DBW>
static int run_wide(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi, const int32_t* vid,
                    int64_t N, int D, int L, int W, double* stats, double* mass, hipStream_t s) {
  using Slot = WideSlot<LBW, DBW>;
  const int gy = (int)ceil_div(ldphi, Slot::kAL), gz = ldz / Slot::kBD;
  const size_t lds = (size_t)kWideRing * Slot::kFloats * sizeof(float);
  // enough workgroups for ~4 per CU, chunks of at most kMaxChunk frames (fp32 partial sums)
  // one workgroup per CU at a time (LDS ring + registers): a whole number of rounds, about 4
  int64_t gx = (int64_t)4 * ctx->num_cu / ((int64_t)gy * gz);
  if (gx < 1) gx = 1;
  int64_t chunk = ceil_div(N, gx);
  chunk = (chunk + kKStep - 1) / kKStep * kKStep;
  if (chunk < 8 * kKStep) chunk = 8 * kKStep;
  if (chunk > kMaxChunk) chunk = kMaxChunk;
  gx = ceil_div(N, chunk);
  const int64_t blocks = ceil_div(gx, 8) * 8 * gy * gz;
  BNPVI_REQUIRE(blocks < (int64_t(1) << 31), "leaf_stats: grid of %lld workgroups", (long long)blocks);
  int rc = ctx->slabs.ensure((size_t)gx * L * W * sizeof(double));
  if (rc) return rc;
  auto kern = leaf_stats_wide_kernel<LBW, DBW>;
  BNPVI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const bool timed = ctx->profiling && ctx->timer[BNPVI_PROF_MSTEP].begin(s);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kWideWaves * 64), lds, s, Z, ldz, phi, ldphi, vid, N, D, L,
                     chunk, (double*)ctx->slabs.ptr, W, mass, (int)gx, gy, gz);
  if (timed) ctx->timer[BNPVI_PROF_MSTEP].end(s);
  BNPVI_LAUNCH_CHECK();
  return launch_slab_reduce((const double*)ctx->slabs.ptr, (int)gx, (int64_t)L * W, stats, s);
}

int launch_leaf_stats_wide(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                           const int32_t* vid, int64_t N, int D, int L, int V, double* stats, double* mass,
                           hipStream_t s) {
  (void)V;
  const int W = 1 + D;
  // workgroup-wide tiles: worth it from ~1000 leaves on, need D padded to >= 128
  if (N < kKStep || (ldphi % 4) || (ldz % 128) || vid == nullptr || ldphi < 4 * 256) return BNPVI_EUNSUPPORTED;
  // full-width dim tiles when D is padded to 256 (phi is then read exactly once; measured 4 % faster
  // than 256 x 128 at 512k frames), else 256 leaves x 128 dims
  if (ldz % 256 == 0) return run_wide<2, 4>(ctx, Z, ldz, phi, ldphi, vid, N, D, L, W, stats, mass, s);
  return run_wide<4, 2>(ctx, Z, ldz, phi, ldphi, vid, N, D, L, W, stats, mass, s);
}

}  // namespace bnpvi
