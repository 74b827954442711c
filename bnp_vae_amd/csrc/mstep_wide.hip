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
// Ring of 3 slots, ONE workgroup barrier per k-step, software-pipelined by one k-step:
//     wait until this wave's pieces of k-step ks+1 have landed (counted vmcnt)
//     barrier            -> everybody's pieces of ks+1 are in LDS, everybody is done reading ks
//     issue k-step ks+3  -> into the slot of ks
//     MFMA on ks (operands split in the previous round) || read + convert ks+1
// The waves of a workgroup work on the SAME frames and DIFFERENT outputs, so there is no
// cross-wave combine: every wave writes its accumulators straight into the chunk's fp64 slab.
#include <type_traits>

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
  // Per piece: a wave-uniform base pointer and row pitch (scalar registers) and ONE per-lane
  // 32-bit element offset that never changes (row-in-k-step * pitch + column); the k-step's
  // first row is added to the uniform part.
  const float* src_base[PW];
  int src_ld[PW];
  unsigned src_off[PW];
  auto piece_row = [&](int q) __attribute__((always_inline)) {          // row of the k-step this lane's 16 bytes of piece q come from
    const int p = wave + kWideWaves * q;
    return p < Slot::kPA ? (256 * p + 4 * lane) / AL : (256 * (p - Slot::kPA) + 4 * lane) / BD;
  };
#pragma unroll
  for (int q = 0; q < PW; ++q) {
    const int p = wave + kWideWaves * q;
    if (p < Slot::kPA) {
      const int f = 256 * p + 4 * lane;
      int col = by * AL + f % AL;
      col = col + 4 <= ldphi ? col : ldphi - 4;                    // past the padded leaves: any valid column
      src_base[q] = phi;
      src_ld[q] = ldphi;
      src_off[q] = (unsigned)(f / AL) * (unsigned)ldphi + (unsigned)col;
    } else {
      const int f = 256 * (p - Slot::kPA) + 4 * lane;
      src_base[q] = Z;
      src_ld[q] = ldz;
      src_off[q] = (unsigned)(f / BD) * (unsigned)ldz + (unsigned)(bz * BD + f % BD);   // Dp is a multiple of BD (host)
    }
  }
  const unsigned l15 = (unsigned)(lane & 15);
  auto issue = [&](int64_t ks) __attribute__((always_inline)) {
    float* slot = ring + (size_t)(ks % R) * Slot::kFloats;
    const int64_t f0 = n0 + ks * kKStep;
    if (__builtin_expect(f0 + kKStep <= n1, 1)) {                  // wave-uniform
#pragma unroll
      for (int q = 0; q < PW; ++q)
        dma16(src_base[q] + f0 * src_ld[q] + src_off[q], slot + 256 * (wave + kWideWaves * q));
      dma4(vid + f0 + l15, slot + Slot::kA + Slot::kB);            // every wave: same 64 bytes, same place
    } else {                                                       // ragged tail: re-read the last valid row
#pragma unroll
      for (int q = 0; q < PW; ++q) {
        const int64_t over = f0 + piece_row(q) - (n1 - 1);
        const unsigned back = over > 0 ? (unsigned)over * (unsigned)src_ld[q] : 0u;
        dma16(src_base[q] + f0 * src_ld[q] + (src_off[q] - back), slot + 256 * (wave + kWideWaves * q));
      }
      int64_t vr = f0 + l15;
      vr = vr < n1 ? vr : n1 - 1;
      dma4(vid + vr, slot + Slot::kA + Slot::kB);
    }
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
  auto flush_mass = [&]() __attribute__((always_inline)) {
    if (my_vid >= 0) {
#pragma unroll
      for (int a = 0; a < TL; ++a)
        if (lcol[a] < L && msum[a] != 0.f) atomicAdd(&mass[(int64_t)my_vid * L + lcol[a]], (double)msum[a]);
    }
#pragma unroll
    for (int a = 0; a < TL; ++a) msum[a] = 0.f;
  };

  struct Ops {
    Frag ah[TL], am[TL], al[TL], bh[TD], bm[TD], bl[TD];
  };
  // raw operands of k-step ks out of its LDS slot; MASK: zero the phi rows past the chunk's end
  auto load_a = [&](int64_t ks, int a, auto mask_c, float (&x)[8]) __attribute__((always_inline)) {
    const float* slot = ring + (size_t)(ks % R) * Slot::kFloats;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = slot[(8 * h + e) * AL + wl * 64 + 32 * a + i];
    if constexpr (decltype(mask_c)::value) {
      const int64_t f0 = n0 + ks * kKStep;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (f0 + 8 * h + e >= n1) x[e] = 0.f;
    }
  };
  auto load_b = [&](int64_t ks, int b, float (&x)[8]) __attribute__((always_inline)) {
    const float* slot = ring + (size_t)(ks % R) * Slot::kFloats;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = slot[Slot::kA + (8 * h + e) * BD + wd * 64 + 32 * b + i];
  };
  // S0 and the per-video masses of k-step ks (lead waves only)
  // (reads its phi rows again: two of the eight waves do, and the others keep 16 registers)
  auto lead_stats = [&](int64_t ks) __attribute__((always_inline)) {
    const float* slot = ring + (size_t)(ks % R) * Slot::kFloats;
    const int64_t f0 = n0 + ks * kKStep;
    const bool ragged = f0 + kKStep > n1;
    float xa[TL][8];
#pragma unroll
    for (int a = 0; a < TL; ++a) {
      if (ragged)
        load_a(ks, a, std::true_type{}, xa[a]);
      else
        load_a(ks, a, std::false_type{}, xa[a]);
    }
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
  };
  auto mac_all = [&](const Ops& o) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < TL; ++a)
#pragma unroll
      for (int b = 0; b < TD; ++b) mac6(o.ah[a], o.am[a], o.al[a], o.bh[b], o.bm[b], o.bl[b], acc[a][b]);
  };
  // One k-step of the software pipeline: the matrix products of k-step ks (operands `cur`,
  // already split) run while the VALU splits k-step ks+1 out of LDS into `nxt`, so the two
  // pipes of a SIMD overlap inside one wave (with one barrier per k-step the waves of a
  // workgroup move in lockstep and cannot cover for each other).
  auto step = [&](int64_t ks, auto mask_c, const Ops& cur, Ops& nxt) __attribute__((always_inline)) {
    if (ks + 2 < nsteps)
      wait_vmcnt<P>();          // only the pieces of ks+2 may still be in flight
    else
      wait_vmcnt<0>();
    wide_barrier();             // everybody's pieces of ks+1 are in LDS, everybody is done reading ks
    if (ks + 3 < nsteps) issue(ks + 3);
    static_assert(TL == 2 && TD == 2, "interleave written for 2 x 2 tiles");
    // Hand-placed schedule (fences keep the compiler from bunching the 24 matrix instructions and
    // doing the ~180 VALU operations of the split afterwards): tile t of k-step ks+1 (0,1 = Z
    // tiles, 2,3 = phi tiles) is read from LDS a few products ahead of its split, and the 16
    // element pairs of the split are spread over products 3..23.
    float x[4][8];
    auto ld = [&](int t) __attribute__((always_inline)) {
      if (t < 2)
        load_b(ks + 1, t, x[t]);
      else
        load_a(ks + 1, t - 2, mask_c, x[t]);
    };
    auto sp = [&](int t, int q) __attribute__((always_inline)) {
      Frag& H = t == 0 ? nxt.bh[0] : t == 1 ? nxt.bh[1] : t == 2 ? nxt.ah[0] : nxt.ah[1];
      Frag& M = t == 0 ? nxt.bm[0] : t == 1 ? nxt.bm[1] : t == 2 ? nxt.am[0] : nxt.am[1];
      Frag& Lo = t == 0 ? nxt.bl[0] : t == 1 ? nxt.bl[1] : t == 2 ? nxt.al[0] : nxt.al[1];
      split_pair(x[t][2 * q], x[t][2 * q + 1], H.u[q], M.u[q], Lo.u[q]);
      // the results are used one k-step later: without this the optimiser sinks the arithmetic there
      asm volatile("" : "+v"(H.u[q]), "+v"(M.u[q]), "+v"(Lo.u[q]));
    };
    ld(0);
#pragma unroll
    for (int g = 0; g < 24; ++g) {
      const int a = g / 12, b = (g / 6) % 2;
      mac1(g % 6, cur.ah[a], cur.am[a], cur.al[a], cur.bh[b], cur.bm[b], cur.bl[b], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
      if (g == 1) ld(1);
      if (g == 6) ld(2);
      if (g == 11) ld(3);
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (3 + (j * 20) / 15 == g) sp(j / 4, j % 4);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (lead) lead_stats(ks + 1);
  };

  if (nsteps > 0) {
    issue(0);
    if (nsteps > 1) issue(1);
    if (nsteps > 2) issue(2);
    if (nsteps > 2)
      wait_vmcnt<2 * P>();
    else if (nsteps > 1)
      wait_vmcnt<P>();
    else
      wait_vmcnt<0>();
    wide_barrier();
    Ops A, B;
    {
      float x[8];
#pragma unroll
      for (int b = 0; b < TD; ++b) {
        load_b(0, b, x);
        split3(x, A.bh[b], A.bm[b], A.bl[b]);
      }
#pragma unroll
      for (int a = 0; a < TL; ++a) {
        load_a(0, a, std::true_type{}, x);
        split3(x, A.ah[a], A.am[a], A.al[a]);
      }
      if (lead) lead_stats(0);
    }
    int64_t ks = 0;
    for (; ks + 3 < nsteps; ks += 2) {       // k-steps ks+1, ks+2 are whole
      step(ks, std::false_type{}, A, B);
      step(ks + 1, std::false_type{}, B, A);
    }
    const int64_t rem = nsteps - ks;         // 1, 2 or 3 left; only the last one can be ragged
    if (rem == 3) {
      step(ks, std::false_type{}, A, B);
      step(ks + 1, std::true_type{}, B, A);
      mac_all(A);
    } else if (rem == 2) {
      step(ks, std::true_type{}, A, B);
      mac_all(B);
    } else {
      mac_all(A);
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
