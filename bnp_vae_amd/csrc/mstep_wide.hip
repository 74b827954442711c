// M-step data pass for LARGE trees (gfx950): same statistics and arithmetic as mstep_dma.hip
// (exact bf16x3 split, six products on v_mfma_f32_32x32x16_bf16, fp32 partial sums over at most
// 8192 frames, fp64 slabs), with the k-step shared by a whole workgroup.
//
// With many (leaf block, dim block) pairs the per-wave ring of mstep_dma.hip re-reads every Z
// block once per leaf block and every phi block once per dim block through L2 (config 4: 48 GB
// of L2 traffic for 4 GB of HBM traffic -- L2-bandwidth bound).  Here a workgroup of 4 waves
// (ONE per SIMD, 512 registers each) owns a tile of (128 LBW leaves) x (128 DBW dims) over one
// chunk of frames; the 16 frames of a k-step -- phi block [16][128 LBW], Z block [16][128 DBW],
// 16 video ids -- are brought into ONE LDS slot by all waves together (LDS-DMA, 1 KiB per
// instruction, whole row segments) and every wave takes its 128 x 128 sub-tile from it.
//
// Why 128 x 128 per wave: the bf16 split costs ~5.5 VALU operations per operand element and a
// wave splits (TL + TD) 32-wide tiles for TL x TD x 6 matrix instructions.  At 2 x 2 tiles that
// is 7.3 VALU per v_mfma_f32_32x32x16_bf16 (issue-bound: 8 + 4 x 7.3 > 32 cycles per product,
// measured 3.2 ms on config 4); at 4 x 4 tiles it is 3.7, which hides under the 32-cycle product
// (MI355X_MICROARCH.md: up to 5 single-issue fillers per gap are free at one wave per SIMD).
//
// Ring of 3 slots, ONE workgroup barrier per k-step, software-pipelined by one k-step:
//     wait until this wave's pieces of k-step ks+1 have landed (counted vmcnt)
//     barrier            -> everybody's pieces of ks+1 are in LDS, everybody is done reading ks
//     issue k-step ks+3  -> into the slot of ks
//     MFMA on ks (operands split in the previous round) || read + convert ks+1
// The waves of a workgroup work on the SAME frames and DIFFERENT outputs, so there is no
// cross-wave combine: every wave writes its accumulators straight into the chunk's fp64 slab.
#include <cstring>
#include <type_traits>

#include "mstep_common.h"

namespace bnpvi {

constexpr int kWideWaves = 4;
constexpr int kWideRing = 3;
// padded leaves x padded dims from which the two-piece f16 split is taken (500k frames, kernel us, exact -> f16:
// D=128 L=1024 773 -> 565, L=4096 2834 -> 1764; D=256 L=256 426 -> 391 (call time 417 -> 435: the max|Z| pass),
// L=512 726 -> 561, L=1296 1868 -> 1216; D=200 L=3125 4516 -> 2787)
constexpr int64_t kWideF16From = (int64_t)512 * 256;

template <int LBW, int DBW, int TL, int TD>   // waves and 32-wide tiles per wave: leaves, dims
struct WideSlot {
  static constexpr int kAL = 32 * TL * LBW;                  // leaves per workgroup
  static constexpr int kBD = 32 * TD * DBW;                  // dims per workgroup
  static constexpr int kA = kKStep * kAL;               // floats of the phi block
  static constexpr int kB = kKStep * kBD;               // floats of the Z block
  static constexpr int kFloats = kA + kB + 64;          // + the video ids (16 used)
  static constexpr int kPA = kA / 256;                  // 1 KiB pieces of the phi block
  static constexpr int kPT = (kA + kB) / 256;           // ... of both blocks
  static constexpr int kPW = kPT / kWideWaves;          // pieces per wave
  static_assert(LBW * DBW == kWideWaves, "4 waves per workgroup");
  static_assert(kPT % kWideWaves == 0, "pieces must divide evenly over the waves");
  static_assert(kPA % kWideWaves == 0, "a piece index is a phi piece for every wave or for none");
};

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Where the pieces of the next k-step's split go between the matrix instructions of this one.
// Tile t (t < TD: Z tiles, then the phi tiles) has 4 element pairs, a pair is split in three
// stages of 4 + 4 + 3 VALU operations; stage idx = 12 t + 3 pair + st runs behind product
// stage_pos(idx), tile t is read from LDS ~9 products ahead of its first stage.
template <int TL, int TD, int PP = 6>     // PP: products per (leaf tile, dim tile) pair
struct WideSched {
  static constexpr int NT = TL + TD, NM = TL * TD * PP, NS = 12 * NT;
  static constexpr int kLead = 3;                         // products before the first stage
  static constexpr int stage_pos(int idx) { return kLead + (idx * (NM - 1 - kLead)) / (NS - 1); }
  static constexpr int first_stage_at(int g) {
    int idx = 0;
    while (idx < NS && stage_pos(idx) < g) ++idx;
    return idx;
  }
  static constexpr int stages_at(int g) {
    int n = 0;
    for (int idx = 0; idx < NS; ++idx) n += stage_pos(idx) == g;
    return n;
  }
  static constexpr int load_pos(int t) { return stage_pos(12 * t) > 9 ? stage_pos(12 * t) - 9 : 0; }
  static constexpr int load_at(int g) {                   // tile to read behind product g, or -1
    for (int t = 1; t < NT; ++t)
      if (load_pos(t) == g) return t;
    return -1;
  }
  // every stage is placed exactly once, every tile is read before its first stage, and no two
  // tiles are read behind the same product (load_at returns one tile per product)
  static constexpr bool consistent() {
    int placed = 0;
    for (int g = 0; g < NM; ++g) placed += stages_at(g);
    if (placed != NS) return false;
    for (int t = 1; t < NT; ++t) {
      if (load_pos(t) >= stage_pos(12 * t) || load_at(load_pos(t)) != t) return false;
    }
    return true;
  }
  static constexpr int max_stages() {
    int m = 0;
    for (int g = 0; g < NM; ++g) m = stages_at(g) > m ? stages_at(g) : m;
    return m;
  }
};

__device__ __forceinline__ void wide_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// F16: the operands are split into TWO f16 pieces after a power-of-two scaling (phi by 2^14, Z by the power of
// two that brings max|Z| -- zamax[0], found by absmax_kernel -- into [2^14, 2^15)) and THREE products are taken
// (lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16): the E-step's arithmetic (operands good to 2^-22, the
// lo*lo product dropped) instead of the exact bf16 x 3 split's six products.  For trees large enough that the
// matrix pipe bounds the pass (config 4) that halves its time.
template <int LBW, int DBW, int TL, int TD, bool F16>
__global__ __launch_bounds__(kWideWaves * 64) void leaf_stats_wide_kernel(
    const float* __restrict__ Z, int ldz, const float* __restrict__ phi, int ldphi,
    const int32_t* __restrict__ vid, int64_t N, int D, int L, int64_t chunk, double* __restrict__ slab,
    int W, double* __restrict__ mass, int gx, int gy, int gz, const uint32_t* __restrict__ zamax) {
  using Slot = WideSlot<LBW, DBW, TL, TD>;
  constexpr int R = kWideRing;
  constexpr int AL = Slot::kAL, BD = Slot::kBD, PW = Slot::kPW;
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
  // S0 and the masses: the DBW waves that share a row of leaf tiles each take TL / DBW of them
  constexpr int TS = TL / DBW;
  static_assert(TS * DBW == TL, "leaf tiles must divide over the waves of a row");
  const bool lead = (bz == 0);
  const bool do_mass = lead && (mass != nullptr);
  constexpr int PP = F16 ? 3 : 6;
  float sc_a = 1.f, sc_b = 1.f;                  // F16: operand scales (powers of two)
  double unscale = 1.0;
  if constexpr (F16) {
    const uint32_t eb = (zamax[0] >> 23) & 0xffu;                  // exponent of max|Z|
    const int kb = (eb == 0u || eb == 255u) ? 0 : 14 - ((int)eb - 127);
    sc_a = 16384.f;
    sc_b = __uint_as_float((uint32_t)(127 + (kb < -100 ? -100 : (kb > 100 ? 100 : kb))) << 23);
    unscale = 1.0 / ((double)sc_a * (double)sc_b);
  }

  const int64_t n0 = (int64_t)cx * chunk < N ? (int64_t)cx * chunk : N;
  const int64_t n1 = (n0 + chunk < N) ? n0 + chunk : N;
  const int64_t nsteps = (n1 - n0 + kKStep - 1) / kKStep;          // same for all waves of the workgroup

  // ---- this wave's DMA pieces: piece p = wave + 4 q covers floats [256 p, 256 p + 256) of the slot
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
  // one DMA instruction of a WHOLE k-step: piece q < PW, or (q == PW) the video ids
  auto issue_piece = [&](int q, int64_t f0, float* slot) __attribute__((always_inline)) {
    // kPA is a multiple of the wave count, so piece q is a phi piece for every wave or for none: phi streams
    // past the caches (non-temporal, as in mstep_dma.hip), Z keeps the default policy
    if (q < PW && q * kWideWaves < Slot::kPA)
      dma16_nt(src_base[q] + f0 * src_ld[q] + src_off[q], slot + 256 * (wave + kWideWaves * q));
    else if (q < PW)
      dma16(src_base[q] + f0 * src_ld[q] + src_off[q], slot + 256 * (wave + kWideWaves * q));
    else
      dma4(vid + f0 + l15, slot + Slot::kA + Slot::kB);            // every wave: same 64 bytes, same place
  };
  auto issue = [&](int64_t ks, int m) __attribute__((always_inline)) {
    float* slot = ring + (size_t)m * Slot::kFloats;
    const int64_t f0 = n0 + ks * kKStep;
    if (__builtin_expect(f0 + kKStep <= n1, 1)) {                  // wave-uniform
#pragma unroll
      for (int q = 0; q <= PW; ++q) issue_piece(q, f0, slot);
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
  float s0[TS], msum[TS];
  S0Check s0chk[TS];
  int lcol[TS];
#pragma unroll
  for (int a = 0; a < TS; ++a) {
    s0[a] = msum[a] = 0.f;
    lcol[a] = by * AL + wl * (32 * TL) + 32 * (wd * TS + a) + i;
  }
  int my_vid = -1;
  auto flush_mass = [&]() __attribute__((always_inline)) {
    if (my_vid >= 0) {
#pragma unroll
      for (int a = 0; a < TS; ++a)
        if (lcol[a] < L && msum[a] != 0.f) atomicAdd(&mass[(int64_t)my_vid * L + lcol[a]], (double)msum[a]);
    }
#pragma unroll
    for (int a = 0; a < TS; ++a) msum[a] = 0.f;
  };

  struct Ops {
    Frag ah[TL], am[TL], al[TL], bh[TD], bm[TD], bl[TD];
  };
  // raw operands of k-step ks out of its LDS slot; MASK: zero the phi rows past the chunk's end
  auto load_a = [&](const float* slot, int64_t ks, int a, auto mask_c, float (&x)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = slot[(8 * h + e) * AL + wl * (32 * TL) + 32 * a + i];
    if constexpr (decltype(mask_c)::value) {
      const int64_t f0 = n0 + ks * kKStep;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (f0 + 8 * h + e >= n1) x[e] = 0.f;
    }
  };
  auto load_b = [&](const float* slot, int b, float (&x)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = slot[Slot::kA + (8 * h + e) * BD + wd * (32 * TD) + 32 * b + i];
  };
  // S0 and the per-video masses of k-step ks from this wave's share of the phi tiles.  The rows
  // are read a second time (unmasked, by load_stat, early enough for the latency to hide).
  auto load_stat = [&](const float* slot, float (&xs)[TS][8], int& v15) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < TS; ++a) load_a(slot, 0, wd * TS + a, std::false_type{}, xs[a]);
    v15 = reinterpret_cast<const int*>(slot + Slot::kA + Slot::kB)[lane & 15];
  };
  auto lead_stats = [&](const float* slot, int64_t ks, float (&xa)[TS][8], int v15) __attribute__((always_inline)) {
    const int64_t f0 = n0 + ks * kKStep;
    const bool ragged = f0 + kKStep > n1;
    if (ragged) {
#pragma unroll
      for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (f0 + 8 * h + e >= n1) xa[a][e] = 0.f;
    }
    const int* vslot = reinterpret_cast<const int*>(slot + Slot::kA + Slot::kB);
    bool same_video = !do_mass;
    if (do_mass && !ragged) same_video = __ballot(v15 != my_vid) == 0ull;
    float t[TS];
#pragma unroll
    for (int a = 0; a < TS; ++a) {
      t[a] = ((xa[a][0] + xa[a][1]) + (xa[a][2] + xa[a][3])) + ((xa[a][4] + xa[a][5]) + (xa[a][6] + xa[a][7]));
      s0_accumulate(s0[a], t[a]);            // mstep_common.h: why this is not `s0[a] += t[a]`
      s0chk[a].add(t[a]);
    }
    if (__builtin_expect(same_video, 1)) {
#pragma unroll
      for (int a = 0; a < TS; ++a) msum[a] += t[a];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ve = vslot[8 * h + e];
        if (ve != my_vid && (!ragged || f0 + 8 * h + e < n1)) {   // video boundary
          flush_mass();
          my_vid = ve;
        }
#pragma unroll
        for (int a = 0; a < TS; ++a) msum[a] += xa[a][e];
      }
    }
  };
  auto mac_all = [&](const Ops& o) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < TL; ++a)
#pragma unroll
      for (int b = 0; b < TD; ++b) {
        if constexpr (F16)
          mac3h(o.ah[a], o.al[a], o.bh[b], o.bl[b], acc[a][b]);
        else
          mac6(o.ah[a], o.am[a], o.al[a], o.bh[b], o.bm[b], o.bl[b], acc[a][b]);
      }
  };
  // One k-step of the software pipeline: the matrix products of k-step ks (operands `cur`,
  // already split) run while the VALU splits k-step ks+1 out of LDS into `nxt`, so the two
  // pipes of a SIMD overlap inside one wave (with one barrier per k-step the waves of a
  // workgroup move in lockstep and cannot cover for each other).  m1 = ring slot of ks+1; k-step
  // ks+3 goes into the slot of ks.  FAST (main loop): ks+2 and a whole ks+3 exist, and the DMA
  // instructions of ks+3 are placed between the products too.
  auto next_slot = [](int m) { return m == R - 1 ? 0 : m + 1; };
  auto step = [&](int64_t ks, int m1, auto mask_c, auto fast_c, const Ops& cur, Ops& nxt)
                  __attribute__((always_inline)) {
    constexpr bool kFast = decltype(fast_c)::value;
    const int m3 = next_slot(next_slot(m1));
    const float* slot1 = ring + (size_t)m1 * Slot::kFloats;
    float* slot3 = ring + (size_t)m3 * Slot::kFloats;
    const int64_t f3 = n0 + (ks + 3) * kKStep;
    if (kFast || ks + 2 < nsteps)
      wait_vmcnt<P>();          // only the pieces of ks+2 may still be in flight
    else
      wait_vmcnt<0>();
    wide_barrier();             // everybody's pieces of ks+1 are in LDS, everybody is done reading ks
#ifndef BNPVI_MABLATE_NOLOAD
    if constexpr (!kFast) {
      if (ks + 3 < nsteps) issue(ks + 3, m3);
    }
#endif
    // Hand-placed schedule (WideSched; fences keep the compiler from bunching the matrix
    // instructions and doing the VALU work of the split afterwards).
    using Sched = WideSched<TL, TD, PP>;
    static_assert(Sched::consistent(), "split schedule");
    constexpr int NT = Sched::NT;
    float x[NT][8];
    auto ld = [&](int t) __attribute__((always_inline)) {
      if (t < TD)
        load_b(slot1, t, x[t]);
      else
        load_a(slot1, ks + 1, t - TD, mask_c, x[t]);
    };
    float r0 = 0.f, r1 = 0.f, q0 = 0.f, q1 = 0.f;       // residuals of the pair being split
    auto stage = [&](int t, int q, int st) __attribute__((always_inline)) {
      const float x0 = x[t][2 * q], x1 = x[t][2 * q + 1];
#ifdef BNPVI_MABLATE_NOSPLIT
      if (st < 2) return;
      r0 = q0 = x0;
      r1 = q1 = x1;
#endif
      if constexpr (F16) {
        const float sc = t < TD ? sc_b : sc_a;
        if (st == 0) {
          r0 = x0 * sc;
          r1 = x1 * sc;
          q0 = __uint_as_float(pack_f16(r0, r1));                  // the hi pieces
          asm volatile("" : "+v"(r0), "+v"(r1), "+v"(q0));
        } else if (st == 1) {
          r0 -= f16_lo_as_float(__float_as_uint(q0));
          r1 -= f16_hi_as_float(__float_as_uint(q0));
          asm volatile("" : "+v"(r0), "+v"(r1));
        } else {
          Frag& H = t < TD ? nxt.bh[t] : nxt.ah[t - TD];
          Frag& Lo = t < TD ? nxt.bl[t] : nxt.al[t - TD];
          H.u[q] = __float_as_uint(q0);
          Lo.u[q] = pack_f16(r0, r1);
          asm volatile("" : "+v"(H.u[q]), "+v"(Lo.u[q]));
        }
        return;
      }
      if (st == 0) {
        r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
        r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        asm volatile("" : "+v"(r0), "+v"(r1));
      } else if (st == 1) {
        q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u);
        q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        asm volatile("" : "+v"(q0), "+v"(q1));
      } else {
        Frag& H = t < TD ? nxt.bh[t] : nxt.ah[t - TD];
        Frag& M = t < TD ? nxt.bm[t] : nxt.am[t - TD];
        Frag& Lo = t < TD ? nxt.bl[t] : nxt.al[t - TD];
        H.u[q] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
        M.u[q] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
        Lo.u[q] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
        // used one k-step later: without this the optimiser sinks the arithmetic there
        asm volatile("" : "+v"(H.u[q]), "+v"(M.u[q]), "+v"(Lo.u[q]));
      }
    };
    float xs[TS][8];
    int v15 = 0;
    ld(0);
    static_for<Sched::NM>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      constexpr int a = (g / PP) / TD, b = (g / PP) % TD;
      if constexpr (F16)
        mac1h(g % PP, cur.ah[a], cur.al[a], cur.bh[b], cur.bl[b], acc[a][b]);
      else
        mac1(g % 6, cur.ah[a], cur.am[a], cur.al[a], cur.bh[b], cur.bm[b], cur.bl[b], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (Sched::load_at(g) >= 0) ld(Sched::load_at(g));
      if constexpr (g == Sched::NM - 8) load_stat(slot1, xs, v15);
#ifndef BNPVI_MABLATE_NOLOAD
      if constexpr (kFast && g >= 2 && g % 2 == 0 && (g - 2) / 2 <= PW) issue_piece((g - 2) / 2, f3, slot3);
#endif
      static_for<Sched::max_stages()>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < Sched::stages_at(g)) {
          constexpr int idx = Sched::first_stage_at(g) + k;
          stage(idx / 12, (idx / 3) % 4, idx % 3);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if (lead) lead_stats(slot1, ks + 1, xs, v15);
  };

  if (nsteps > 0) {
    issue(0, 0);
    if (nsteps > 1) issue(1, 1);
    if (nsteps > 2) issue(2, 2);
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
        load_b(ring, b, x);
        if constexpr (F16)
          split2h(x, sc_b, A.bh[b], A.bl[b]);
        else
          split3(x, A.bh[b], A.bm[b], A.bl[b]);
      }
#pragma unroll
      for (int a = 0; a < TL; ++a) {
        load_a(ring, 0, a, std::true_type{}, x);
        if constexpr (F16)
          split2h(x, sc_a, A.ah[a], A.al[a]);
        else
          split3(x, A.ah[a], A.am[a], A.al[a]);
      }
      if (lead) {
        float xs[TS][8];
        int v15;
        load_stat(ring, xs, v15);
        lead_stats(ring, 0, xs, v15);
      }
    }
    int64_t ks = 0;
    int m1 = 1;                              // ring slot of k-step ks+1
    for (; ks + 5 < nsteps; ks += 2) {       // k-steps up to ks+4 are whole
      step(ks, m1, std::false_type{}, std::true_type{}, A, B);
      m1 = next_slot(m1);
      step(ks + 1, m1, std::false_type{}, std::true_type{}, B, A);
      m1 = next_slot(m1);
    }
    for (; ks + 1 < nsteps; ++ks) {          // the last few k-steps (the last one may be ragged)
      step(ks, m1, std::true_type{}, std::false_type{}, A, B);
      m1 = next_slot(m1);
      A = B;
    }
    mac_all(A);
  }
  if (do_mass) flush_mass();

  // ---- every wave owns its outputs: fp64 straight into the chunk's slab
  double* out = slab + (int64_t)cx * L * W;
#pragma unroll
  for (int a = 0; a < TL; ++a)
#pragma unroll
    for (int b = 0; b < TD; ++b) {
      const int dim = bz * BD + wd * (32 * TD) + 32 * b + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int leaf = by * AL + wl * (32 * TL) + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (leaf < L && dim < D) out[(int64_t)leaf * W + 1 + dim] = F16 ? (double)acc[a][b][r] * unscale : (double)acc[a][b][r];
      }
    }
  if (lead) {
#pragma unroll
    for (int a = 0; a < TS; ++a) {
      s0chk[a].verify(s0[a], "leaf_stats_wide", a);
      const float t = s0[a] + __shfl_xor(s0[a], 32);
      if (h == 0 && lcol[a] < L) out[(int64_t)lcol[a] * W] = (double)t;
    }
  }
}

// max |Z| as the bit pattern of a non-negative float (unsigned order = float order); out[0] zeroed by the caller
__global__ __launch_bounds__(256) void absmax_kernel(const float4* __restrict__ Z4, int64_t n4, uint32_t* out) {
  float m = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = Z4[e];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  // one atomic per workgroup (every wave of a large grid on ONE address serialises: 195 us at 131k frames)
  if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

template <int LBW, int DBW, int TL, int TD, bool F16 = false>
static int run_wide(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi, const int32_t* vid,
                    int64_t N, int D, int L, int W, double* stats, double* mass, hipStream_t s) {
  using Slot = WideSlot<LBW, DBW, TL, TD>;
  const int gy = (int)ceil_div(ldphi, Slot::kAL), gz = ldz / Slot::kBD;
  const size_t lds = (size_t)kWideRing * Slot::kFloats * sizeof(float);
  // One workgroup per CU at a time (LDS ring + 512 registers per wave).  About two rounds of
  // workgroups, the number of chunks a multiple of 8 (chunk cx runs on XCD cx % 8, so anything else
  // leaves one XCD a round behind), chunks of at most kMaxChunk frames (fp32 partial sums); long
  // chunks keep the fp64 slab traffic (gx L W 8 bytes, written and reduced) small.
  int rounds = 2;
  if (const char* e = dev_getenv("BNPVI_WIDE_ROUNDS")) rounds = atoi(e) > 0 ? atoi(e) : rounds;
  int64_t gx = (int64_t)rounds * ctx->num_cu / ((int64_t)gy * gz);
  gx = gx >= 8 ? gx / 8 * 8 : (gx < 1 ? 1 : gx);
  if (ceil_div(N, gx) > kMaxChunk) gx = ceil_div(ceil_div(N, (int64_t)kMaxChunk), 8) * 8;
  int64_t chunk = ceil_div(N, gx);
  chunk = (chunk + kKStep - 1) / kKStep * kKStep;
  if (chunk < 8 * kKStep) chunk = 8 * kKStep;
  gx = ceil_div(N, chunk);
  const int64_t cap = slab_cap(ctx, (size_t)L * W * sizeof(double));
  if (gx > cap && cap >= 1) {          // bounded workspace: fewer workgroups with longer chunks
    const int64_t c2 = (ceil_div(N, cap) + kKStep - 1) / kKStep * kKStep;
    if (c2 <= kMaxChunk) {
      chunk = c2;
      gx = ceil_div(N, chunk);
    }
  }
  const int64_t blocks = ceil_div(gx, 8) * 8 * gy * gz;
  BNPVI_REQUIRE(blocks < (int64_t(1) << 31), "leaf_stats: grid of %lld workgroups", (long long)blocks);
  double* slab_ws = nullptr;
  int rc = slab_workspace(ctx, (size_t)gx * L * W * sizeof(double), &slab_ws);
  if (rc) return rc;
  auto kern = leaf_stats_wide_kernel<LBW, DBW, TL, TD, F16>;
  BNPVI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const bool timed = ctx->profiling && ctx->timer[BNPVI_PROF_MSTEP].begin(s);
  uint32_t* zamax = nullptr;
  if (F16) {                                    // the scale of the Z pieces: one more read of Z (rows are ldz long, padding zero)
    zamax = (uint32_t*)ctx->small;
    BNPVI_HIP(hipMemsetAsync(zamax, 0, sizeof(uint32_t), s));
    const int64_t n4 = N * ldz / 4;
    int64_t ab = ceil_div(n4, (int64_t)256 * 8);
    if (ab > (int64_t)ctx->num_cu * 4) ab = (int64_t)ctx->num_cu * 4;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)ab), dim3(256), 0, s, (const float4*)Z, n4, zamax);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kWideWaves * 64), lds, s, Z, ldz, phi, ldphi, vid, N, D, L,
                     chunk, slab_ws, W, mass, (int)gx, gy, gz, zamax);
  if (timed) ctx->timer[BNPVI_PROF_MSTEP].end(s);
  BNPVI_LAUNCH_CHECK();
  return launch_slab_reduce(slab_ws, (int)gx, (int64_t)L * W, stats, s);
}

// padded leaves from which the workgroup-wide tiles take over from the per-wave ring, by padded latent dim
constexpr int kWideFrom256 = 160, kWideFrom128 = 320, kWideFrom64 = 384;

int launch_leaf_stats_wide(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                           const int32_t* vid, int64_t N, int D, int L, int V, double* stats, double* mass,
                           hipStream_t s) {
  (void)V;
  const int W = 1 + D;
  if (N < kKStep || (ldphi % 4) || vid == nullptr) return BNPVI_EUNSUPPORTED;
  // Workgroup tiles: 256 leaves x 256 dims when D is padded to 256 (phi is then read exactly once),
  // 512 x 128 at D <= 128, 512 x 64 at D <= 64.  Thresholds measured against the per-wave ring of
  // mstep_dma.hip (1M frames, kernel us, ring -> wide): D=256 L=256 993 -> 737, L=343 1637 -> 1196;
  // D=128 L=512 1040 -> 787, L=625 1606 -> 1267; D=64 L=512 (2M frames) 1113 -> 993.
  // F16 mode (three products instead of six) where the matrix pipe bounds the pass (kWideF16From); smaller trees
  // are HBM-bound and keep the exact split.
  const char* env = dev_getenv("BNPVI_WIDE_F16");                       // development switch: "off" / "on"
  const bool can16 = ctx->small != nullptr && (((uintptr_t)Z) & 15) == 0;
  const bool f16 = can16 && (env ? strcmp(env, "off") != 0 : ((int64_t)ldphi * ldz >= kWideF16From));
  // A workgroup tile need not be full: columns past the padded leaves re-read a valid column and are dropped at the
  // output, so a tile costs the same whatever part of it the tree fills, while the ring's time grows with the
  // leaves.  From the regime sweep (1M frames; ring at L = 144 / 196 / 400 against ONE wide tile): D = 256 896 /
  // 1166 us against ~700; D = 128 1105 us (L = 400) against 750; D = 64 625 us (L = 400) against 534.
  const int from256 = kWideFrom256, from128 = kWideFrom128, from64 = kWideFrom64;
  if (ldz % 256 == 0 && ldphi >= from256 && f16)
    return run_wide<2, 2, 4, 4, true>(ctx, Z, ldz, phi, ldphi, vid, N, D, L, W, stats, mass, s);
  if (ldz % 256 == 0 && ldphi >= from256)
    return run_wide<2, 2, 4, 4>(ctx, Z, ldz, phi, ldphi, vid, N, D, L, W, stats, mass, s);
  if (ldz % 128 == 0 && ldz % 256 != 0 && ldphi >= from128 && f16)
    return run_wide<4, 1, 4, 4, true>(ctx, Z, ldz, phi, ldphi, vid, N, D, L, W, stats, mass, s);
  if (ldz % 128 == 0 && ldz % 256 != 0 && ldphi >= from128)
    return run_wide<4, 1, 4, 4>(ctx, Z, ldz, phi, ldphi, vid, N, D, L, W, stats, mass, s);
  if (ldz == 64 && ldphi >= from64) return run_wide<4, 1, 4, 2>(ctx, Z, ldz, phi, ldphi, vid, N, D, L, W, stats, mass, s);
  return BNPVI_EUNSUPPORTED;
}

}  // namespace bnpvi
