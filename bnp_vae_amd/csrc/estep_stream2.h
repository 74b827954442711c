// Streamed regime of the E-step (trees whose packed leaf operands do not fit the LDS: BASELINE config 4), pass A,
// soft mode: raw logits of (frame tile x 128-leaf chunk) + per-chunk (max, sum-exp) partials, as
// estep_logits_kernel (estep_kernels.h) writes them -- same arithmetic, bit-identical output -- but organised
// round what bounds that kernel.  Its ablations at config 4's shape (131k frames, profiles/r03_notes.md): the
// matrix instructions with their operand reads alone take 1.31 of the 2.8 ms, and that skeleton is LDS bandwidth:
// every 16-frame tile re-reads the chunk's whole operand image (128 KiB at D = 256) from LDS, 64 GB per launch
// against the ~60 TB/s all LDS pipes deliver.  Here
//   * a wave takes FT frame tiles at a time and walks the contraction OUTSIDE the leaf tiles, so a leaf fragment
//     read from LDS feeds 3 * FT matrix instructions instead of 3 (LDS traffic / FT), and only the current
//     k-step's frame fragments are in registers (double-buffered: FT * 16 registers instead of the tile's 128);
//   * the frame fragments come pre-packed: a pre-pass (pack_latents_kernel) converts Z once per call into the
//     scaled f16 hi/lo fragments in operand order plus (||z||^2, 2^-k) per frame -- the old kernel repeated that
//     conversion for every one of the L/128 leaf chunks (61 times at config 4); the image is as large as Z
//     (4 bytes per element) and costs one read + one write of it;
//   * the next unit's first fragments and the tile's prior entries are requested under the matrix block.
#pragma once

#include "estep_kernels.h"

namespace bnpvi {

// ---- pre-pass: Z -> B-operand fragments.  zpack[((tile * S + s) * 2 + part) * 64 + lane], zmeta[tile * 16 + j] =
// (||z_j||^2, 2^-kz_j): exactly the values tile_distances derives from the raw rows.
template <int DP>
__global__ __launch_bounds__(256) void pack_latents_kernel(const float* __restrict__ Z, int ldz, int64_t N,
                                                           f16x8* __restrict__ zpack, float2* __restrict__ zmeta) {
  constexpr int S = DP / 32;
  const int lane = threadIdx.x & 63;
  const int64_t ntiles = (N + kFT - 1) / kFT;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    float4 zA[S], zB[S];
    load_ztile<DP>(Z, ldz, tile * kFT, N, lane, zA, zB);
    float zz = 0.f, amax = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float v[8] = {zA[s].x, zA[s].y, zA[s].z, zA[s].w, zB[s].x, zB[s].y, zB[s].z, zB[s].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        zz = fmaf(v[e], v[e], zz);
        amax = fmaxf(amax, fabsf(v[e]));
      }
    }
    zz = cols4_sum(zz);
    amax = cols4_max(amax);
    const int kz = scale_exponent_bits(__float_as_int(amax));
    const float sz = pow2f(kz);
    f16x8* out = zpack + tile * (S * 2 * 64) + lane;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      f16x8 h, l;
      split8(zA[s], zB[s], sz, h, l);
      __builtin_nontemporal_store(h, out + (s * 2 + 0) * 64);
      __builtin_nontemporal_store(l, out + (s * 2 + 1) * 64);
    }
    if (lane < kFT) zmeta[tile * kFT + lane] = make_float2(zz, pow2f(-kz));
  }
}

// d^2 in place from the accumulated products of one frame tile, with the cancellation screen and the exact
// recomputation of tile_distances (estep_kernels.h), whose text this follows
template <int DP, int NT>
__device__ __forceinline__ void finish_distances(const float* __restrict__ Z, int ldz, int64_t row0, int64_t N,
                                                 float zz, float pow_neg_kz, const LeafOperands& lo_, int lane,
                                                 int tiles_valid, int lvalid, f32x4 (&acc)[NT]) {
  const int kg = lane >> 4;
  const float unscale = -2.f * pow_neg_kz * lo_.inv_sa;
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t un2 = {unscale, unscale}, zz2 = {zz, zz}, zero2 = {0.f, 0.f};
  f32x2_t dmin2 = {INFINITY, INFINITY};
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    const float4 aa4 = *reinterpret_cast<const float4*>(lo_.aa + kLT * T + 4 * kg);
    const f32x2_t aaA = {aa4.x, aa4.y}, aaB = {aa4.z, aa4.w};
    const f32x2_t accA = {acc[T][0], acc[T][1]}, accB = {acc[T][2], acc[T][3]};
    const f32x2_t vA = __builtin_elementwise_max(un2 * accA + (zz2 + aaA), zero2);
    const f32x2_t vB = __builtin_elementwise_max(un2 * accB + (zz2 + aaB), zero2);
    dmin2 = __builtin_elementwise_min(dmin2, __builtin_elementwise_min(vA, vB));
    acc[T][0] = vA[0];
    acc[T][1] = vA[1];
    acc[T][2] = vB[0];
    acc[T][3] = vB[1];
  }
  const float dmin = fminf(dmin2[0], dmin2[1]);
  const float snmax = zz + lo_.aamax;
  unsigned pend = 0u;
  if (__builtin_expect(__ballot(dmin < refine_threshold(snmax)) != 0ull, 0)) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      if (T < tiles_valid) {
        const float4 aa4 = *reinterpret_cast<const float4*>(lo_.aa + kLT * T + 4 * kg);
        const float aav[4] = {aa4.x, aa4.y, aa4.z, aa4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sn = zz + aav[r];
          if ((acc[T][r] < refine_threshold(sn)) && (kLT * T + 4 * kg + r < lvalid)) pend |= 1u << ((T * 4 + r) & 31);
        }
      }
    }
  }
  static_assert(NT <= 8, "one bit per (leaf tile, register) in the pending mask");
  while (true) {
    const unsigned long long b = __ballot(pend != 0u);
    if (b == 0ull) break;
    const unsigned g = (unsigned)(b >> (16 * kg)) & 0xffffu;      // uniform over the row group
    if (g != 0u) {
      const int srcj = __ffs((int)g) - 1;                           // owner: frame column of the group
      const unsigned pm = (unsigned)__shfl((int)pend, 16 * kg + srcj);
      const int idx = __ffs((int)pm) - 1;
      int64_t ns = row0 + srcj;
      if (ns >= N) ns = N - 1;
      const float p = row16_direct_dist2<DP>(Z + ns * ldz, lo_.a32 + (kLT * (idx >> 2) + 4 * kg + (idx & 3)) * DP, lane);
      if ((lane & 15) == srcj) {
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (idx == T * 4 + r) acc[T][r] = p;
        pend &= pend - 1u;
      }
    }
  }
}

constexpr int kS2Waves = 8;       // waves per workgroup (they share the chunk's operands in LDS)
#ifndef BNPVI_STREAM2_FT
#define BNPVI_STREAM2_FT 2
#endif
constexpr int kStream2FT = BNPVI_STREAM2_FT;   // frame tiles per wave and unit

template <int DP, int NT, int FT>
__global__ __launch_bounds__(kS2Waves * 64) void estep_logits2_kernel(
    const float* __restrict__ Z, int ldz, const int32_t* __restrict__ vid, int64_t N, int L, int Lp,
    const f16x8* __restrict__ zpack, const float2* __restrict__ zmeta, const f16x8* __restrict__ apack,
    const float* __restrict__ aavec, const float* __restrict__ a32, const float* __restrict__ scal,
    const float* __restrict__ prior, float half_sz, float* __restrict__ phi, float2* __restrict__ part,
    int nchunk) {
  constexpr int S = DP / 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int kg = lane >> 4;
  const int chunk = blockIdx.y;
  const int T0 = chunk * NT;
  int tiles_valid = Lp / kLT - T0;
  tiles_valid = tiles_valid > NT ? NT : tiles_valid;
  const int64_t ntiles = (N + kFT - 1) / kFT;
  const int64_t nunits = (ntiles + FT - 1) / FT;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f16x8* a_lds = reinterpret_cast<f16x8*>(smem_raw);                  // [NT*S*2*64]
  float* aa_lds = reinterpret_cast<float*>(a_lds + NT * S * 2 * 64);  // [16*NT]
  {
    const f16x8* a_src = apack + (int64_t)T0 * S * 2 * 64;
    const float* aa_src = aavec + kLT * T0;
    for (int e = threadIdx.x; e < NT * S * 2 * 64; e += blockDim.x)
      a_lds[e] = e < tiles_valid * S * 2 * 64 ? a_src[e] : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = threadIdx.x; e < kLT * NT; e += blockDim.x) aa_lds[e] = e < kLT * tiles_valid ? aa_src[e] : 0.f;
    __syncthreads();
  }
  LeafOperands lo_{a_lds, aa_lds, a32 + (int64_t)kLT * T0 * DP, scal[0], scal[2]};
  const f16x8* a_lane = a_lds + lane;

  // fragment double buffer: b0 holds the fragments of even k-steps, b1 of odd ones
  f16x8 b0[FT][2], b1[FT][2];
  auto tile_of = [&](int64_t unit, int f) -> int64_t {
    const int64_t t = unit * FT + f;
    return t < ntiles ? t : ntiles - 1;            // past the end: a tile that exists (nothing of it is stored)
  };
  auto load_frags = [&](int64_t unit, int s, f16x8 (&b)[FT][2]) {
#pragma unroll
    for (int f = 0; f < FT; ++f) {
#ifdef BNPVI_ABLATE_S2_NOBLOAD
      const _Float16 c = (_Float16)(float)(lane + s + f);
      b[f][0] = f16x8{c, c, c, c, c, c, c, c};
      b[f][1] = b[f][0];
#else
      const f16x8* zp = zpack + tile_of(unit, f) * (S * 2 * 64) + lane;
      b[f][0] = zp[(s * 2 + 0) * 64];
      b[f][1] = zp[(s * 2 + 1) * 64];
#endif
    }
  };

  const int64_t ustride = (int64_t)gridDim.x * kS2Waves;
  int64_t unit = (int64_t)blockIdx.x * kS2Waves + wave;
  if (unit < nunits) load_frags(unit, 0, b0);
  for (; unit < nunits; unit += ustride) {
    float2 meta[FT];
    int vrow[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int64_t t = tile_of(unit, f);
      meta[f] = zmeta[t * kFT + j];
      const int64_t n = t * kFT + j;
      vrow[f] = vid[n < N ? n : N - 1];
    }
    f32x4 acc[FT][NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // prior entries of a tile: requested a phase ahead of their use (tile 0 under the last k-step's products,
    // tile f+1 at the top of tile f's epilogue), so their round trip is never waited for
    float4 p4[FT][NT];
    auto load_prior = [&](int f) {
      const float* prow = prior + (int64_t)vrow[f] * Lp + kLT * T0 + 4 * kg;
#pragma unroll
      for (int T = 0; T < NT; ++T)
        p4[f][T] = T < tiles_valid ? *reinterpret_cast<const float4*>(prow + kLT * T) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int s = 0; s < S; ++s) {
      f16x8 (&bc)[FT][2] = (s & 1) ? b1 : b0;
      f16x8 (&bn)[FT][2] = (s & 1) ? b0 : b1;
      if (s + 1 < S) load_frags(unit, s + 1, bn);
      if (s == S - 1) load_prior(0);
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        const f16x8 ah = a_lane[((T * S + s) * 2 + 0) * 64];
        const f16x8 al = a_lane[((T * S + s) * 2 + 1) * 64];
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          acc[f][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bc[f][0], s ? acc[f][T] : zero4, 0, 0, 0);
          acc[f][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bc[f][1], acc[f][T], 0, 0, 0);
          acc[f][T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bc[f][0], acc[f][T], 0, 0, 0);
        }
      }
    }
    // the next unit's first fragments fly during the epilogues (for even S, b0 was last read a k-step ago)
    if (unit + ustride < nunits) load_frags(unit + ustride, 0, b0);

#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int64_t tile = unit * FT + f;
      if (tile >= ntiles) break;                                   // wave-uniform
      const int64_t n = tile * kFT + j;
      const bool ok = n < N;
      f32x4 (&d2)[NT] = acc[f];
      if (f + 1 < FT) load_prior(f + 1);
#ifndef BNPVI_ABLATE_S2_NOEPI
      finish_distances<DP, NT>(Z, ldz, tile * kFT, N, meta[f].x, meta[f].y, lo_, lane, tiles_valid, L - kLT * T0, d2);
#endif
      float m = -INFINITY;
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        const bool tv = T < tiles_valid;
        const float pr[4] = {p4[f][T].x, p4[f][T].y, p4[f][T].z, p4[f][T].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#ifdef BNPVI_ABLATE_S2_NOEPI
          const float v = tv ? fmaf(-half_sz, d2[T][r], pr[r] * kLn2) : -INFINITY;
#else
          const float v = tv ? fmaf(-half_sz, __builtin_amdgcn_sqrtf(d2[T][r]), pr[r] * kLn2) : -INFINITY;
#endif
          d2[T][r] = v;
          m = fmaxf(m, v);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float ssum = 0.f;
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#ifdef BNPVI_ABLATE_S2_NOEPI
          ssum += (m == -INFINITY) ? 0.f : (d2[T][r] - m) * kLog2e;
#else
          ssum += (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((d2[T][r] - m) * kLog2e);
#endif
      ssum += __shfl_xor(ssum, 16);
      ssum += __shfl_xor(ssum, 32);
#ifdef BNPVI_ABLATE_S2_NOSTORE
      if (m == -123.f)
#endif
      if (tile * kFT + kFT <= N) {
        // whole 128-byte lines per store instruction (see estep_logits_kernel)
        float* base = phi + (n - j + (j & 7)) * Lp + kLT * T0 + (j >> 3) * kLT + 4 * kg;
#pragma unroll
        for (int T = 0; T + 1 < NT; T += 2) {
          if (T < tiles_valid) {
            f32x4 va, vb;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int a = (int)__float_as_uint(d2[T][c]), b = (int)__float_as_uint(d2[T + 1][c]);
              va[c] = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xc, false));
              vb[c] = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false));
            }
            store_phi4(base + kLT * T, va);
            store_phi4(base + 8 * Lp + kLT * T, vb);
          }
        }
        if (kg == 0) part[n * nchunk + chunk] = make_float2(m, ssum);
      } else if (ok) {
        float* row = phi + n * Lp + kLT * T0;
#pragma unroll
        for (int T = 0; T < NT; ++T)
          if (T < tiles_valid)
            *reinterpret_cast<float4*>(row + kLT * T + 4 * kg) = make_float4(d2[T][0], d2[T][1], d2[T][2], d2[T][3]);
        if (kg == 0) part[n * nchunk + chunk] = make_float2(m, ssum);
      }
    }
  }
}

}  // namespace bnpvi
