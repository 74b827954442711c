// E-step kernels for gfx950 (MI355X), templated on the padded latent dim DP.
//   logit[n,l] = prior[vid(n),l] - 0.5*sZ*||z_n - alpha_l||_2      (UN-squared norm)
//   phi[n,:]   = softmax_l(logit[n,:])
// replacing /root/reference/src/var_inf.py:313-357 and :383-393 (the per-leaf NumPy norm, the
// per-frame Python loop with two list.index lookups, and the five tree walks of the softmax).
//
// Dense part.  ||z-a||^2 = ||z||^2 - 2 z.a + ||a||^2; only z.a is matrix-shaped.  It runs on
// the f16 matrix pipe (v_mfma_f32_16x16x32_f16, 16x the f32-MFMA rate) WITHOUT giving up fp32
// accuracy: every fp32 operand is split exactly into hi + lo halves (11 + 11 significant bits
// after a power-of-two scaling that keeps both halves in f16 range),
//     z.a = zh.ah + zl.ah + zh.al      (zl.al ~ 2^-22 relative, dropped)
// three products per K-step, fp32 accumulation in the matrix core.  Operand rounding is
// <= 2^-22 relative per element -- below the fp32 accumulation rounding that an all-fp32
// kernel has as well (measured against the fp64 oracle in tests/test_gpu_parity.py).
//
// Layout.  rows i = leaves (A operand = alpha, pre-packed in operand order, from LDS or L2),
// cols j = frames (B operand = Z, 16 frames per wave tile, converted in registers).  The
// accumulator then holds one FRAME per lane column and 4 consecutive LEAVES per lane
// (row = 4*(lane>>4)+reg), so max / sum-exp of the softmax are in-register loops plus two
// cross-lane exchanges (lane^16, lane^32), and prior loads / phi stores are 16 B per lane,
// 64 B contiguous per frame.  The contraction index is permuted so that a lane's fragment of
// K-step s is two float4 of its own row: dims 32s+4kg+{0..3} and 32s+16+4kg+{0..3}
// (kg = lane>>4); pack_leaves_kernel writes alpha in exactly that order.
#pragma once

#include "common.h"

namespace bnpvi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kEWaves = 4;            // waves per workgroup
constexpr int kFT = 16;               // frames per wave tile
constexpr int kLT = 16;               // leaves per MFMA tile
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
// When is d^2 = ||z||^2 + ||a||^2 - 2 z.a good enough?  With s = ||z||^2 + ||a||^2 its absolute error is
// bounded by E = 2e-7 * s: the products of the f16 hi/lo parts are exact, what is lost is (i) the operands'
// third part (|x - hi - lo| <= 2^-22 |x| each) and the dropped lo*lo product, together <= 3 * 2^-22 * sum|z_i a_i|
// <= 3.6e-7 * s / 2, (ii) the fp32 roundings of the accumulator chain (3 * DP/32 of them, each <= 2^-24 |z.a|)
// and (iii) those of ||z||^2, ||a||^2 and their sum (~2^-23 * s) -- the worst cases do not coincide.  The bound
// is checked through its consequence: tools/refine_fraction.py finds log phi within 3e-5 of the fp64 evaluation
// of the same logits wherever phi >= 1e-6, on the bench mixtures and their noise-0.5 variants.  The
// logit needs the UN-squared distance, whose error is E / (2 d).  It is accepted when EITHER
//   * it is below 4e-5 absolute:  d >= 2.5e-3 * s, i.e. d^2 >= kRefineRel * s^2  (logit error <= 2e-5 at
//     sigma_Z = 1; the parity tests hold log phi to 1e-4), OR
//   * it is below 2^-22 * d, four units in the last place of the fp32 distance itself:  d^2 >= E / 2^-21 =
//     kRefineLin * s.  Without this branch every ordinary pair of a widely spread data set (||z|| ~ 256, leaves
//     ~360 apart) was "too close" by the absolute rule although neither the direct-difference path nor the fp32
//     logit it feeds can do better than an ulp of d there: 10-15 % of ALL pairs took the exact path (round 2's
//     8x cliff, profiles/r03_notes.md).
// Below both bounds the distance is recomputed by direct differences (a frame that coincides with a leaf mean
// -- exactly what weighted_kmeans produces right after a split -- or a tight cluster far from the origin).
constexpr float kRefineRel = 6.25e-6f;
constexpr float kRefineLin = 0.42f;
// the bound on d^2 for s = ||z||^2 + ||a||^2; increasing in s, so the bound for the largest ||a||^2 screens a tile
__device__ __forceinline__ float refine_threshold(float s) {
#ifdef BNPVI_ABLATE_NOSPLIT
  return -1.f;
#endif
  return s * fminf(kRefineRel * s, kRefineLin);
}
constexpr int kScaleExp = 13;         // operands are scaled so that max |x| lies in [2^13, 2^14)

// power-of-two scale that brings `amax` into [2^13, 2^14); returns the exponent k (scale = 2^k)
__host__ __device__ __forceinline__ int scale_exponent_bits(int float_bits) {
  int eb = (float_bits >> 23) & 0xff;
  eb = eb < 80 ? 80 : eb;                      // |x| < 2^-47 is treated as 2^-47 (k <= 60)
  return kScaleExp - (eb - 127);
}
__device__ __forceinline__ float pow2f(int k) { return __int_as_float((127 + k) << 23); }

// exact hi/lo split of 8 scaled fp32 values into two f16x8 fragments
__device__ __forceinline__ void split8(const float4& a, const float4& b, float s, f16x8& hi, f16x8& lo) {
#ifdef BNPVI_ABLATE_NOSPLIT   // development only: the conversion's VALU work taken out (results are garbage)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t ra = {a.x, a.y, a.z, a.w}, rb = {b.x * s, b.y, b.z, b.w};
  hi = __builtin_bit_cast(f16x8, ra);
  lo = __builtin_bit_cast(f16x8, rb);
  return;
#endif
  const float v[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)v[e];
    hi[e] = h;
    lo[e] = (_Float16)(v[e] - (float)h);
  }
}

// ---- cross-lane reductions on the VALU (no LDS round trips) ---------------------------------
// The 4 lanes {l, l^16, l^32, l^48} (same column of the four 16-lane rows): permlane16/32 swaps.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cols4_sum(float x) {
  u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float cols4_max(float x) {
  u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// The 16 lanes of a row: DPP row rotations by 1, 2, 4, 8.
#define BNPVI_ROR(v, n) __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x120 + (n), 0xf, 0xf, false))
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, BNPVI_ROR(v, 1));
  v = fmaxf(v, BNPVI_ROR(v, 2));
  v = fmaxf(v, BNPVI_ROR(v, 4));
  return fmaxf(v, BNPVI_ROR(v, 8));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += BNPVI_ROR(v, 1);
  v += BNPVI_ROR(v, 2);
  v += BNPVI_ROR(v, 4);
  return v + BNPVI_ROR(v, 8);
}

// Everything a wave needs to know about the leaf side
struct LeafOperands {
  const f16x8* a;        // packed alpha: [(T*S+s)*2+part][64 lanes] x 8 halfs (LDS or global)
  const float* aa;       // ||alpha_l||^2 (fp32 of the fp32-rounded alpha), tile-relative
  const float* a32;      // row-major fp32 alpha [leaf][DP], tile-relative (refinement path)
  float inv_sa;          // 2^-ka, ka = scale exponent of alpha
  float aamax;           // max_l ||alpha_l||^2
};

// ||z - a||^2 of one (frame, leaf) pair by direct fp32 differences, computed by the 16 lanes of a DPP
// row together: lane j takes float4 number j, j+16, ... of both rows (coalesced 256 B per row and
// step), row16_sum adds the partial sums.  All 16 lanes of the row must be active; every lane
// returns the sum.  Rows are 16-byte aligned and DP floats long (padded).
template <int DP>
__device__ __forceinline__ float row16_direct_dist2(const float* __restrict__ zrow, const float* __restrict__ arow,
                                                    int lane) {
  float p = 0.f;
  for (int q = lane & 15; q < DP / 4; q += 16) {
    const float4 z4 = *reinterpret_cast<const float4*>(zrow + 4 * q);
    const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * q);
    float d;
    d = z4.x - a4.x; p = fmaf(d, d, p);
    d = z4.y - a4.y; p = fmaf(d, d, p);
    d = z4.z - a4.z; p = fmaf(d, d, p);
    d = z4.w - a4.w; p = fmaf(d, d, p);
  }
  return row16_sum(p);
}

// ------------------------------------------------------------------------------------------
// One wave tile: 16 frames x NT 16-leaf tiles.  zA/zB hold the lane's raw fp32 fragments.
// On return acc[T][r] = ||z - alpha||^2 for frame (lane&15), leaf 16T + 4(lane>>4) + r.
// ------------------------------------------------------------------------------------------
template <int DP>
__device__ __forceinline__ void load_ztile(const float* __restrict__ Z, int ldz, int64_t row0, int64_t N,
                                           int lane, float4 (&zA)[DP / 32], float4 (&zB)[DP / 32]);

// next_row0 >= 0: once zA/zB are converted they are reloaded with the rows of that later tile
// (software prefetch: the loads are in flight during this tile's MFMAs and epilogue).
struct NoLoads {
  __device__ __forceinline__ void operator()() const {}
};

// more_loads(): further global loads of the caller, issued together with the next tile's Z rows
// (after this tile's rows have been consumed, before the MFMA block that hides their latency).
template <int DP, int NT, class MoreLoads = NoLoads, bool ALWAYS_NEXT = false>
__device__ __forceinline__ void tile_distances(const float* __restrict__ Z, int ldz, int64_t row0,
                                               int64_t N, float4 (&zA)[DP / 32],
                                               float4 (&zB)[DP / 32], int64_t next_row0,
                                               const LeafOperands& lo_,
                                               int lane, int tiles_valid, int lvalid,
                                               f32x4 (&acc)[NT], MoreLoads more_loads = MoreLoads()) {
  constexpr int S = DP / 32;
  const int kg = lane >> 4;
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
  f16x8 bh[S], bl[S];
#pragma unroll
  for (int s = 0; s < S; ++s) split8(zA[s], zB[s], sz, bh[s], bl[s]);
  // Order matters for the vmcnt bookkeeping (in-order completion): the caller's loads for THIS tile
  // (prior entries, addressed through a video index that arrived a tile ago) are issued first, the
  // next tile's Z rows after them.  The other way round the wait for the video index is a
  // vmcnt(0) right behind the Z prefetch -- a full HBM round trip exposed on every tile.
#ifdef BNPVI_PRIOR_AFTER_Z
  if (next_row0 >= 0) load_ztile<DP>(Z, ldz, next_row0, N, lane, zA, zB);
  more_loads();
#else
  more_loads();
  __builtin_amdgcn_sched_barrier(0);
  // ALWAYS_NEXT: the prefetch is unconditional (the caller passes a valid row), so the loop body
  // is straight-line code and the compiler's vmcnt counts are exact, not a merge over branches
  if (ALWAYS_NEXT || next_row0 >= 0) load_ztile<DP>(Z, ldz, next_row0, N, lane, zA, zB);
#endif

  // every tile's MFMAs run unconditionally (operands of tiles past the padded leaves are zero)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int T = 0; T < NT; ++T) {
#ifdef BNPVI_ABLATE_NOMFMA
      acc[T][s & 3] = (s ? acc[T][s & 3] : 0.f) + (float)bh[s][0] + (float)bl[s][1];
#else
      const f16x8 ah = lo_.a[((T * S + s) * 2 + 0) * 64 + lane];
      const f16x8 al = lo_.a[((T * S + s) * 2 + 1) * 64 + lane];
      acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[s], s ? acc[T] : zero4, 0, 0, 0);
      acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[s], acc[T], 0, 0, 0);
      acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[s], acc[T], 0, 0, 0);
#endif
    }
  }
  // d^2 in place.  Cancellation check: an entry below refine_threshold(||z||^2+||a||^2) cannot be
  // trusted; the common case is screened with ONE compare per lane (min d^2 against the bound
  // for the largest ||a||^2), the per-entry test runs only for waves that fail the screen.
  const float unscale = -2.f * pow2f(-kz) * lo_.inv_sa;
  // pairs of entries on the packed fp32 pipe: d^2 = max(unscale * acc + (||z||^2 + ||a||^2), 0)
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
    asm volatile("" ::: "memory");        // nothing of the rare path is speculated above the branch
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      if (T < tiles_valid) {
        const float4 aa4 = *reinterpret_cast<const float4*>(lo_.aa + kLT * T + 4 * kg);
        const float aav[4] = {aa4.x, aa4.y, aa4.z, aa4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sn = zz + aav[r];
          if ((acc[T][r] < refine_threshold(sn)) && (kLT * T + 4 * kg + r < lvalid))
            pend |= 1u << ((T * 4 + r) & 31);
        }
      }
    }
  }
  // rare path: per pass every 16-lane row group (one kg: 16 frames x its leaf quads) takes ONE of its
  // flagged (frame, leaf) pairs and recomputes it by direct fp32 differences, the group's lanes
  // splitting the two rows -- four pairs per L2 round trip instead of one whole-wave round trip per
  // pair.  It matters for tight clusters far from the origin, which flag every frame's own leaf
  // (profiles/r02_notes.md: 2.3 ms -> 0.93 ms at config 2's shape with ~10 flagged leaves per frame);
  // a lane-parallel version (one pair per LANE and pass) was 20 % faster there but cost the common
  // path 8 % through register pressure, this one costs it nothing.
  // (NT > 8: the 32-bit mask aliases tiles T and T+8; every aliased slot is then patched with
  //  its own exact value, which is harmless.)
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
#pragma unroll
      for (int Tb = 0; Tb < NT; Tb += 8) {
        const int Tq = Tb + (idx >> 2);
        if (Tq < tiles_valid) {
          const float p = row16_direct_dist2<DP>(Z + ns * ldz, lo_.a32 + (kLT * Tq + 4 * kg + (idx & 3)) * DP, lane);
          if ((lane & 15) == srcj) {
#pragma unroll
            for (int T = Tb; T < (Tb + 8 < NT ? Tb + 8 : NT); ++T)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (idx == (T - Tb) * 4 + r) acc[T][r] = p;
          }
        }
      }
      if ((lane & 15) == srcj) pend &= pend - 1u;
    }
  }
}

template <int DP>
__device__ __forceinline__ void load_ztile(const float* __restrict__ Z, int ldz, int64_t row0, int64_t N,
                                           int lane, float4 (&zA)[DP / 32], float4 (&zB)[DP / 32]) {
  int64_t n = row0 + (lane & 15);
  if (n >= N) n = N - 1;
  const float* zrow = Z + n * ldz + 4 * (lane >> 4);
#pragma unroll
  for (int s = 0; s < DP / 32; ++s) {
#ifdef BNPVI_ABLATE_NOZLOAD
    zA[s] = make_float4(0.01f * lane, 0.02f * s, 0.5f, -0.25f);
    zB[s] = make_float4(0.03f * lane, 0.01f * s, 0.25f, -0.5f);
    (void)zrow;
#else
    zA[s] = *reinterpret_cast<const float4*>(zrow + 32 * s);
    zB[s] = *reinterpret_cast<const float4*>(zrow + 32 * s + 16);
#endif
  }
}

// ------------------------------------------------------------------------------------------
// Fused regime: the whole row of logits of a frame stays in registers (4*NT per lane), phi is
// written once (algorithmic traffic: read Z once, write phi once).  NT = 16-leaf tiles.
// A wave keeps PF tiles of Z in flight (PF register sets, each reloaded right after its
// conversion to f16 fragments).
// ------------------------------------------------------------------------------------------
struct FusedParams {
  const float* Z;
  int ldz;
  const int32_t* vid;
  int64_t N;
  int L, Lp;
  const float* prior;
  float half_sz;
  float* phi;
  float* logphi;
  int32_t* assign;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// phi is written once and next read by another kernel, after Z has gone by again: stored with the
// non-temporal policy so that it does not push the Z rows out of the memory-side cache
// (BNPVI_PHI_STORE_PLAIN rebuilds the default-policy variant for A/B runs).
__device__ __forceinline__ void store_phi4(float* dst, const f32x4& v) {
#ifdef BNPVI_PHI_STORE_PLAIN
  *reinterpret_cast<f32x4*>(dst) = v;
#else
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
#endif
}

// Generic epilogue of a tile (hard mode, or log phi / arg-max asked for): natural-log logits,
// arg-max tracking, phi = exp(log phi).
template <int NT, bool HARD, bool STATS>
__device__ __forceinline__ void tile_epilogue_generic(const FusedParams& p, int64_t n, bool ok, int lane,
                                                      int tiles_valid, const float* prow, f32x4 (&d2)[NT],
                                                      float (&s0)[STATS ? NT : 1][4],
                                                      float (&s2)[STATS ? NT : 1][4]) {
  const int kg = lane >> 4;
  f32x4 lg[NT];
  float m = -INFINITY;
  int best = 0x7fffffff;
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool tv = T < tiles_valid;
    if (!HARD && tv && prow) p4 = *reinterpret_cast<const float4*>(prow + kLT * T + 4 * kg);
    const float pr[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int leaf = kLT * T + 4 * kg + r;
      float v;
      if (HARD)
        v = leaf < p.L ? -d2[T][r] : -INFINITY;
      else      // the prior table is kept in log2 units
        v = tv ? fmaf(-p.half_sz, __builtin_amdgcn_sqrtf(d2[T][r]), pr[r] * kLn2) : -INFINITY;
      lg[T][r] = v;
      if (v > m) {
        m = v;
        best = leaf;
      }
    }
  }
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {
    const float om = __shfl_xor(m, off);
    const int ob = __shfl_xor(best, off);
    if (om > m || (om == m && ob < best)) {
      m = om;
      best = ob;
    }
  }
  float lse = 0.f;
  if (!HARD) {
    float ssum = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = lg[T][r] - m;
        lg[T][r] = x;
        ssum += __builtin_amdgcn_exp2f(x * kLog2e);
      }
    ssum += __shfl_xor(ssum, 16);
    ssum += __shfl_xor(ssum, 32);
    lse = __builtin_amdgcn_logf(ssum) * kLn2;
  }
  float* prow_out = p.phi + n * p.Lp;
  float* lrow_out = p.logphi ? p.logphi + n * p.Lp : nullptr;
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    if (T < tiles_valid) {
      float pv[4], lv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (HARD) {
          const bool hit = (kLT * T + 4 * kg + r) == best;
          pv[r] = hit ? 1.f : 0.f;
          lv[r] = hit ? 0.f : -INFINITY;
        } else {
          lv[r] = lg[T][r] - lse;
          pv[r] = __builtin_amdgcn_exp2f(lv[r] * kLog2e);
        }
        if (STATS && ok) {
          s0[T][r] += pv[r];
          s2[T][r] = fmaf(pv[r], d2[T][r], s2[T][r]);
        }
      }
      if (ok) {
        *reinterpret_cast<float4*>(prow_out + kLT * T + 4 * kg) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        if (lrow_out)
          *reinterpret_cast<float4*>(lrow_out + kLT * T + 4 * kg) = make_float4(lv[0], lv[1], lv[2], lv[3]);
      }
    }
  }
  if (p.assign && ok && kg == 0) p.assign[n] = best;
}

// Fast epilogue (soft responsibilities only, the production path of compute_phi): log2-domain
// logits straight from the log2-unit prior table, one exponential per entry, phi = e / sum,
// pairs of entries on the packed fp32 pipe, one store block per frame.
template <int NT, int TV, bool STATS, bool FULL = false>
__device__ __forceinline__ void tile_epilogue_soft(const FusedParams& p, int64_t n, bool ok, int lane,
                                                   int tiles_valid, const float* prow,
                                                   const float4 (&pre)[NT] /* prior entries, TV > 0 only */,
                                                   f32x4 (&d2)[NT], float (&s0)[STATS ? NT : 1][4],
                                                   float (&s2)[STATS ? NT : 1][4]) {
  const int kg = lane >> 4;
  const float c2 = -p.half_sz * kLog2e;
  f32x4 lg[NT];
  float m = -INFINITY;
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    float4 p4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (TV > 0)
      p4 = pre[T];
    else if (T < tiles_valid)
      p4 = *reinterpret_cast<const float4*>(prow + kLT * T + 4 * kg);
    lg[T][0] = fmaf(c2, __builtin_amdgcn_sqrtf(d2[T][0]), p4.x);
    lg[T][1] = fmaf(c2, __builtin_amdgcn_sqrtf(d2[T][1]), p4.y);
    lg[T][2] = fmaf(c2, __builtin_amdgcn_sqrtf(d2[T][2]), p4.z);
    lg[T][3] = fmaf(c2, __builtin_amdgcn_sqrtf(d2[T][3]), p4.w);
    m = fmaxf(fmaxf(m, lg[T][0]), lg[T][1]);
    m = fmaxf(fmaxf(m, lg[T][2]), lg[T][3]);
  }
  m = cols4_max(m);
  f32x2 ss = {0.f, 0.f};
  const f32x2 m2 = {m, m};
#pragma unroll
  for (int T = 0; T < NT; ++T) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x2 x = (f32x2){lg[T][2 * h], lg[T][2 * h + 1]} - m2;
      const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
      ss += e;
      lg[T][2 * h] = e[0];
      lg[T][2 * h + 1] = e[1];
    }
  }
  const float ssum = cols4_sum(ss[0] + ss[1]);
  // lanes past the last frame get phi = 0: nothing enters the statistics, nothing is stored
  const float inv = (FULL || ok) ? __builtin_amdgcn_rcpf(ssum) : 0.f;
  const f32x2 inv2 = {inv, inv};
#pragma unroll
  for (int T = 0; T < NT; ++T) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x2 pv = (f32x2){lg[T][2 * h], lg[T][2 * h + 1]} * inv2;
      lg[T][2 * h] = pv[0];
      lg[T][2 * h + 1] = pv[1];
      if (STATS) {
        f32x2 a0 = {s0[T][2 * h], s0[T][2 * h + 1]};
        f32x2 a2 = {s2[T][2 * h], s2[T][2 * h + 1]};
        const f32x2 dd = {d2[T][2 * h], d2[T][2 * h + 1]};
        a0 += pv;
        a2 += pv * dd;
        s0[T][2 * h] = a0[0];
        s0[T][2 * h + 1] = a0[1];
        s2[T][2 * h] = a2[0];
        s2[T][2 * h + 1] = a2[1];
      }
    }
  }
#if !defined(BNPVI_ABLATE_NOSTORE) && !defined(BNPVI_PHI_STORE_64B)
  if (FULL) {
    // Whole 128-byte lines per store instruction.  A lane holds 16 B of its frame per leaf tile, so the
    // natural store (below) writes 16 frames x 64 B per instruction and every line is completed by a
    // second one -- with the non-temporal policy that costs a copy-shaped kernel 13 us per 256 MB
    // (tools/ubench/stream_patterns: 91.0 against 77.7 us).  Instead the two halves of the frame columns
    // trade tiles (DPP rotation by 8 inside the 16-lane row): frames 0-7 go out with one instruction
    // (lanes j < 8: their tile T, lanes j >= 8: tile T+1 of frame j-8), frames 8-15 with the next.
    const int j = lane & 15;
    float* base = p.phi + (n - j + (j & 7)) * p.Lp + (j >> 3) * kLT + 4 * kg;
#pragma unroll
    for (int T = 0; T + 1 < NT; T += 2) {
      if (TV > 0 || T < tiles_valid) {
        f32x4 va, vb;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int a = (int)__float_as_uint(lg[T][c]), b = (int)__float_as_uint(lg[T + 1][c]);
          va[c] = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xc, false));
          vb[c] = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false));
        }
        store_phi4(base + kLT * T, va);
        store_phi4(base + 8 * p.Lp + kLT * T, vb);
      }
    }
    return;
  }
#endif
#ifdef BNPVI_ABLATE_NOSTORE
  if (ok && lg[0][0] + lg[0][1] == -123.f) {
#else
  if (FULL || ok) {
#endif
    float* prow_out = p.phi + n * p.Lp + 4 * kg;
#pragma unroll
    for (int T = 0; T < NT; ++T)
      if (TV > 0 || T < tiles_valid)
        store_phi4(prow_out + kLT * T, lg[T]);
  }
}

// FULL: every frame of the tile exists, a next tile is always prefetched (next_tile valid) and only
// soft responsibilities are asked for -- the body is then straight-line code (see tile_distances).
template <int DP, int NT, bool HARD, bool STATS, bool FULL = false>
__device__ __forceinline__ void fused_tile(const FusedParams& p, const LeafOperands& lo_, int64_t tile,
                                           int64_t next_tile, int64_t ntiles, int lane, int tiles_valid,
                                           float4 (&zA)[DP / 32], float4 (&zB)[DP / 32],
                                           int& vcur /* in: video of this tile's frame (lane & 15);
                                                        out: the same for next_tile */,
                                           float (&s0)[STATS ? NT : 1][4], float (&s2)[STATS ? NT : 1][4]) {
  // NT <= 4 is only dispatched with every tile valid (Lp = 16 NT): compile-time tile count there
  constexpr int TV = NT <= 4 ? NT : 0;
  const int j = lane & 15;
  const int64_t n = tile * kFT + j;
  const bool ok = n < p.N;
#ifdef BNPVI_ABLATE_NOPRIOR
  const float* prow = p.prior;
#else
  const float* prow = HARD ? nullptr : p.prior + (int64_t)vcur * p.Lp;
#endif
  const bool soft_only = FULL || !(HARD || p.logphi != nullptr || p.assign != nullptr);   // wave-uniform
  // The video index came with the previous tile's prefetch, so the prior entries are requested
  // BEFORE the MFMA block (with the next tile's Z rows and video index): their L2 round trip
  // runs under the MFMAs instead of in front of the epilogue.
  float4 pre[NT];
  auto more_loads = [&]() {
    if (TV > 0 && soft_only) {
#pragma unroll
#ifdef BNPVI_ABLATE_NOPRIORLOAD   // development only: no prior loads at all (results are garbage)
      for (int T = 0; T < NT; ++T) pre[T] = make_float4(-1.f * T, -2.f, -3.f, -0.5f * lane);
#else
      for (int T = 0; T < NT; ++T) pre[T] = *reinterpret_cast<const float4*>(prow + kLT * T + 4 * (lane >> 4));
#endif
    }
    if (FULL || (!HARD && next_tile < ntiles)) {
      const int64_t n2 = next_tile * kFT + j;
      vcur = p.vid[n2 < p.N ? n2 : p.N - 1];
    }
  };
  f32x4 d2[NT];
  tile_distances<DP, NT, decltype(more_loads), FULL>(
      p.Z, p.ldz, tile * kFT, p.N, zA, zB, (FULL || next_tile < ntiles) ? next_tile * kFT : -1, lo_, lane,
      tiles_valid, p.L, d2, more_loads);
  if (FULL)
    tile_epilogue_soft<NT, TV, STATS, true>(p, n, true, lane, tiles_valid, prow, pre, d2, s0, s2);
  else if (!soft_only)
    tile_epilogue_generic<NT, HARD, STATS>(p, n, ok, lane, tiles_valid, prow, d2, s0, s2);
  else
    tile_epilogue_soft<NT, TV, STATS>(p, n, ok, lane, tiles_valid, prow, pre, d2, s0, s2);
}

#ifndef BNPVI_FUSED_MINW3_DP
#define BNPVI_FUSED_MINW3_DP 128
#endif
#ifndef BNPVI_FUSED_MINW_SMALL
#define BNPVI_FUSED_MINW_SMALL 3
#endif
// waves per SIMD the fused kernel is compiled for (register budget 512 / that many per lane).  Up to four leaf tiles
// three waves also at D = 128 (round 3: with statistics the kernel took 201 registers and ran two; at 168 it keeps
// 11 dwords in scratch and is 13 % faster, 428 -> 374 us per 2M frames at L = 64) and two at D = 256.
constexpr int fused_min_waves(int DP, int NT) {
  return (NT <= 4 && DP <= BNPVI_FUSED_MINW3_DP) ? BNPVI_FUSED_MINW_SMALL : (NT <= 8 && DP <= 128) ? 2 : (NT <= 4) ? 2 : 1;
}

template <int DP, int NT, int PF, bool HARD, bool STATS>
#ifdef BNPVI_NO_MINW
__global__ __launch_bounds__(kEWaves * 64) void estep_fused_kernel(
#else
__global__ __launch_bounds__(kEWaves * 64, fused_min_waves(DP, NT)) void estep_fused_kernel(
#endif
    const float* __restrict__ Z, int ldz, const int32_t* __restrict__ vid, int64_t N, int L, int Lp,
    const f16x8* __restrict__ apack, const float* __restrict__ aavec, const float* __restrict__ a32,
    const float* __restrict__ scal /* [0] = 2^-ka */, const float* __restrict__ prior, float half_sz,
    float* __restrict__ phi, float* __restrict__ logphi, int32_t* __restrict__ assign,
    double* __restrict__ eslab) {
  static_assert(PF == 1 || PF == 2, "one or two tiles in flight per wave");
  constexpr int S = DP / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f16x8* a_lds = reinterpret_cast<f16x8*>(smem_raw);                  // [NT*S*2*64]
  float* aa_lds = reinterpret_cast<float*>(a_lds + NT * S * 2 * 64);  // [16*NT]
  double* st_lds = reinterpret_cast<double*>(aa_lds + kLT * NT);      // [16*NT][2]
  const int tiles_valid = Lp / kLT;

  for (int e = threadIdx.x; e < NT * S * 2 * 64; e += blockDim.x)
    a_lds[e] = e < tiles_valid * S * 2 * 64 ? apack[e] : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = threadIdx.x; e < kLT * NT; e += blockDim.x) aa_lds[e] = e < Lp ? aavec[e] : 0.f;
  if (STATS)
    for (int e = threadIdx.x; e < 2 * kLT * NT; e += blockDim.x) st_lds[e] = 0.0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; said so, the tile counters and loop control live in scalar registers
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int kg = lane >> 4;
  const int64_t ntiles = (N + kFT - 1) / kFT;
  const int64_t stride = (int64_t)gridDim.x * kEWaves;
  const LeafOperands lo_{a_lds, aa_lds, a32, scal[0], scal[2]};
  const FusedParams p{Z, ldz, vid, N, L, Lp, prior, half_sz, phi, logphi, assign};

  float s0[STATS ? NT : 1][4], s2[STATS ? NT : 1][4];
  if (STATS) {
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) s0[T][r] = s2[T][r] = 0.f;
  }

  int64_t tile = (int64_t)blockIdx.x * kEWaves + wave;
  float4 zA0[S], zB0[S], zA1[PF == 2 ? S : 1], zB1[PF == 2 ? S : 1];
  int vcur0 = 0, vcur1 = 0;
  if (tile < ntiles) {
    load_ztile<DP>(Z, ldz, tile * kFT, N, lane, zA0, zB0);
    if (!HARD) vcur0 = vid[tile * kFT + j < N ? tile * kFT + j : N - 1];
  }
  // Production path (soft responsibilities only, small trees): the whole tiles of the wave run in a
  // loop whose body has no branch around a memory instruction -- the prefetch of the next tile is
  // unconditional (the last iteration re-requests its own rows) -- so every s_waitcnt vmcnt the
  // compiler places is exact: the wait for this tile's rows leaves the previous tile's phi stores
  // in flight, the wait for the prior entries leaves the next tile's rows in flight.  With
  // conditional loads / stores in the body the counts are merged over the branches and degrade to
  // vmcnt(0): one exposed HBM round trip per tile (rocprofv3: 54 % of wave cycles in s_waitcnt).
  if constexpr (!HARD && PF == 1 && NT <= 4) {
    if (logphi == nullptr && assign == nullptr) {
      const int64_t nfull = N / kFT;
      if (tile < nfull) {
        // clean entry state: nothing outstanding when the loop is entered
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));        // vmcnt(0)
        while (tile < nfull) {
          const int64_t nxt = tile + stride < nfull ? tile + stride : tile;
          fused_tile<DP, NT, false, STATS, true>(p, lo_, tile, nxt, ntiles, lane, tiles_valid, zA0, zB0, vcur0, s0,
                                                 s2);
          tile += stride;
        }
        if (tile < ntiles) {            // the ragged last tile of the data set: fresh loads, generic body
          load_ztile<DP>(Z, ldz, tile * kFT, N, lane, zA0, zB0);
          vcur0 = vid[tile * kFT + j < N ? tile * kFT + j : N - 1];
        }
      }
    }
  }
  if constexpr (PF == 2) {
    if (tile + stride < ntiles) {
      load_ztile<DP>(Z, ldz, (tile + stride) * kFT, N, lane, zA1, zB1);
      if (!HARD) vcur1 = vid[(tile + stride) * kFT + j < N ? (tile + stride) * kFT + j : N - 1];
    }
  }
  while (tile < ntiles) {
    fused_tile<DP, NT, HARD, STATS>(p, lo_, tile, tile + PF * stride, ntiles, lane, tiles_valid, zA0, zB0,
                                    vcur0, s0, s2);
    tile += stride;
    if constexpr (PF == 2) {
      if (tile >= ntiles) break;
      // keep the two copies of the body apart: interleaved they need twice the registers
      __builtin_amdgcn_sched_barrier(0);
      fused_tile<DP, NT, HARD, STATS>(p, lo_, tile, tile + PF * stride, ntiles, lane, tiles_valid, zA1,
                                      zB1, vcur1, s0, s2);
      tile += stride;
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (STATS) {
    // reduce the per-lane leaf accumulators over the 16 frame lanes of each lane group, then
    // fp64 accumulation in LDS, one slab per workgroup (summed in fixed order afterwards)
#pragma unroll
    for (int T = 0; T < NT; ++T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s0[T][r]), b = row16_sum(s2[T][r]);      // DPP row rotations, no LDS
        if (j == 0) {
          const int leaf = kLT * T + 4 * kg + r;
          atomicAdd(&st_lds[2 * leaf], (double)a);
          atomicAdd(&st_lds[2 * leaf + 1], (double)b);
        }
      }
    }
    __syncthreads();
    double* out = eslab + (int64_t)blockIdx.x * 2 * L;
    for (int e = threadIdx.x; e < 2 * L; e += blockDim.x) out[e] = st_lds[e];
  }
}

// ------------------------------------------------------------------------------------------
// Streamed regime (trees too large for the fused kernel), pass A: raw logits for one 16-frame
// tile x one chunk of NT 16-leaf tiles, written into phi; part[n][chunk] = (max,
// sum exp(logit-max)) in soft mode, (max of -d^2, arg-max leaf as float bits) in hard mode.
// A workgroup keeps ONE leaf chunk (blockIdx.y) for all its frame tiles.  WAVES > 0: the chunk's
// packed operands (NT * DP * 64 B, 128 KiB at DP = 256) are staged in LDS once per workgroup and
// shared by its WAVES waves; WAVES = 0: four waves read them straight from L2 for every tile
// (kept for operand chunks that do not fit the LDS).
// ------------------------------------------------------------------------------------------
template <int DP, int NT, bool HARD, int WAVES = 0>
__global__ __launch_bounds__((WAVES > 0 ? WAVES : kEWaves) * 64) void estep_logits_kernel(
    const float* __restrict__ Z, int ldz, const int32_t* __restrict__ vid, int64_t N, int L, int Lp,
    const f16x8* __restrict__ apack, const float* __restrict__ aavec, const float* __restrict__ a32,
    const float* __restrict__ scal, const float* __restrict__ prior, float half_sz,
    float* __restrict__ phi, float2* __restrict__ part, int nchunk) {
  constexpr int S = DP / 32;
  constexpr int kWaves = WAVES > 0 ? WAVES : kEWaves;
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; said so, the tile counters and loop control live in scalar registers
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int kg = lane >> 4;
  const int chunk = blockIdx.y;
  const int T0 = chunk * NT;                      // first 16-leaf tile of this chunk
  int tiles_valid = Lp / kLT - T0;
  tiles_valid = tiles_valid > NT ? NT : tiles_valid;
  const int64_t ntiles = (N + kFT - 1) / kFT;
  const f16x8* a_src = apack + (int64_t)T0 * S * 2 * 64;
  const float* aa_src = aavec + kLT * T0;
  if constexpr (WAVES > 0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16x8* a_lds = reinterpret_cast<f16x8*>(smem_raw);                  // [NT*S*2*64]
    float* aa_lds = reinterpret_cast<float*>(a_lds + NT * S * 2 * 64);  // [16*NT]
    for (int e = threadIdx.x; e < NT * S * 2 * 64; e += blockDim.x)
      a_lds[e] = e < tiles_valid * S * 2 * 64 ? a_src[e] : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = threadIdx.x; e < kLT * NT; e += blockDim.x) aa_lds[e] = e < kLT * tiles_valid ? aa_src[e] : 0.f;
    __syncthreads();
    a_src = a_lds;
    aa_src = aa_lds;
  }
  LeafOperands lo_{a_src, aa_src, a32 + (int64_t)kLT * T0 * DP, scal[0], scal[2]};

  // the next tile's Z rows are requested as soon as this tile's are converted (in flight during the
  // MFMA block, the epilogue and the stores)
  const int64_t tstride = (int64_t)gridDim.x * kWaves;
  float4 zA[S], zB[S];
  {
    const int64_t first = (int64_t)blockIdx.x * kWaves + wave;
    if (first < ntiles) load_ztile<DP>(Z, ldz, first * kFT, N, lane, zA, zB);
  }
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < ntiles; tile += tstride) {
    const int64_t n = tile * kFT + j;
    const bool ok = n < N;
    const int64_t nn = ok ? n : N - 1;
    f32x4 d2[NT];
#ifdef BNPVI_LOGITS_NO_PREFETCH
    if (tile != (int64_t)blockIdx.x * kWaves + wave) load_ztile<DP>(Z, ldz, tile * kFT, N, lane, zA, zB);
    tile_distances<DP, NT>(Z, ldz, tile * kFT, N, zA, zB, -1, lo_, lane, tiles_valid, L - kLT * T0, d2);
#else
    tile_distances<DP, NT>(Z, ldz, tile * kFT, N, zA, zB, tile + tstride < ntiles ? (tile + tstride) * kFT : -1, lo_,
                           lane, tiles_valid, L - kLT * T0, d2);
#endif
#ifdef BNPVI_ABLATE_A_NOPRIOR
    const float* prow = nullptr;
#else
    const float* prow = HARD ? nullptr : prior + (int64_t)vid[nn] * Lp + kLT * T0;
#endif
    float m = -INFINITY;
    int best = 0x7fffffff;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const bool tv = T < tiles_valid;
      float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
#ifndef BNPVI_ABLATE_A_NOPRIOR
      if (!HARD && tv) p4 = *reinterpret_cast<const float4*>(prow + kLT * T + 4 * kg);
#endif
      const float pr[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int leaf = kLT * (T0 + T) + 4 * kg + r;
        float v;
        if (HARD)
          v = leaf < L ? -d2[T][r] : -INFINITY;
        else
#ifdef BNPVI_ABLATE_A_NOEPI
          v = tv ? fmaf(-half_sz, d2[T][r], pr[r] * kLn2) : -INFINITY;
#else
          v = tv ? fmaf(-half_sz, __builtin_amdgcn_sqrtf(d2[T][r]), pr[r] * kLn2) : -INFINITY;
#endif
        d2[T][r] = v;
        if (v > m) {
          m = v;
          best = leaf;
        }
      }
    }
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      const float om = __shfl_xor(m, off);
      const int ob = __shfl_xor(best, off);
      if (om > m || (om == m && ob < best)) {
        m = om;
        best = ob;
      }
    }
    float second;
    if (HARD) {
      second = __int_as_float(best);
    } else {
      float ssum = 0.f;
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#ifdef BNPVI_ABLATE_A_NOEPI
          ssum += (m == -INFINITY) ? 0.f : (d2[T][r] - m) * kLog2e;
#else
          ssum += (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((d2[T][r] - m) * kLog2e);
#endif
      ssum += __shfl_xor(ssum, 16);
      ssum += __shfl_xor(ssum, 32);
      second = ssum;
    }
#ifdef BNPVI_ABLATE_A_NOSTORE
    if (m == -123.f)
#endif
#ifndef BNPVI_LOGITS_STORE_64B
    if (tile * kFT + kFT <= N) {
      // whole 128-byte lines per store instruction, non-temporal (the logits are next read by the normalising
      // pass, after ~L/128 times their size has gone by): pairs of leaf tiles are traded between the two halves
      // of the frame columns as in the fused kernel's epilogue (tiles_valid is even: Lp is a multiple of 32)
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
      if (kg == 0) part[n * nchunk + chunk] = make_float2(m, second);
    } else
#endif
    if (ok) {
      float* row = phi + n * Lp + kLT * T0;
#pragma unroll
      for (int T = 0; T < NT; ++T)
        if (T < tiles_valid)
          *reinterpret_cast<float4*>(row + kLT * T + 4 * kg) = make_float4(d2[T][0], d2[T][1], d2[T][2], d2[T][3]);
      if (kg == 0) part[n * nchunk + chunk] = make_float2(m, second);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Leaf operands of a small tree by ONE workgroup: fp32 rows, ||alpha||^2, max|alpha| (block
// reduction), barrier, then the scaled hi/lo f16 fragments in MFMA operand order (layout in
// estep.hip).  a32 is re-read after the barrier by other threads of the same workgroup.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void leaf_operands_block(
    const double* __restrict__ alpha, int D, int DP, const int32_t* __restrict__ node_of_leaf, int L, int Lp,
    float* a32, float* __restrict__ aavec, f16x8* __restrict__ apack, float* __restrict__ scal) {
  __shared__ float red[2][16];          // up to 16 waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float amax = 0.f, aamax = 0.f;
  for (int leaf = wave; leaf < Lp; leaf += nw) {
    const double* a = leaf < L ? alpha + (int64_t)node_of_leaf[leaf] * D : nullptr;
    double aa = 0.0;
    for (int d = lane; d < DP; d += 64) {
      const float f = (a && d < D) ? (float)a[d] : 0.f;
      a32[(int64_t)leaf * DP + d] = f;
      aa += (double)f * (double)f;
      amax = fmaxf(amax, fabsf(f));
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) aa += __shfl_xor(aa, off);
    if (lane == 0) aavec[leaf] = (float)aa;
    aamax = fmaxf(aamax, (float)aa);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if (lane == 0) {
    red[0][wave] = amax;
    red[1][wave] = aamax;
  }
  __syncthreads();                                 // also orders the a32 rows for the readers below
  amax = 0.f;
  aamax = 0.f;
  for (int w = 0; w < nw; ++w) {
    amax = fmaxf(amax, red[0][w]);
    aamax = fmaxf(aamax, red[1][w]);
  }
  const int ka = scale_exponent_bits(__float_as_int(amax));
  if (threadIdx.x == 0) {
    scal[0] = pow2f(-ka);
    scal[1] = amax;
    scal[2] = aamax;
  }
  const int S = DP / 32;
  const float sa = pow2f(ka);
  for (int u = threadIdx.x; u < (Lp / kLT) * S * 2 * 64; u += blockDim.x) {
    const int ln = u & 63;
    const int part = (u >> 6) & 1;
    const int ts = u >> 7;
    const int s = ts % S, T = ts / S;
    const int leaf = kLT * T + (ln & 15);
    const int kg = ln >> 4;
    f16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int dim = 32 * s + 16 * (e >> 2) + 4 * kg + (e & 3);
      const float v = a32[(int64_t)leaf * DP + dim] * sa;
      const _Float16 h = (_Float16)v;
      out[e] = part == 0 ? h : (_Float16)(v - (float)h);
    }
    apack[u] = out;
  }
}


struct LeafWorkspace {
  int LpPack;
  f16x8* apack;
  float* a32;
  float* aavec;
  float* scal;
};
int leaf_workspace(bnpvi_ctx* ctx, const bnpvi_tree* t, int DP, LeafWorkspace* w);

// per-DP launchers (estep_inst.hip, compiled once per DP)
struct EstepArgs {
  bnpvi_ctx* ctx;
  const bnpvi_tree* tree;
  const float* Z;
  int ldz;
  const int32_t* vid;
  int64_t N;
  const f16x8* apack;
  const float* aavec;
  const float* a32;
  const float* scal;
  const float* prior;
  float half_sz;
  int mode;
  float* phi;
  float* logphi;
  int32_t* assign;
  double* estats;
  hipStream_t stream;
  bool s0_only;        // BNPVI_ESTEP_STATS_S0: estats[l][1] may be left 0
};

template <int DP> int estep_dispatch(const EstepArgs& a);

// shared (estep.hip)
int estep_finish_streamed(const EstepArgs& a, const float2* part, int nchunk);
int estep_reduce_slabs(const EstepArgs& a, const double* eslab, int nslab);
constexpr int kStreamNT = 8;      // 16-leaf tiles per chunk in the streamed regime (128 leaves)

}  // namespace bnpvi
