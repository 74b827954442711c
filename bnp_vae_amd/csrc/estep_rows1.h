// One-pass row E-step: soft responsibilities for trees of 64 < L <= 64 * NCH leaves (NCH up to 8) whose packed leaf
// operands fit the LDS, with the per-leaf mass S0' = sum_n phi[n,l] as its only statistic.  Replaces
// /root/reference/src/var_inf.py:313-357,383-393 like the other E-step kernels (estep_kernels.h, estep_rows.h)
// and feeds the two tests that read mean(phi): merge_nodes (:194) and the gate of split_nodes (:217).
//
// Why a kernel of its own.  The two-pass row kernel (estep_rows.h) computes every logit twice -- matrix products,
// distances, square roots and exponentials -- because a frame's row of logits must be complete before any of it
// can be normalised, and carries S2 = sum phi d^2 for every leaf although the reference looks at the stdev of a
// leaf only behind mean(phi) >= 1e-2 (src/var_inf.py:217-220), which at most 100 leaves can pass and none does on
// a balanced tree of hundreds of leaves.  It is bound by vector instruction issue at a third of the HBM roofline
// (profiles/r02_notes.md).  Here the logits of a wave's 16-frame tile stay in REGISTERS (16 * NCH per lane:
// 128 for 512 leaves, two waves per SIMD at 256 registers each), so the tile costs one pass of products,
// distances, square roots and ONE exponential per entry; S2 for the few gated leaves is a separate pass over
// their columns (bnpvi_gated_s2, tree_kernels.hip).
//
// Orientation as in estep_rows.h: A = Z fragment (row = frame), B = alpha fragment (col = leaf): a lane holds ONE
// leaf column (lane & 15) of a 16-leaf tile and the 4 frames 4 * (lane >> 4) + r.
//   phase 1  per 64-leaf chunk: prior entries requested, 3 f16 products per k-step (exact hi/lo split), squared
//            distances, cancellation screen + exact recomputation (estep_kernels.h), v = prior - c * sqrt(d2) kept,
//            running maximum per frame;
//   phase 2  row maximum over the 16 leaf lanes (DPP), e = exp2(v - max) overwrites v, row sums;
//   phase 3  per chunk: phi = e / sum, S0' partial sums (4 leaf tiles reduced over the 4 lane groups with three
//            permlane swaps, one fp64 LDS add per lane), whole-line non-temporal stores.
// The next tile's Z rows are requested between phase 1 and phase 2 (their registers are the f16 fragments' until
// then).  HBM traffic is the algorithmic minimum: Z read once, phi written once.
#pragma once

#include <type_traits>

#include "estep_rows.h"

namespace bnpvi {

constexpr int kR1Waves = 8;          // waves per workgroup: two per SIMD, 256 registers each

template <int DP>
struct Rows1Lds {
  static constexpr size_t bytes(int nchunk) {
    return (size_t)nchunk * kRChunk * ((DP / 32) * 2 * 64 * 16 + kLT * 4 + kLT * 8);
  }
  // most 64-leaf chunks whose logits (16 registers each) and f16 fragments fit two waves per SIMD and the LDS
  static constexpr int max_chunks() {
    int n = DP <= 64 ? 8 : DP <= 128 ? 4 : 2;
    while (n > 0 && bytes(n) > (size_t)kMaxLdsBytes) --n;
    return n;
  }
};

template <int DP, int NCH>
__global__ __launch_bounds__(kR1Waves * 64, 2) void estep_rows1_kernel(
    const float* __restrict__ Z, int ldz, const int32_t* __restrict__ vid, int64_t N, int L, int Lp,
    const f16x8* __restrict__ apack, const float* __restrict__ aavec,
    const float* __restrict__ a32, const float* __restrict__ scal, const float* __restrict__ prior, float half_sz,
    float* __restrict__ phi, double* __restrict__ eslab /* [workgroups][L][2] (S0', 0), or null */, int tiles_per_wg) {
  constexpr int S = DP / 32;
  constexpr int kCT = kRChunk;
  const bool STATS = eslab != nullptr;                        // uniform
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int NTp = NCH * kCT;
  f16x8* a_lds = reinterpret_cast<f16x8*>(smem_raw);                            // [NTp*S*2*64]
  float* aa_lds = reinterpret_cast<float*>(a_lds + (size_t)NTp * S * 2 * 64);   // [16*NTp]
  double* st_lds = reinterpret_cast<double*>(aa_lds + kLT * NTp);               // [16*NTp]
  const int tiles_valid = Lp / kLT;

  for (int e = threadIdx.x; e < NTp * S * 2 * 64; e += blockDim.x)
    a_lds[e] = e < tiles_valid * S * 2 * 64 ? apack[e] : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = threadIdx.x; e < kLT * NTp; e += blockDim.x) aa_lds[e] = e < Lp ? aavec[e] : 0.f;
  if (STATS)
    for (int e = threadIdx.x; e < kLT * NTp; e += blockDim.x) st_lds[e] = 0.0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15;
  const int kg = lane >> 4;
  const float inv_sa = scal[0];
  const float aamax = scal[2];
  const float c2 = -half_sz * kLog2e;
  // Lp is a multiple of 32 and NCH = ceil(Lp / 64): only the last chunk's tiles 2 and 3 can be padding
  const bool last_half = Lp < 64 * NCH;                       // uniform
  const float* aa_lane = aa_lds + c;                          // + 16 T: immediate offsets
  const f16x8* a_lane = a_lds + lane;

  const int64_t ntiles = (N + kFT - 1) / kFT;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wg;
  const int64_t t1 = (t0 + tiles_per_wg < ntiles) ? t0 + tiles_per_wg : ntiles;

  int64_t tile = t0 + wave;
  float4 zA[S], zB[S];
  if (tile < t1) load_ztile<DP>(Z, ldz, tile * kFT, N, lane, zA, zB);

  for (; tile < t1; tile += kR1Waves) {
    const int64_t row0 = tile * kFT;
    // ---- Z fragments of the tile: ||z||^2 and the f16 scale per frame (frame = lane & 15 here)
    float zz = 0.f, amax = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float q[8] = {zA[s].x, zA[s].y, zA[s].z, zA[s].w, zB[s].x, zB[s].y, zB[s].z, zB[s].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        zz = fmaf(q[e], q[e], zz);
        amax = fmaxf(amax, fabsf(q[e]));
      }
    }
    zz = cols4_sum(zz);
    amax = cols4_max(amax);
    const int kz = scale_exponent_bits(__float_as_int(amax));
    f16x8 zh[S], zl[S];
#pragma unroll
    for (int s = 0; s < S; ++s) split8(zA[s], zB[s], pow2f(kz), zh[s], zl[s]);

    // the lane's 4 frames (rows 4kg + r): ||z||^2, un-scaling factor, validity
    const float unscale = -2.f * pow2f(-kz) * inv_sa;
    float zzr[4], unr[4];
    bool okr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zzr[r] = __shfl(zz, 4 * kg + r);
      unr[r] = __shfl(unscale, 4 * kg + r);
      okr[r] = row0 + 4 * kg + r < N;
    }
    const int64_t nlast = row0 + kFT - 1 < N ? row0 + kFT - 1 : N - 1;
    const int vfirst = __builtin_amdgcn_readfirstlane(vid[row0]);
    const bool onevid = vfirst == __builtin_amdgcn_readfirstlane(vid[nlast]);   // wave-uniform: one prior row
    const bool full = row0 + kFT <= N;                                           // wave-uniform
    // prior entries of the lane's leaf column: one row pointer per frame (one video in the tile: a single row,
    // wave-uniform base); the leaf tile is an immediate offset
    const float* prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = row0 + 4 * kg + r;
      prow[r] = prior + (int64_t)(onevid ? vfirst : vid[n < N ? n : N - 1]) * Lp + c;
    }
    // tiles 2 and 3 of a half-filled last chunk re-read tiles 0 and 1 (overwritten with -inf below)
    const int last_off = kLT * (last_half ? (NCH - 1) * kCT : (NCH - 1) * kCT + 2);

    float v[NCH][kCT][4];                 // logits, then exp2(logit - max), of the lane's (leaf column, 4 frames)
    float mrun[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};

    // ---- phase 1
    auto phase1_chunk = [&](auto onevid_c, int ch) {
      // One video in the tile (the usual case, wave-uniform): one prior entry per leaf tile serves the lane's 4
      // frames.  The chunk's entries are requested first: their L2 round trip runs under the matrix products
      // (padded tiles read the last valid tile's address and are overwritten with -inf below).
      constexpr int kPR = decltype(onevid_c)::value ? 1 : 4;
      float prv[kCT][kPR];
#pragma unroll
      for (int t = 0; t < kCT; ++t) {
#pragma unroll
        for (int r = 0; r < kPR; ++r)
#ifdef BNPVI_R1_ABLATE_NOPRIOR
          prv[t][r] = -0.001f * (ch * kCT + t);
#else
          prv[t][r] = (ch == NCH - 1 && t >= 2) ? prow[r][last_off + kLT * (t - 2)] : prow[r][kLT * (ch * kCT + t)];
#endif
      }
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 acc[kCT];
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          const int T = ch * kCT + t;
          const f16x8 ah = a_lane[((T * S + s) * 2 + 0) * 64];
          const f16x8 al = a_lane[((T * S + s) * 2 + 1) * 64];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zl[s], ah, s ? acc[t] : zero4, 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[s], al, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[s], ah, acc[t], 0, 0, 0);
        }
      }
      // issue order: the first k-step's 8 operand reads, then every 3 products hide the next 2 reads -- also
      // bounds the operand registers in flight (all 16 reads up front would take 64)
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * kCT, 0);
#pragma unroll
      for (int g = 0; g < (S - 1) * kCT; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * kCT, 0);
      float d[kCT][4];
      float dmin[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
      for (int t = 0; t < kCT; ++t) {
        const float aa = aa_lane[kLT * (ch * kCT + t)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d[t][r] = fmaf(unr[r], acc[t][r], zzr[r] + aa);
          dmin[r] = fminf(dmin[r], d[t][r]);
        }
      }
      // cancellation screen (estep_kernels.h): an entry below refine_threshold(||z||^2 + ||a||^2) is recomputed by
      // direct fp32 differences; one compare per frame against the bound for the largest ||a||^2 first
      bool sus = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sn = zzr[r] + aamax;
        sus |= dmin[r] < refine_threshold(sn);
      }
      if (__builtin_expect(__ballot(sus) != 0ull, 0)) {
        asm volatile("" ::: "memory");      // nothing of the rare path is speculated above the branch
        unsigned pend = 0u;
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          // (padded leaves are not excluded: their operands are zero rows that exist, their prior is -inf)
          const float aa = aa_lane[kLT * (ch * kCT + t)];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sn = zzr[r] + aa;
            if (d[t][r] < refine_threshold(sn)) pend |= 1u << (t * 4 + r);
          }
        }
        // one flagged pair per 16-lane row group (4 frames x 16 leaf columns) and pass, the group's lanes
        // splitting the two rows (see tile_distances)
        while (true) {
          const unsigned long long b = __ballot(pend != 0u);
          if (b == 0ull) break;
          const unsigned g = (unsigned)(b >> (16 * kg)) & 0xffffu;
          if (g != 0u) {
            const int srcc = __ffs((int)g) - 1;                     // owner: leaf column of the group
            const unsigned pm = (unsigned)__shfl((int)pend, 16 * kg + srcc);
            const int idx = __ffs((int)pm) - 1;
            int64_t ns = row0 + 4 * kg + (idx & 3);
            if (ns >= N) ns = N - 1;
            const int leaf = kLT * (ch * kCT + (idx >> 2)) + srcc;
            const float p = row16_direct_dist2<DP>(Z + ns * ldz, a32 + (int64_t)leaf * DP, lane);
            if (c == srcc) {
#pragma unroll
              for (int t = 0; t < kCT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (idx == t * 4 + r) d[t][r] = p;
              pend &= pend - 1u;
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < kCT; ++t) {
        const bool tv = !(ch == NCH - 1 && t >= 2 && last_half);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = tv ? prv[t][kPR == 1 ? 0 : r] : -INFINITY;
          const float x = fmaf(c2, __builtin_amdgcn_sqrtf(__builtin_fabsf(d[t][r])), pr);
          v[ch][t][r] = x;
          mrun[r] = fmaxf(mrun[r], x);
        }
      }
      // the logits are COMPUTED here: without this pin on the running maxima the optimiser sinks the square roots
      // of every chunk below the last chunk's products (their first use is phase 2) and keeps all squared
      // distances and prior entries of the tile alive next to them
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(mrun[r]));
    };
    if (onevid) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        phase1_chunk(std::true_type{}, ch);
        __builtin_amdgcn_sched_barrier(0);       // chunks one after the other: bounded register use
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        phase1_chunk(std::false_type{}, ch);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // next tile's Z rows: in flight during phases 2 and 3
    if (tile + kR1Waves < t1) load_ztile<DP>(Z, ldz, (tile + kR1Waves) * kFT, N, lane, zA, zB);

    // ---- phase 2: row maximum, one exponential per entry, row sum
    float ssum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mrun[r] = row16_max(mrun[r]);              // finite: L >= 1
      ssum[r] = 0.f;
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      {
#pragma unroll
        for (int t = 0; t < kCT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(v[ch][t][r] - mrun[r]);
            v[ch][t][r] = e;
            ssum[r] += e;
          }
      }
      __builtin_amdgcn_sched_barrier(0);         // in place, chunk by chunk: no second copy of the tile in flight
    }
    float inv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float tot = row16_sum(ssum[r]);
      inv[r] = okr[r] ? __builtin_amdgcn_rcpf(tot) : 0.f;       // rows past N: phi = 0, not stored
    }

    // ---- phase 3: normalise, statistics, store
    // whole-line stores of tile pairs: row 4 (kg & ~1) + r resp. 4 (kg | 1) + r, leaf tile + (kg & 1);
    // per frame row one lane address, the leaf tile is an immediate offset
    // (computed here, behind an opaque copy of the lane group: hoisted above phase 1 the 12 addresses spill)
    float* tile_base = phi + row0 * Lp;
    double* st_lane = st_lds + lane;                            // leaf 64 ch + 16 kg + c = 64 ch + lane
    float *stA[4], *stB[4], *stP[4];
    int kgl = kg;
    asm volatile("" : "+v"(kgl));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      stP[r] = tile_base + (4 * kgl + r) * Lp + c;
      stA[r] = stP[r] + ((kgl & 1) ? kLT - 4 * Lp : 0);
      stB[r] = stP[r] + ((kgl & 1) ? kLT : 4 * Lp);
    }
    bool okA[4], okB[4];                                        // rows 4 (kg & ~1) + r, 4 (kg | 1) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      okA[r] = row0 + 4 * (kg & ~1) + r < N;
      okB[r] = row0 + 4 * (kg | 1) + r < N;
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      {
        float pv[kCT][4];
#pragma unroll
        for (int t = 0; t < kCT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[t][r] = v[ch][t][r] * inv[r];
        if (STATS) {
          // the lane's 4 frames in registers, then the 4 lane groups (frames 4kg..4kg+3 of the same leaf) for
          // the chunk's 4 leaf tiles at once: three permlane swaps leave tile t = kg's sum in lane group kg
          float a[kCT];
#pragma unroll
          for (int t = 0; t < kCT; ++t) a[t] = (pv[t][0] + pv[t][1]) + (pv[t][2] + pv[t][3]);
          u32x2 w = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[0]), __float_as_uint(a[2]), false, false);
          const float u02 = __uint_as_float(w[0]) + __uint_as_float(w[1]);   // lanes 0-31: tile 0, lanes 32-63: tile 2
          w = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[1]), __float_as_uint(a[3]), false, false);
          const float u13 = __uint_as_float(w[0]) + __uint_as_float(w[1]);   // tiles 1 | 3
          w = __builtin_amdgcn_permlane16_swap(__float_as_uint(u02), __float_as_uint(u13), false, false);
          const float tot = __uint_as_float(w[0]) + __uint_as_float(w[1]);   // lane group kg: tile kg
          atomicAdd(st_lane + kLT * kCT * ch, (double)tot);     // padded leaves: adds of 0 to slots nobody reads
        }
        // lane-group pairs trade two leaf tiles (v_permlane16_swap, in place) so that every store instruction
        // writes whole 128-byte lines: lanes of even kg carry tile t, of odd kg tile t + 1
#pragma unroll
        for (int t = 0; t < kCT; t += 2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(pv[t][r]),
                                                              __float_as_uint(pv[t + 1][r]), false, false);
            pv[t][r] = __uint_as_float(sw[0]);
            pv[t + 1][r] = __uint_as_float(sw[1]);
          }
#ifdef BNPVI_R1_ABLATE_NOSTORE
        if (pv[0][0] == -123.f)
#endif
        if (full) {
#pragma unroll
          for (int t = 0; t < kCT; t += 2) {
            const int T = ch * kCT + t;
            if (!(ch == NCH - 1 && t >= 2 && last_half)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
#ifdef BNPVI_R1_STORE_PLAIN
                stA[r][kLT * T] = pv[t][r];
                stB[r][kLT * T] = pv[t + 1][r];
#else
                __builtin_nontemporal_store(pv[t][r], stA[r] + kLT * T);
                __builtin_nontemporal_store(pv[t + 1][r], stB[r] + kLT * T);
#endif
              }
            }
          }
        } else {
          // ragged last tile of the data set: the same traded values, rows past N masked
#pragma unroll
          for (int t = 0; t < kCT; t += 2) {
            const int T = ch * kCT + t;
            if (!(ch == NCH - 1 && t >= 2 && last_half)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (okA[r]) stA[r][kLT * T] = pv[t][r];
                if (okB[r]) stB[r][kLT * T] = pv[t + 1][r];
              }
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (STATS) {
    __syncthreads();
    double* out = eslab + (int64_t)blockIdx.x * 2 * L;
    for (int e = threadIdx.x; e < L; e += blockDim.x) {
      out[2 * e] = st_lds[e];
      out[2 * e + 1] = 0.0;
    }
  }
}

}  // namespace bnpvi
