// Two-pass ("row") E-step for trees whose logit row does not fit in registers (64 < L, packed
// leaf operands <= 128 KiB), soft responsibilities.  Same math and the same packed operands as
// the fused kernel (estep_kernels.h); replaces /root/reference/src/var_inf.py:313-357,383-393.
//
// Orientation.  The MFMA operands are swapped: A = Z fragment (row = frame), B = alpha fragment
// (col = leaf), so the accumulator holds ONE LEAF per lane (column lane & 15) and 4 FRAMES per
// lane (rows 4*(lane>>4) + r).  Consequences:
//   * per-leaf statistics (S0' = sum phi, S2 = sum phi d^2) need one accumulator per leaf tile
//     instead of four per frame lane: they are summed over the lane's 4 frames in registers and
//     added to a workgroup-wide fp32 table in LDS once per (tile, leaf tile);
//   * the softmax runs over lanes and leaf tiles: per frame an ONLINE max / sum-exp over chunks
//     of 64 leaves (pass 1), combined over the 16 leaf lanes once per frame tile;
//   * pass 2 recomputes the chunk's logits (MFMAs on the f16 pipe are cheap) and writes
//     phi = exp2(logit - max) / sum through a wave-private LDS staging tile, so global stores
//     are whole 128-byte row segments, 16 B per lane.
// HBM traffic is the algorithmic minimum: Z read once, phi written once (no logits round trip).
#pragma once

#include "estep_kernels.h"

namespace bnpvi {

constexpr int kRWaves = 8;            // waves per workgroup
constexpr int kRChunk = 4;            // 16-leaf tiles per online-softmax chunk (64 leaves)
constexpr int kRStageW = 32;          // leaves staged per store round
constexpr int kRStageLd = kRStageW + 4;   // staging row stride in floats (bank-conflict-free)

template <int DP>
struct RowsLds {
  static size_t bytes(int NTp, bool stats) {
    return (size_t)NTp * (DP / 32) * 2 * 64 * 16 + (size_t)kLT * NTp * 4 +
           (stats ? (size_t)kLT * NTp * 8 : 0) + (size_t)kRWaves * kFT * kRStageLd * 4;
  }
};

// 16 lanes (one leaf column each, same lane>>4) -> all-reduce
__device__ __forceinline__ float row16_max(float v) {
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off);
  return v;
}

template <int DP, bool STATS>
__global__ __launch_bounds__(kRWaves * 64, 2) void estep_rows_kernel(
    const float* __restrict__ Z, int ldz, const int32_t* __restrict__ vid, int64_t N, int L, int Lp,
    int NTp /* 16-leaf tiles, padded to a multiple of kRChunk */, const f16x8* __restrict__ apack,
    const float* __restrict__ aavec, const float* __restrict__ a32, const float* __restrict__ scal,
    const float* __restrict__ prior, float half_sz, float* __restrict__ phi, float* __restrict__ logphi,
    double* __restrict__ eslab, int tiles_per_wg) {
  constexpr int S = DP / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f16x8* a_lds = reinterpret_cast<f16x8*>(smem_raw);                    // [NTp*S*2*64]
  float* aa_lds = reinterpret_cast<float*>(a_lds + (size_t)NTp * S * 2 * 64);   // [16*NTp]
  float* st_lds = aa_lds + kLT * NTp;                                   // [16*NTp][2] (STATS)
  float* stage_all = st_lds + (STATS ? 2 * kLT * NTp : 0);              // [waves][16][kRStageLd]
  const int tiles_valid = Lp / kLT;

  for (int e = threadIdx.x; e < NTp * S * 2 * 64; e += blockDim.x)
    a_lds[e] = e < tiles_valid * S * 2 * 64 ? apack[e] : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = threadIdx.x; e < kLT * NTp; e += blockDim.x) aa_lds[e] = e < Lp ? aavec[e] : 0.f;
  if (STATS)
    for (int e = threadIdx.x; e < 2 * kLT * NTp; e += blockDim.x) st_lds[e] = 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15;
  const int kg = lane >> 4;
  float* stage = stage_all + wave * (kFT * kRStageLd);
  const float inv_sa = scal[0];
  const float aamax = scal[2];
  const float c2 = -half_sz * kLog2e;
  const int nchunk = NTp / kRChunk;

  const int64_t ntiles = (N + kFT - 1) / kFT;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_wg;
  const int64_t t1 = (t0 + tiles_per_wg < ntiles) ? t0 + tiles_per_wg : ntiles;

  int64_t tile = t0 + wave;
  float4 zA[S], zB[S];
  if (tile < t1) load_ztile<DP>(Z, ldz, tile * kFT, N, lane, zA, zB);

  for (; tile < t1; tile += kRWaves) {
    const int64_t row0 = tile * kFT;
    // ---- Z fragments of the tile: ||z||^2 and the f16 scale per frame (frame = lane & 15 here)
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
    zz += __shfl_xor(zz, 16);
    zz += __shfl_xor(zz, 32);
    amax = fmaxf(amax, __shfl_xor(amax, 16));
    amax = fmaxf(amax, __shfl_xor(amax, 32));
    const int kz = scale_exponent_bits(__float_as_int(amax));
    f16x8 zh[S], zl[S];
#pragma unroll
    for (int s = 0; s < S; ++s) split8(zA[s], zB[s], pow2f(kz), zh[s], zl[s]);
    if (tile + kRWaves < t1) load_ztile<DP>(Z, ldz, (tile + kRWaves) * kFT, N, lane, zA, zB);

    // the lane's 4 frames (rows 4kg + r): ||z||^2, un-scaling factor, prior row, validity
    const float unscale = -2.f * pow2f(-kz) * inv_sa;
    float zzr[4], unr[4], thr[4];
    const float* prow[4];
    bool okr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zzr[r] = __shfl(zz, 4 * kg + r);
      unr[r] = __shfl(unscale, 4 * kg + r);
      const float sn = zzr[r] + aamax;
      thr[r] = kRefineRel * sn * sn;
      const int64_t n = row0 + 4 * kg + r;
      okr[r] = n < N;
      prow[r] = prior + (int64_t)vid[okr[r] ? n : N - 1] * Lp + c;
    }
    const int64_t nlast = row0 + kFT - 1 < N ? row0 + kFT - 1 : N - 1;
    const bool onevid = vid[row0] == vid[nlast];       // wave-uniform: one prior row for the tile

    // logits of one 64-leaf chunk for the lane's (leaf column, 4 frames): v[t][r] (log2 units)
    // and d2[t][r]
    auto chunk_logits = [&](int ch, f32x4 (&d2)[kRChunk], f32x4 (&v)[kRChunk]) {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 acc[kRChunk];
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int t = 0; t < kRChunk; ++t) {
          const int T = ch * kRChunk + t;
          const f16x8 ah = a_lds[((T * S + s) * 2 + 0) * 64 + lane];
          const f16x8 al = a_lds[((T * S + s) * 2 + 1) * 64 + lane];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zl[s], ah, s ? acc[t] : zero4, 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[s], al, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[s], ah, acc[t], 0, 0, 0);
        }
      }
      float pr[kRChunk][4];
#pragma unroll
      for (int t = 0; t < kRChunk; ++t) {
        const int T = ch * kRChunk + t;
        const bool tv = T < tiles_valid;
        if (onevid) {
          const float p = tv ? prow[0][kLT * T] : -INFINITY;
          pr[t][0] = pr[t][1] = pr[t][2] = pr[t][3] = p;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) pr[t][r] = tv ? prow[r][kLT * T] : -INFINITY;
        }
      }
      float dmin[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
      for (int t = 0; t < kRChunk; ++t) {
        const float aa = aa_lds[kLT * (ch * kRChunk + t) + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = fmaxf(fmaf(unr[r], acc[t][r], zzr[r] + aa), 0.f);
          d2[t][r] = d;
          dmin[r] = fminf(dmin[r], d);
        }
      }
      // cancellation screen (see estep_kernels.h): rare exact recomputation by direct differences
      const bool sus = (dmin[0] < thr[0]) | (dmin[1] < thr[1]) | (dmin[2] < thr[2]) | (dmin[3] < thr[3]);
      if (__ballot(sus) != 0ull) {
        unsigned pend = 0u;
#pragma unroll
        for (int t = 0; t < kRChunk; ++t) {
          const int leaf = kLT * (ch * kRChunk + t) + c;
          const float aa = aa_lds[leaf];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sn = zzr[r] + aa;
            if (d2[t][r] < kRefineRel * sn * sn && leaf < L) pend |= 1u << (t * 4 + r);
          }
        }
        while (true) {
          const unsigned long long b = __ballot(pend != 0u);
          if (b == 0ull) break;
          const int src = __ffsll((long long)b) - 1;
          const unsigned pm = __builtin_amdgcn_readlane(pend, src);
          const int idx = __ffs((int)pm) - 1;
          int64_t ns = row0 + 4 * (src >> 4) + (idx & 3);
          if (ns >= N) ns = N - 1;
          const int leaf = kLT * (ch * kRChunk + (idx >> 2)) + (src & 15);
          float p = 0.f;
          for (int d = lane; d < DP; d += 64) {
            const float df = Z[ns * ldz + d] - a32[(int64_t)leaf * DP + d];
            p = fmaf(df, df, p);
          }
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) p += __shfl_xor(p, off);
          if (lane == src) {
#pragma unroll
            for (int t = 0; t < kRChunk; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (idx == t * 4 + r) d2[t][r] = p;
            pend &= ~(1u << idx);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < kRChunk; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] = fmaf(c2, __builtin_amdgcn_sqrtf(d2[t][r]), pr[t][r]);
    };

    // ---- pass 1: online max / sum-exp per frame over the lane's leaf column
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float ssum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nchunk; ++ch) {
      f32x4 d2[kRChunk], v[kRChunk];
      chunk_logits(ch, d2, v);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float cm = v[0][r];
#pragma unroll
        for (int t = 1; t < kRChunk; ++t) cm = fmaxf(cm, v[t][r]);
        const float mn = fmaxf(m[r], cm);
        // a lane whose leaves so far are all padding keeps (m, sum) = (-inf, 0)
        float acc = (mn == -INFINITY) ? 0.f : ssum[r] * __builtin_amdgcn_exp2f(m[r] - mn);
#pragma unroll
        for (int t = 0; t < kRChunk; ++t) acc += (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(v[t][r] - mn);
        ssum[r] = acc;
        m[r] = mn;
      }
    }
    // combine the 16 leaf columns of each frame
    float M[4], inv[4], lse2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      M[r] = row16_max(m[r]);
      const float sc = (m[r] == -INFINITY) ? 0.f : ssum[r] * __builtin_amdgcn_exp2f(m[r] - M[r]);
      const float tot = row16_sum(sc);
      inv[r] = okr[r] ? __builtin_amdgcn_rcpf(tot) : 0.f;      // rows past N: phi = 0, not stored
      lse2[r] = __builtin_amdgcn_logf(tot);                    // log2 of the sum
    }

    // ---- pass 2: recompute, normalise, accumulate statistics, store through the staging tile
    for (int ch = 0; ch < nchunk; ++ch) {
      f32x4 d2[kRChunk], v[kRChunk];
      chunk_logits(ch, d2, v);
      f32x4 pv[kRChunk];
#pragma unroll
      for (int t = 0; t < kRChunk; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[t][r] = __builtin_amdgcn_exp2f(v[t][r] - M[r]) * inv[r];
      if (STATS) {
#pragma unroll
        for (int t = 0; t < kRChunk; ++t) {
          const int leaf = kLT * (ch * kRChunk + t) + c;
          float a0 = (pv[t][0] + pv[t][1]) + (pv[t][2] + pv[t][3]);
          float a2 = fmaf(pv[t][0], d2[t][0], fmaf(pv[t][1], d2[t][1], fmaf(pv[t][2], d2[t][2], pv[t][3] * d2[t][3])));
          // the 4 lane groups hold different frames of the same leaf
          a0 += __shfl_xor(a0, 16);
          a2 += __shfl_xor(a2, 16);
          a0 += __shfl_xor(a0, 32);
          a2 += __shfl_xor(a2, 32);
          if (kg == 0 && leaf < L) {
            atomicAdd(&st_lds[2 * leaf], a0);
            atomicAdd(&st_lds[2 * leaf + 1], a2);
          }
        }
      }
      // staging: [frame row][32 leaves] -> 16 B per lane, 128-byte row segments
#pragma unroll
      for (int hf = 0; hf < kRChunk / 2; ++hf) {
        const int T0 = ch * kRChunk + 2 * hf;
        if (T0 < tiles_valid) {                      // Lp is a multiple of 32: both tiles or none
          const int srow = lane >> 3, scol = 4 * (lane & 7);
#pragma unroll
          for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && logphi == nullptr) break;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int t = 2 * hf + tt;
                const float val = pass == 0 ? pv[t][r] : (v[t][r] - M[r] - lse2[r]) * kLn2;
                stage[(4 * kg + r) * kRStageLd + kLT * tt + c] = val;
              }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float* dst = (pass == 0 ? phi : logphi) + kLT * T0 + scol;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int fr = srow + 8 * q;
              const float4 o = *reinterpret_cast<const float4*>(&stage[fr * kRStageLd + scol]);
              const int64_t n = row0 + fr;
              if (n < N) *reinterpret_cast<float4*>(dst + n * Lp) = o;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
    }
  }

  if (STATS) {
    __syncthreads();
    double* out = eslab + (int64_t)blockIdx.x * 2 * L;
    for (int e = threadIdx.x; e < 2 * L; e += blockDim.x) out[e] = (double)st_lds[e];
  }
}

}  // namespace bnpvi
