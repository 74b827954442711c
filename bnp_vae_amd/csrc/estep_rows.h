// Two-pass ("row") E-step for trees whose logit row does not fit in registers (64 < L, packed
// leaf operands <= 128 KiB), soft responsibilities.  Same math and the same packed operands as
// the fused kernel (estep_kernels.h); replaces /root/reference/src/var_inf.py:313-357,383-393.
//
// Orientation.  The MFMA operands are swapped: A = Z fragment (row = frame), B = alpha fragment
// (col = leaf), so the accumulator holds ONE LEAF per lane (column lane & 15) and 4 FRAMES per
// lane (rows 4*(lane>>4) + r).  Consequences:
//   * per-leaf statistics (S0' = sum phi, S2 = sum phi d^2) need one accumulator per leaf tile
//     instead of four per frame lane: they are summed over the lane's 4 frames in registers and
//     added with fp64 LDS atomics to a workgroup-wide table once per (tile, leaf tile);
//   * the softmax runs over lanes and leaf tiles: per frame an ONLINE max / sum-exp over chunks
//     of 64 leaves (pass 1), combined over the 16 leaf lanes once per frame tile;
//   * pass 2 recomputes the chunk's logits (MFMAs on the f16 pipe are cheap) and writes
//     phi = exp2(logit - max) / sum from the accumulator layout.  A store instruction of that layout
//     covers 4 frame rows x 64 B; for whole tiles the lane-group pairs trade two leaf tiles first
//     (v_permlane16_swap) so that every instruction writes whole 128-byte lines, non-temporal
//     (an LDS staging tile for 16-byte stores measured slower: it serialises the wave on LDS round
//     trips and caps the workgroup at 8 waves).
// HBM traffic is the algorithmic minimum: Z read once, phi written once (no logits round trip).
#pragma once

#include <type_traits>

#include "estep_kernels.h"

namespace bnpvi {

#ifndef BNPVI_ROWS_WAVES
#define BNPVI_ROWS_WAVES 12
#endif
constexpr int kRWaves = BNPVI_ROWS_WAVES;   // waves per workgroup (3 per SIMD)
constexpr int kRChunk = 4;            // 16-leaf tiles per online-softmax chunk (64 leaves)

template <int DP>
struct RowsLds {
  static size_t bytes(int NTp, bool stats) {
    return (size_t)NTp * (DP / 32) * 2 * 64 * 16 + (size_t)kLT * NTp * 4 +
           (stats ? (size_t)kLT * NTp * 16 : 0);
  }
};

// NCH = 0: two passes over any number of 64-leaf chunks.  NCH = 2 or 4 (up to 128 / 256
// leaves, no statistics): exactly NCH chunks whose logits stay in registers (16 * NCH per lane), ONE pass of
// MFMAs / distances / square roots (see the body).  512 leaves would need 128 logit registers
// per lane: measured slower than two passes both at 1 wave per SIMD (no spills) and at 2 (spills).
constexpr int rows_waves(int NCH) { return NCH >= 4 ? 8 : kRWaves; }

template <int DP, bool STATS, int NCH = 0>
__global__ __launch_bounds__(rows_waves(NCH) * 64, rows_waves(NCH) / 4) void estep_rows_kernel(
    const float* __restrict__ Z, int ldz, const int32_t* __restrict__ vid, int64_t N, int L, int Lp,
    int NTp /* 16-leaf tiles, padded to a multiple of kRChunk */, const f16x8* __restrict__ apack,
    const float* __restrict__ aavec, const float* __restrict__ a32, const float* __restrict__ scal,
    const float* __restrict__ prior, float half_sz, float* __restrict__ phi, float* __restrict__ logphi,
    double* __restrict__ eslab, int tiles_per_wg) {
  constexpr int S = DP / 32;
  constexpr int kRWaves = rows_waves(NCH);      // shadows the namespace-level default
  constexpr int kCT = kRChunk;                  // leaf tiles per chunk
  constexpr int kNChunk = NCH;                  // single-pass variants: chunks per tile
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f16x8* a_lds = reinterpret_cast<f16x8*>(smem_raw);                    // [NTp*S*2*64]
  float* aa_lds = reinterpret_cast<float*>(a_lds + (size_t)NTp * S * 2 * 64);   // [16*NTp]
  // [16*NTp][2] (STATS), fp64: the waves of the workgroup add their per-(tile, leaf tile) partial sums with
  // LDS fp64 atomics, whose order varies from run to run -- in fp64 that moves the result by ~1e-16
  // relative (fp32 atomics moved it by ~1e-5, and the split / merge decisions sit on these sums)
  double* st_lds = reinterpret_cast<double*>(aa_lds + kLT * NTp);
  const int tiles_valid = Lp / kLT;

  for (int e = threadIdx.x; e < NTp * S * 2 * 64; e += blockDim.x)
    a_lds[e] = e < tiles_valid * S * 2 * 64 ? apack[e] : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = threadIdx.x; e < kLT * NTp; e += blockDim.x) aa_lds[e] = e < Lp ? aavec[e] : 0.f;
  if (STATS)
    for (int e = threadIdx.x; e < 2 * kLT * NTp; e += blockDim.x) st_lds[e] = 0.0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15;
  const int kg = lane >> 4;
  const float inv_sa = scal[0];
  const float aamax = scal[2];
  const float c2 = -half_sz * kLog2e;
  const int nchunk = NTp / kCT;

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
    zz = cols4_sum(zz);
    amax = cols4_max(amax);
    const int kz = scale_exponent_bits(__float_as_int(amax));
    f16x8 zh[S], zl[S];
#pragma unroll
    for (int s = 0; s < S; ++s) split8(zA[s], zB[s], pow2f(kz), zh[s], zl[s]);
    // next tile's Z rows: in flight across the whole tile (two-pass) / issued after phase 1 of the
    // single-pass variant, whose phase 1 has no registers to spare
    if (NCH == 0 && tile + kRWaves < t1) load_ztile<DP>(Z, ldz, (tile + kRWaves) * kFT, N, lane, zA, zB);

    // the lane's 4 frames (rows 4kg + r): ||z||^2, un-scaling factor, prior row, validity.
    // Frames are handled in PAIRS (r, r+1) on the packed fp32 pipe.
    const float unscale = -2.f * pow2f(-kz) * inv_sa;
    f32x2 zz2[2], un2[2], thr2[2];
    const float* prow[4];
    bool okr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float zr = __shfl(zz, 4 * kg + r);
      zz2[r >> 1][r & 1] = zr;
      un2[r >> 1][r & 1] = __shfl(unscale, 4 * kg + r);
      const float sn = zr + aamax;
      thr2[r >> 1][r & 1] = refine_threshold(sn);
      const int64_t n = row0 + 4 * kg + r;
      okr[r] = n < N;
      prow[r] = prior + (int64_t)vid[okr[r] ? n : N - 1] * Lp + c;
    }
    const int64_t nlast = row0 + kFT - 1 < N ? row0 + kFT - 1 : N - 1;
    const bool onevid = vid[row0] == vid[nlast];       // wave-uniform: one prior row for the tile
    const f32x2 c22 = {c2, c2};
    const bool full = row0 + kFT <= N;                  // wave-uniform: no row of the tile is past N
    unsigned voff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) voff[r] = (unsigned)((4 * kg + r) * Lp + c);
    // whole-line stores of tile pairs (pass 2): row 4 (kg & ~1) + r resp. 4 (kg | 1) + r, leaf tile + (kg & 1)
    const int dA = (kg & 1) ? kLT - 4 * Lp : 0;
    const int dB = (kg & 1) ? kLT : 4 * Lp;

    // One 64-leaf chunk for the lane's (leaf column, 4 frames): squared distances d[t][pair] and
    // log2-unit logits v[t][pair].  No clamp of d at zero: a negative d is below the
    // cancellation bound, so it is replaced by the exact value on the rare path; padded leaves
    // (prior -inf) go through |d|.
    // full_c: every tile of the chunk is a valid leaf tile (all chunks but possibly the last): no
    // per-entry select of -inf priors.  screen_c: run the cancellation screen; pass 2 drops it
    // when pass 1 (same arithmetic, same values) found nothing in the whole tile.  Returns
    // whether the screen fired (wave-uniform).
    auto chunk_logits = [&](auto full_c, auto screen_c, int ch, f32x2 (&d)[kCT][2], f32x2 (&v)[kCT][2],
                            f32x2 (&pr)[kCT][2]) -> bool {
      constexpr bool kFullChunk = decltype(full_c)::value;
      constexpr bool kScreen = decltype(screen_c)::value;
      // the chunk's prior entries first: their L2 round trip runs under the MFMA block (padded
      // tiles read the last valid tile's address and are overwritten with -inf below)
      float prv[kCT][4];
      if (onevid) {
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          const int T = ch * kCT + t;
          const float p = prow[0][kLT * (T < tiles_valid ? T : tiles_valid - 1)];
#pragma unroll
          for (int r = 0; r < 4; ++r) prv[t][r] = p;
        }
      } else {
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          const int T = ch * kCT + t;
#pragma unroll
          for (int r = 0; r < 4; ++r) prv[t][r] = prow[r][kLT * (T < tiles_valid ? T : tiles_valid - 1)];
        }
      }
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 acc[kCT];
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          const int T = ch * kCT + t;
          const f16x8 ah = a_lds[((T * S + s) * 2 + 0) * 64 + lane];
          const f16x8 al = a_lds[((T * S + s) * 2 + 1) * 64 + lane];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zl[s], ah, s ? acc[t] : zero4, 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[s], al, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[s], ah, acc[t], 0, 0, 0);
        }
      }
#ifndef BNPVI_ROWS_NO_SCHED
      // issue order: the first k-step's 8 operand reads, then every 3 MFMAs hide the next 2 reads
      // (LDS latency overlaps matrix-pipe time instead of stalling in front of each MFMA)
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * kCT, 0);
#pragma unroll
      for (int g = 0; g < (S - 1) * kCT; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 3 * kCT, 0);
#endif
#pragma unroll
      for (int t = 0; t < kCT; ++t) {
        const int T = ch * kCT + t;
        const bool tv = kFullChunk || T < tiles_valid;
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[t][r >> 1][r & 1] = tv ? prv[t][r] : -INFINITY;
      }
      f32x2 dmin[2] = {{INFINITY, INFINITY}, {INFINITY, INFINITY}};
#pragma unroll
      for (int t = 0; t < kCT; ++t) {
        const float aa = aa_lds[kLT * (ch * kCT + t) + c];
        const f32x2 aa2 = {aa, aa};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 a2 = {acc[t][2 * h], acc[t][2 * h + 1]};
          d[t][h] = un2[h] * a2 + zz2[h] + aa2;
          if (kScreen) {
            dmin[h][0] = fminf(dmin[h][0], d[t][h][0]);
            dmin[h][1] = fminf(dmin[h][1], d[t][h][1]);
          }
        }
      }
      bool fired = false;
      // cancellation screen (see estep_kernels.h): rare exact recomputation by direct differences
      const bool sus = (dmin[0][0] < thr2[0][0]) | (dmin[0][1] < thr2[0][1]) | (dmin[1][0] < thr2[1][0]) |
                       (dmin[1][1] < thr2[1][1]);
      if (kScreen && __builtin_expect(__ballot(sus) != 0ull, 0)) {
        fired = true;
        asm volatile("" ::: "memory");      // nothing of the rare path is speculated above the branch
        unsigned pend = 0u;
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          const int leaf = kLT * (ch * kCT + t) + c;
          const float aa = aa_lds[leaf];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sn = zz2[r >> 1][r & 1] + aa;
            if (d[t][r >> 1][r & 1] < refine_threshold(sn) && leaf < L) pend |= 1u << (t * 4 + r);
          }
        }
        // as in tile_distances: one flagged pair per 16-lane row group (4 frames x 16 leaf columns)
        // and pass, the group's lanes splitting the rows
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
                  if (idx == t * 4 + r) d[t][r >> 1][r & 1] = p;
              pend &= pend - 1u;
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < kCT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 sq = {__builtin_amdgcn_sqrtf(__builtin_fabsf(d[t][h][0])),
                            __builtin_amdgcn_sqrtf(__builtin_fabsf(d[t][h][1]))};
          v[t][h] = c22 * sq + pr[t][h];
        }
      return fired;
    };
    auto run_chunk = [&](bool screen, int ch, f32x2 (&d)[kCT][2], f32x2 (&v)[kCT][2], f32x2 (&pr)[kCT][2]) -> bool {
      using Yes = std::true_type;
      using No = std::false_type;
      const bool full_chunk = (ch + 1) * kCT <= tiles_valid;          // wave-uniform
      if (screen) return full_chunk ? chunk_logits(Yes{}, Yes{}, ch, d, v, pr) : chunk_logits(No{}, Yes{}, ch, d, v, pr);
      return full_chunk ? chunk_logits(Yes{}, No{}, ch, d, v, pr) : chunk_logits(No{}, No{}, ch, d, v, pr);
    };

    if constexpr (NCH > 0) {
      // ---- single pass: every logit of the lane's (leaf column, 4 frames) stays in registers.
      // Phase 1 computes them (the only MFMAs, distances and square roots) with a running max;
      // phase 2 sums exp2(v - max); phase 3 normalises and stores.  No statistics here: they
      // need d next to phi, and every way of having it (recovered from v, per-lane partials,
      // per-tile cross-lane sums) measured slower than the two-pass kernel, where d is live.
      static_assert(!STATS, "the single-pass row kernel has no statistics epilogue");
      f32x2 v[kNChunk][kCT][2];
      f32x2 m2[2] = {{-INFINITY, -INFINITY}, {-INFINITY, -INFINITY}};
#pragma unroll
      for (int ch = 0; ch < kNChunk; ++ch) {
        f32x2 d[kCT][2], pr[kCT][2];
        run_chunk(true, ch, d, v[ch], pr);
#pragma unroll
        for (int t = 0; t < kCT; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            m2[h][0] = fmaxf(m2[h][0], v[ch][t][h][0]);
            m2[h][1] = fmaxf(m2[h][1], v[ch][t][h][1]);
          }
        __builtin_amdgcn_sched_barrier(0);       // chunks one after the other: bounded register use
      }
      if (tile + kRWaves < t1) load_ztile<DP>(Z, ldz, (tile + kRWaves) * kFT, N, lane, zA, zB);
      f32x2 M2[2], inv2[2], lse2[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) M2[r >> 1][r & 1] = row16_max(m2[r >> 1][r & 1]);   // finite: L >= 1
      f32x2 ss2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int ch = 0; ch < kNChunk; ++ch)
#pragma unroll
        for (int t = 0; t < kCT; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 x = v[ch][t][h] - M2[h];
            ss2[h] += (f32x2){__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
          }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tot = row16_sum(ss2[r >> 1][r & 1]);
        inv2[r >> 1][r & 1] = okr[r] ? __builtin_amdgcn_rcpf(tot) : 0.f;
        lse2[r >> 1][r & 1] = __builtin_amdgcn_logf(tot);
      }
#pragma unroll
      for (int ch = 0; ch < kNChunk; ++ch) {
        f32x2 pv[kCT][2];
#pragma unroll
        for (int t = 0; t < kCT; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x2 x = v[ch][t][h] - M2[h];
            pv[t][h] = (f32x2){__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])} * inv2[h];
          }
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          const int T = ch * kCT + t;
          if (T < tiles_valid) {
            float* tb = phi + row0 * Lp + kLT * T;
            float* lb = logphi ? logphi + row0 * Lp + kLT * T : nullptr;
            if (full) {
#pragma unroll
              for (int r = 0; r < 4; ++r) tb[voff[r]] = pv[t][r >> 1][r & 1];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (okr[r]) tb[voff[r]] = pv[t][r >> 1][r & 1];
            }
            if (lb) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (okr[r])
                  lb[voff[r]] = (v[ch][t][r >> 1][r & 1] - M2[r >> 1][r & 1] - lse2[r >> 1][r & 1]) * kLn2;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      continue;
    }

    // ---- pass 1: online max / sum-exp per frame over the lane's leaf column
    bool tile_sus = false;             // wave-uniform: the cancellation screen fired somewhere in this tile
    f32x2 m2[2] = {{-INFINITY, -INFINITY}, {-INFINITY, -INFINITY}};
    f32x2 ss2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int ch = 0; ch < nchunk; ++ch) {
      f32x2 d[kCT][2], v[kCT][2], pr[kCT][2];
      tile_sus |= run_chunk(true, ch, d, v, pr);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x2 mn;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float cm = fmaxf(fmaxf(v[0][h][q], v[1][h][q]), fmaxf(v[2][h][q], v[3][h][q]));
          mn[q] = fmaxf(m2[h][q], cm);
        }
        // a lane whose leaves so far are all padding has mn = -inf: keep (m, sum) = (-inf, 0);
        // exp2 of (-inf) - (finite) is 0, of (-inf) - (-inf) NaN -> shift by 0 instead
        const f32x2 sh = {mn[0] == -INFINITY ? 0.f : mn[0], mn[1] == -INFINITY ? 0.f : mn[1]};
        const f32x2 resc = m2[h] - sh;
        f32x2 acc2 = ss2[h] * (f32x2){__builtin_amdgcn_exp2f(resc[0]), __builtin_amdgcn_exp2f(resc[1])};
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          const f32x2 x = v[t][h] - sh;
          acc2 += (f32x2){__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        }
        ss2[h] = acc2;
        m2[h] = mn;
      }
    }
    // combine the 16 leaf columns of each frame
    f32x2 M2[2], inv2[2], lse2[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mr = m2[r >> 1][r & 1];
      const float Mr = row16_max(mr);
      const float sc = (mr == -INFINITY) ? 0.f : ss2[r >> 1][r & 1] * __builtin_amdgcn_exp2f(mr - Mr);
      const float tot = row16_sum(sc);
      M2[r >> 1][r & 1] = Mr;
      inv2[r >> 1][r & 1] = okr[r] ? __builtin_amdgcn_rcpf(tot) : 0.f;   // rows past N: phi = 0, not stored
      lse2[r >> 1][r & 1] = __builtin_amdgcn_logf(tot);                  // log2 of the sum
    }

    // ---- pass 2: recompute, normalise, accumulate statistics, store
    for (int ch = 0; ch < nchunk; ++ch) {
      f32x2 d[kCT][2], v[kCT][2], pr[kCT][2];
      run_chunk(tile_sus, ch, d, v, pr);
      f32x2 pv[kCT][2];
#pragma unroll
      for (int t = 0; t < kCT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 x = v[t][h] - M2[h];
          pv[t][h] = (f32x2){__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])} * inv2[h];
        }
      if (STATS) {
#pragma unroll
        for (int t = 0; t < kCT; ++t) {
          const int leaf = kLT * (ch * kCT + t) + c;
          const f32x2 s0p = pv[t][0] + pv[t][1];
          const f32x2 s2p = pv[t][0] * d[t][0] + pv[t][1] * d[t][1];
          // the 4 lane groups hold different frames of the same leaf
          const float a0 = cols4_sum(s0p[0] + s0p[1]);
          const float a2 = cols4_sum(s2p[0] + s2p[1]);
          if (kg == 0 && leaf < L) {
            atomicAdd(&st_lds[2 * leaf], (double)a0);
            atomicAdd(&st_lds[2 * leaf + 1], (double)a2);
          }
        }
      }
      // direct stores: per instruction the wave writes 4 frame rows x 64 B (16 leaf lanes x 4 B);
      // the other half of each 128-byte line follows with the next leaf tile.  Wave-uniform base
      // (tile row, leaf tile) + a fixed 32-bit lane offset per frame row.
#ifndef BNPVI_ROWS_STORE_64B
      // Whole tiles without log phi: two leaf tiles at a time, the lane-group pairs (kg, kg ^ 1) trade them
      // (v_permlane16_swap) so that an instruction writes whole 128-byte lines: rows 4 (kg & ~1) + r with one
      // instruction, rows 4 (kg | 1) + r with the next; lanes of even kg carry tile T, of odd kg tile T + 1.
      // Non-temporal like the fused kernel's (phi is next read after Z has gone by again); with 64-byte
      // pieces that policy cost +11 %, with whole lines it gains 2 % here and 3 % on the iteration
      // (config 3, 2M frames: E-step 1.71 -> 1.68 ms, iteration 2.74 -> 2.67 ms).
      if (full && logphi == nullptr) {
#pragma unroll
        for (int t = 0; t < kCT; t += 2) {
          const int T = ch * kCT + t;
          if (T < tiles_valid) {
            float* tb = phi + row0 * Lp + kLT * T;
            float* tbA = tb + dA;
            float* tbB = tb + dB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(pv[t][r >> 1][r & 1]),
                                                                __float_as_uint(pv[t + 1][r >> 1][r & 1]), false, false);
#ifndef BNPVI_ROWS_STORE_PLAIN
              __builtin_nontemporal_store(__uint_as_float(sw[0]), tbA + voff[r]);
              __builtin_nontemporal_store(__uint_as_float(sw[1]), tbB + voff[r]);
#else
              tbA[voff[r]] = __uint_as_float(sw[0]);
              tbB[voff[r]] = __uint_as_float(sw[1]);
#endif
            }
          }
        }
        continue;
      }
#endif
#pragma unroll
      for (int t = 0; t < kCT; ++t) {
        const int T = ch * kCT + t;
        if (T < tiles_valid) {
          float* tb = phi + row0 * Lp + kLT * T;
          float* lb = logphi ? logphi + row0 * Lp + kLT * T : nullptr;
          if (full) {
#ifdef BNPVI_ROWS_ABLATE_NOSTORE
            if (pv[t][0][0] == -123.f)
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) tb[voff[r]] = pv[t][r >> 1][r & 1];   // (non-temporal dword stores: +11 %)
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (okr[r]) tb[voff[r]] = pv[t][r >> 1][r & 1];
          }
          if (lb) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (okr[r]) lb[voff[r]] = (v[t][r >> 1][r & 1] - M2[r >> 1][r & 1] - lse2[r >> 1][r & 1]) * kLn2;
          }
        }
      }
    }
  }

  if (STATS) {
    __syncthreads();
    double* out = eslab + (int64_t)blockIdx.x * 2 * L;
    for (int e = threadIdx.x; e < 2 * L; e += blockDim.x) out[e] = st_lds[e];
  }
}

}  // namespace bnpvi
