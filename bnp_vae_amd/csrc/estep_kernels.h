// E-step kernels for gfx950 (MI355X), templated on the padded latent dim DP.
//   logit[n,l] = prior[vid(n),l] - 0.5*sZ*||z_n - alpha_l||_2      (UN-squared norm)
//   phi[n,:]   = softmax_l(logit[n,:])
// replacing /root/reference/src/var_inf.py:313-357 and :383-393 (the per-leaf NumPy norm, the
// per-frame Python loop with two list.index lookups, and the five tree walks of the softmax).
//
// Layout of the dense part.  ||z-a||^2 = ||z||^2 - 2 z.a + ||a||^2; only z.a is matrix-shaped
// and runs on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains) with
//     rows  i  = leaves  (A operand = alpha, from LDS or L2, pre-packed in operand order)
//     cols  j  = frames  (B operand = Z, 32 frames per wave, the row kept in registers)
// so that the accumulator puts one FRAME per lane and its LEAVES in registers: max / sum-exp
// of the softmax are in-register loops plus ONE cross-lane exchange (lane <-> lane^32), and a
// lane owns 4 consecutive leaves per register quad => 16-byte phi stores.  The accumulator is
// seeded with -||a||^2/2 so d^2 = ||z||^2 - 2*acc.
//
// The contraction index (latent dim) is permuted so that each lane's Z fragment is one
// float4 of its own row: lane (j, h=lane>>5) holds dims 8q+4h+{0..3} in zreg[q]; alpha is
// packed by pack_leaves_kernel in exactly that order: apack[((t*(DP/8)+q)*64+lane)*4+c] =
// alpha[leaf 32t+(lane&31)][dim 8q+4(lane>>5)+c], so one ds_read_b128 / global_load_dwordx4
// per lane feeds four MFMAs and is conflict-free / coalesced.
#pragma once

#include "common.h"

namespace bnpvi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kEWaves = 4;            // waves per workgroup
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
// d^2 from the expansion carries an absolute error ~ 2e-7*(||z||^2+||a||^2); below this
// fraction of (||z||^2+||a||^2)^2 its square root is no longer good to ~2e-5 and the distance
// is recomputed by direct differences (rare: a frame that coincides with a leaf mean, which is
// exactly what weighted_kmeans produces right after a split).
constexpr float kRefineRel = 6.25e-6f;

__device__ __forceinline__ int leaf_slot(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }

// index of alpha[leaf][dim] inside the packed operand image (floats)
template <int DP>
__device__ __forceinline__ int pack_index(int leaf, int dim) {
  return (((leaf >> 5) * (DP / 8) + (dim >> 3)) * 64 + ((dim >> 2) & 1) * 32 + (leaf & 31)) * 4 + (dim & 3);
}

// ------------------------------------------------------------------------------------------
// Cross term + squared distance for NT leaf tiles of one 32-frame tile.  On return acc[t][r]
// holds ||z - alpha||^2 of frame (lane&31) to leaf 32t+slot(r,h) (tile-relative numbering).
//   a4 / cv : packed alpha / -|alpha|^2/2 of the first tile (LDS in the fused regime, global
//             in the streamed one);  apack_f : the same image as plain floats (refinement).
// ------------------------------------------------------------------------------------------
template <int DP, int NT>
__device__ __forceinline__ void tile_distances(const float* __restrict__ Z, int ldz, int64_t row0,
                                               int64_t N, const float4* a4, const float* cv,
                                               const float* apack_f, int lane, int lvalid,
                                               f32x16 (&acc)[NT]) {
  const int j = lane & 31, h = lane >> 5;
  int64_t n = row0 + j;
  if (n >= N) n = N - 1;
  const float* zrow = Z + n * ldz;
  float4 zreg[DP / 8];
  float zz = 0.f;
#pragma unroll
  for (int q = 0; q < DP / 8; ++q) {
#ifdef BNPVI_ABLATE_NOZLOAD
    zreg[q] = make_float4(0.01f * lane, 0.02f * q, 0.5f, -0.25f);
    (void)zrow;
#else
    zreg[q] = *reinterpret_cast<const float4*>(zrow + 8 * q + 4 * h);
#endif
    zz = fmaf(zreg[q].x, zreg[q].x, zz);
    zz = fmaf(zreg[q].y, zreg[q].y, zz);
    zz = fmaf(zreg[q].z, zreg[q].z, zz);
    zz = fmaf(zreg[q].w, zreg[q].w, zz);
  }
  zz += __shfl_xor(zz, 32);

#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 c4 = *reinterpret_cast<const float4*>(cv + 32 * t + 8 * g + 4 * h);
      acc[t][4 * g + 0] = c4.x;
      acc[t][4 * g + 1] = c4.y;
      acc[t][4 * g + 2] = c4.z;
      acc[t][4 * g + 3] = c4.w;
    }
  }
#pragma unroll
  for (int q = 0; q < DP / 8; ++q) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#ifdef BNPVI_ABLATE_NOMFMA
      acc[t][q & 15] += zreg[q].x + zreg[q].y + zreg[q].z + zreg[q].w;
#else
      const float4 a = a4[(t * (DP / 8) + q) * 64 + lane];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, zreg[q].x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, zreg[q].y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, zreg[q].z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, zreg[q].w, acc[t], 0, 0, 0);
#endif
    }
  }
  // d^2 in place; remember which entries are too cancelled to trust
  unsigned long long pend = 0ull;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int slot = 32 * t + leaf_slot(r, h);
      const float aa = -2.f * cv[slot];
      const float v = fmaxf(fmaf(-2.f, acc[t][r], zz), 0.f);
      const float sn = zz + aa;
      if ((v < kRefineRel * sn * sn) && (slot < lvalid)) pend |= 1ull << (t * 16 + r);
      acc[t][r] = v;
    }
  }
  // rare path, one flagged (frame, leaf) pair at a time: the whole wave computes that pair's
  // distance by direct differences (64 lanes split the dims), the owning lane patches it in.
  while (true) {
    const unsigned long long b = __ballot(pend != 0ull);
    if (b == 0ull) break;
    const int src = __ffsll((long long)b) - 1;
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(pend & 0xffffffffull), src);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(pend >> 32), src);
    const unsigned long long pm = ((unsigned long long)hi << 32) | lo;
    const int idx = __ffsll((long long)pm) - 1;
    const int leaf = 32 * (idx >> 4) + leaf_slot(idx & 15, src >> 5);
    int64_t ns = row0 + (src & 31);
    if (ns >= N) ns = N - 1;
    float p = 0.f;
    for (int d = lane; d < DP; d += 64) {
      const float df = Z[ns * ldz + d] - apack_f[pack_index<DP>(leaf, d)];
      p = fmaf(df, df, p);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) p += __shfl_xor(p, off);
    if (lane == src) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (idx == t * 16 + r) acc[t][r] = p;
      pend &= ~(1ull << idx);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused regime: Lp = 32*NT <= 128 -- the whole row of logits stays in registers, phi is
// written once (algorithmic traffic: read Z once, write phi once).
// ------------------------------------------------------------------------------------------
template <int DP, int NT, bool STATS>
__global__ __launch_bounds__(kEWaves * 64, (DP <= 64 && NT <= 2 && !STATS) ? 2 : 1) void estep_fused_kernel(
    const float* __restrict__ Z, int ldz, const int32_t* __restrict__ vid, int64_t N, int L,
    const float* __restrict__ apack, const float* __restrict__ cvec,
    const float* __restrict__ prior, float half_sz, float* __restrict__ phi,
    float* __restrict__ logphi, int32_t* __restrict__ assign, double* __restrict__ eslab) {
  constexpr int Lp = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float4* a_lds = reinterpret_cast<float4*>(smem_raw);                  // [NT*(DP/8)*64]
  float* c_lds = reinterpret_cast<float*>(a_lds + NT * (DP / 8) * 64);  // [Lp]
  double* st_lds = reinterpret_cast<double*>(c_lds + Lp);               // [Lp][2]

  for (int e = threadIdx.x; e < NT * (DP / 8) * 64; e += blockDim.x)
    a_lds[e] = reinterpret_cast<const float4*>(apack)[e];
  for (int e = threadIdx.x; e < Lp; e += blockDim.x) c_lds[e] = cvec[e];
  for (int e = threadIdx.x; e < 2 * Lp; e += blockDim.x) st_lds[e] = 0.0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  const int64_t ntiles = (N + 31) / 32;
  // per-lane fp32 accumulators of (phi, phi*d^2) per logit register (STATS only; LDS float
  // atomics were tried instead and stalled the LDS pipe 4x: profiles/r01_notes.md)
  float s0[STATS ? NT : 1][16], s2[STATS ? NT : 1][16];
  if (STATS) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[t][r] = s2[t][r] = 0.f;
  }

  for (int64_t tile = (int64_t)blockIdx.x * kEWaves + wave; tile < ntiles;
       tile += (int64_t)gridDim.x * kEWaves) {
    const int64_t n = tile * 32 + j;
    const bool ok = n < N;
    const int64_t nn = ok ? n : N - 1;

    f32x16 d2[NT];
    tile_distances<DP, NT>(Z, ldz, tile * 32, N, a_lds, c_lds, reinterpret_cast<const float*>(a_lds),
                           lane, L, d2);

    // logits (in place of the prior), running max and arg-max (first index wins ties)
    const float* prow = prior + (int64_t)vid[nn] * Lp;
    float lg[NT][16];
    float m = -INFINITY;
    int best = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#ifdef BNPVI_ABLATE_NOPRIOR
        const float4 p4 = make_float4(0.f, 0.1f, 0.2f, 0.3f);
        (void)prow;
#else
        const float4 p4 = *reinterpret_cast<const float4*>(prow + 32 * t + 8 * g + 4 * h);
#endif
        const float pr[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = 4 * g + c;
          const float v = fmaf(-half_sz, __builtin_amdgcn_sqrtf(d2[t][r]), pr[c]);
          lg[t][r] = v;
          if (v > m) {
            m = v;
            best = 32 * t + leaf_slot(r, h);
          }
        }
      }
    {
      const float om = __shfl_xor(m, 32);
      const int ob = __shfl_xor(best, 32);
      if (om > m || (om == m && ob < best)) {
        m = om;
        best = ob;
      }
    }
    float ssum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = lg[t][r] - m;
        lg[t][r] = x;
        ssum += __builtin_amdgcn_exp2f(x * kLog2e);
      }
    ssum += __shfl_xor(ssum, 32);
    const float lse = __builtin_amdgcn_logf(ssum) * kLn2;

    float* prow_out = phi + n * Lp;
    float* lrow_out = logphi ? logphi + n * Lp : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float pv[4], lv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = 4 * g + c;
          lv[c] = lg[t][r] - lse;
          pv[c] = __builtin_amdgcn_exp2f(lv[c] * kLog2e);
          if (STATS && ok) {
            s0[t][r] += pv[c];
            s2[t][r] = fmaf(pv[c], d2[t][r], s2[t][r]);
          }
        }
#ifdef BNPVI_ABLATE_NOSTORE
        if (ok && pv[0] + pv[1] + pv[2] + pv[3] == -123.f) {
#else
        if (ok) {
#endif
          *reinterpret_cast<float4*>(prow_out + 32 * t + 8 * g + 4 * h) =
              make_float4(pv[0], pv[1], pv[2], pv[3]);
          if (lrow_out)
            *reinterpret_cast<float4*>(lrow_out + 32 * t + 8 * g + 4 * h) =
                make_float4(lv[0], lv[1], lv[2], lv[3]);
        }
      }
    }
    if (assign && ok && h == 0) assign[n] = best;
  }

  if (STATS) {
    // reduce the per-lane leaf accumulators over the 32 frame lanes of each half-wave, then
    // fp64 accumulation in LDS, one slab per workgroup (summed in fixed order afterwards)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a = s0[t][r], b = s2[t][r];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
          a += __shfl_xor(a, off);
          b += __shfl_xor(b, off);
        }
        if (j == 0) {
          const int leaf = 32 * t + leaf_slot(r, h);
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
// Streamed regime (larger trees, and the one-hot mode), pass A: raw logits for one 32-frame
// tile x one chunk of NT leaf tiles, written into phi; part[n][chunk] = (max, sum exp(logit-max))
// in soft mode, (max of -d^2, arg-max leaf as float bits) in hard mode.
// ------------------------------------------------------------------------------------------
template <int DP, int NT, bool HARD>
__global__ __launch_bounds__(kEWaves * 64) void estep_logits_kernel(
    const float* __restrict__ Z, int ldz, const int32_t* __restrict__ vid, int64_t N, int L,
    int Lp, const float* __restrict__ apack, const float* __restrict__ cvec,
    const float* __restrict__ prior, float half_sz, float* __restrict__ phi,
    float2* __restrict__ part, int nchunk) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  const int chunk = blockIdx.y;
  const int t0 = chunk * NT;                   // first leaf tile of this chunk
  const int64_t ntiles = (N + 31) / 32;
  const float* ap = apack + (int64_t)t0 * (DP / 8) * 256;
  const float4* a4 = reinterpret_cast<const float4*>(ap);
  const float* cv = cvec + 32 * t0;

  for (int64_t tile = (int64_t)blockIdx.x * kEWaves + wave; tile < ntiles;
       tile += (int64_t)gridDim.x * kEWaves) {
    const int64_t n = tile * 32 + j;
    const bool ok = n < N;
    const int64_t nn = ok ? n : N - 1;
    f32x16 d2[NT];
    tile_distances<DP, NT>(Z, ldz, tile * 32, N, a4, cv, ap, lane, L - 32 * t0, d2);
    const float* prow = HARD ? nullptr : prior + (int64_t)vid[nn] * Lp + 32 * t0;
    float m = -INFINITY;
    int best = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool tile_ok = 32 * (t0 + t) < Lp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!HARD && tile_ok) p4 = *reinterpret_cast<const float4*>(prow + 32 * t + 8 * g + 4 * h);
        const float pr[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = 4 * g + c;
          const int leaf = 32 * (t0 + t) + leaf_slot(r, h);
          float v;
          if (HARD)
            v = leaf < L ? -d2[t][r] : -INFINITY;
          else
            v = tile_ok ? fmaf(-half_sz, __builtin_amdgcn_sqrtf(d2[t][r]), pr[c]) : -INFINITY;
          d2[t][r] = v;
          if (v > m) {
            m = v;
            best = leaf;
          }
        }
      }
    }
    {
      const float om = __shfl_xor(m, 32);
      const int ob = __shfl_xor(best, 32);
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
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ssum += (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((d2[t][r] - m) * kLog2e);
      ssum += __shfl_xor(ssum, 32);
      second = ssum;
    }
    if (ok) {
      float* row = phi + n * Lp + 32 * t0;
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (32 * (t0 + t) < Lp) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(row + 32 * t + 8 * g + 4 * h) =
                make_float4(d2[t][4 * g], d2[t][4 * g + 1], d2[t][4 * g + 2], d2[t][4 * g + 3]);
        }
      if (h == 0) part[n * nchunk + chunk] = make_float2(m, second);
    }
  }
}

// per-DP launchers (estep_inst.hip, compiled once per DP)
struct EstepArgs {
  bnpvi_ctx* ctx;
  const bnpvi_tree* tree;
  const float* Z;
  int ldz;
  const int32_t* vid;
  int64_t N;
  const float* apack;
  const float* cvec;
  const float* prior;
  float half_sz;
  int mode;
  float* phi;
  float* logphi;
  int32_t* assign;
  double* estats;
  hipStream_t stream;
};

template <int DP> int estep_dispatch(const EstepArgs& a);

// shared (estep.hip)
int estep_finish_streamed(const EstepArgs& a, const float2* part, int nchunk);
int estep_reduce_slabs(const EstepArgs& a, const double* eslab, int nslab);
constexpr int kStreamNT = 2;      // leaf tiles per chunk in the streamed regime (64 leaves)

}  // namespace bnpvi
