// M-step data pass for gfx950: responsibility-weighted sufficient statistics
//   S0[l] = sum_n phi[n,l],  S1[l,:] = sum_n phi[n,l] z_n,  Q[l,:] = sum_n phi[n,l] z_n^2
//   m[v,l] = sum_{n in video v} phi[n,l]
// replacing the per-leaf NumPy passes of /root/reference/src/var_inf.py:265-297 and :400-403.
//
// R^T.Z is a GEMM whose contraction index is the FRAME: with v_mfma_f32_32x32x2_f32 the A
// operand is lane(i=leaf, k=frame parity) and the B operand lane(k=frame parity, j=dim), so a
// wave reads 32 consecutive floats of a phi row and of a Z row per half-wave -- both operands
// stream straight from HBM, fully coalesced, with no LDS transpose.  Products and partial sums
// are fp32 (exact f32 FMA chains in the matrix core) over at most kMaxChunk frames; everything
// above that is accumulated in fp64 (LDS across the waves of a workgroup, then a fixed-order
// slab reduction, so results are bit-reproducible).
#include "common.h"

namespace bnpvi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kStatWaves = 4;          // waves per workgroup
constexpr int kMaxChunk = 8192;        // frames summed in fp32 before promotion to fp64
constexpr int kPairUnroll = 4;

template <int TL, int TD, bool WANT_Q>
__global__ __launch_bounds__(kStatWaves * 64) void leaf_stats_kernel(
    const float* __restrict__ Z, int ldz, const float* __restrict__ phi, int ldphi,
    const int32_t* __restrict__ vid, int64_t N, int D, int Dp, int L, int64_t chunk,
    double* __restrict__ slab, int W, double* __restrict__ mass) {
  constexpr int NQ = WANT_Q ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* acc_lds = reinterpret_cast<double*>(smem_raw);          // [TL*TD*NQ][16][64]
  double* s0_lds = acc_lds + TL * TD * NQ * 16 * 64;              // [TL][32]

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 31;
  const int h = lane >> 5;
  const int ltile0 = blockIdx.y * TL;
  const int dtile0 = blockIdx.z * TD;
  const bool lead_dim = (blockIdx.z == 0);
  const bool do_mass = lead_dim && (mass != nullptr);

  const int64_t c = (int64_t)blockIdx.x * kStatWaves + wave;
  const int64_t n0 = c * chunk;
  const int64_t n1 = (n0 + chunk < N) ? n0 + chunk : N;

  f32x16 acc[TL][TD][NQ];
#pragma unroll
  for (int a = 0; a < TL; ++a)
#pragma unroll
    for (int b = 0; b < TD; ++b)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][q][r] = 0.f;

  int lcol[TL];
  bool lok[TL];
  int dcol[TD];
  bool dok[TD];
#pragma unroll
  for (int a = 0; a < TL; ++a) {
    lcol[a] = (ltile0 + a) * kTile + i;
    lok[a] = lcol[a] < ldphi;
  }
#pragma unroll
  for (int b = 0; b < TD; ++b) {
    dcol[b] = (dtile0 + b) * kTile + i;
    dok[b] = dcol[b] < Dp;
  }

  float s0[TL], msum[TL];
#pragma unroll
  for (int a = 0; a < TL; ++a) s0[a] = msum[a] = 0.f;
  int cur_vid = -1;

  for (int64_t p0 = n0; p0 < n1; p0 += 2 * kPairUnroll) {
    float av[kPairUnroll][TL], bv[kPairUnroll][TD];
    int vv[kPairUnroll];
#pragma unroll
    for (int u = 0; u < kPairUnroll; ++u) {
      const int64_t n = p0 + 2 * u + h;
      const bool ok = n < n1;
      const int64_t nn = ok ? n : (n1 - 1);
#pragma unroll
      for (int a = 0; a < TL; ++a) av[u][a] = (ok && lok[a]) ? phi[nn * ldphi + lcol[a]] : 0.f;
#pragma unroll
      for (int b = 0; b < TD; ++b) bv[u][b] = (ok && dok[b]) ? Z[nn * ldz + dcol[b]] : 0.f;
      vv[u] = (do_mass && ok) ? vid[nn] : -2;
    }
#pragma unroll
    for (int u = 0; u < kPairUnroll; ++u) {
#pragma unroll
      for (int a = 0; a < TL; ++a) {
#pragma unroll
        for (int b = 0; b < TD; ++b) {
          acc[a][b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][b], acc[a][b][0], 0, 0, 0);
          if (WANT_Q)
            acc[a][b][NQ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][b] * bv[u][b],
                                                                   acc[a][b][NQ - 1], 0, 0, 0);
        }
      }
      if (lead_dim) {
        if (do_mass && vv[u] != -2 && vv[u] != cur_vid) {
          // video boundary (frames of a video are contiguous): flush the running per-video sum
          if (cur_vid >= 0) {
#pragma unroll
            for (int a = 0; a < TL; ++a)
              if (lcol[a] < L) atomicAdd(&mass[(int64_t)cur_vid * L + lcol[a]], (double)msum[a]);
          }
#pragma unroll
          for (int a = 0; a < TL; ++a) msum[a] = 0.f;
          cur_vid = vv[u];
        }
#pragma unroll
        for (int a = 0; a < TL; ++a) {
          s0[a] += av[u][a];
          msum[a] += av[u][a];
        }
      }
    }
  }
  if (do_mass && cur_vid >= 0) {
#pragma unroll
    for (int a = 0; a < TL; ++a)
      if (lcol[a] < L) atomicAdd(&mass[(int64_t)cur_vid * L + lcol[a]], (double)msum[a]);
  }

  // ---- combine the workgroup's waves in fp64 (wave order fixed => deterministic) ----
  for (int w = 0; w < kStatWaves; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < TL; ++a)
#pragma unroll
        for (int b = 0; b < TD; ++b)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              double* dst = &acc_lds[(((a * TD + b) * NQ + q) * 16 + r) * 64 + lane];
              const double v = (double)acc[a][b][q][r];
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

  // ---- write this workgroup's slab: slab[blockIdx.x][leaf][W] ----
  double* out = slab + (int64_t)blockIdx.x * L * W;
  for (int e = threadIdx.x; e < TL * TD * NQ * 16 * 64; e += blockDim.x) {
    const int ln = e & 63;
    const int r = (e >> 6) & 15;
    const int rest = e >> 10;
    const int q = rest % NQ;
    const int b = (rest / NQ) % TD;
    const int a = rest / (NQ * TD);
    const int leaf = (ltile0 + a) * kTile + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    const int dim = (dtile0 + b) * kTile + (ln & 31);
    if (leaf < L && dim < D) out[(int64_t)leaf * W + 1 + q * D + dim] = acc_lds[e];
  }
  if (lead_dim) {
    for (int e = threadIdx.x; e < TL * 32; e += blockDim.x) {
      const int leaf = (ltile0 + (e >> 5)) * kTile + (e & 31);
      if (leaf < L) out[(int64_t)leaf * W] = s0_lds[e];
    }
  }
}

template <int TL, int TD, bool Q>
static int run_stats(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                     const int32_t* vid, int64_t N, int D, int Dp, int L, int W, double* stats,
                     double* mass, hipStream_t s) {
  const int tilesL = ldphi / kTile, tilesD = (Dp + kTile - 1) / kTile;
  const int gy = (tilesL + TL - 1) / TL, gz = (tilesD + TD - 1) / TD;
  int64_t waves = (int64_t)ctx->num_cu * 8 / ((int64_t)gy * gz);
  if (waves < kStatWaves) waves = kStatWaves;
  int64_t chunk = ceil_div(N, waves);
  chunk = (chunk + 2 * kPairUnroll - 1) / (2 * kPairUnroll) * (2 * kPairUnroll);
  if (chunk < 64) chunk = 64;
  if (chunk > kMaxChunk) chunk = kMaxChunk;
  const int64_t gx = ceil_div(N, chunk * kStatWaves);
  BNPVI_REQUIRE(gx >= 1 && gx < (int64_t(1) << 31), "leaf_stats: grid.x=%lld", (long long)gx);
  const size_t slab_bytes = (size_t)gx * L * W * sizeof(double);
  int rc = ctx->slabs.ensure(slab_bytes);
  if (rc) return rc;
  constexpr int NQ = Q ? 2 : 1;
  const size_t lds = (size_t)(TL * TD * NQ * 16 * 64 + TL * 32) * sizeof(double);
  auto kern = leaf_stats_kernel<TL, TD, Q>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gz);
  const bool timed = ctx->profiling && ctx->timer[BNPVI_PROF_MSTEP].begin(s);
  hipLaunchKernelGGL(kern, grid, dim3(kStatWaves * 64), lds, s, Z, ldz, phi, ldphi, vid, N, D, Dp, L,
                     chunk, (double*)ctx->slabs.ptr, W, mass);
  if (timed) ctx->timer[BNPVI_PROF_MSTEP].end(s);
  BNPVI_LAUNCH_CHECK();
  const int64_t elems = (int64_t)L * W;
  return launch_slab_reduce((const double*)ctx->slabs.ptr, (int)gx, elems, stats, s);
}

int launch_leaf_stats(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                      const int32_t* vid, int64_t N, int D, int L, int V, int want_q,
                      double* stats, double* mass, hipStream_t s) {
  const int Dp = bnpvi_padded_dim(D);
  const int W = 1 + D * (want_q ? 2 : 1);
  if (mass) BNPVI_HIP(hipMemsetAsync(mass, 0, (size_t)V * L * sizeof(double), s));
  if (N == 0) {
    BNPVI_HIP(hipMemsetAsync(stats, 0, (size_t)L * W * sizeof(double), s));
    return BNPVI_OK;
  }
  const int tilesL = ldphi / kTile, tilesD = (Dp + kTile - 1) / kTile;
  const bool td2 = tilesD >= 2;
#define RUN(TL, TD, Q) \
  return run_stats<TL, TD, Q>(ctx, Z, ldz, phi, ldphi, vid, N, D, Dp, L, W, stats, mass, s)
  if (want_q) {
    if (tilesL >= 2) { if (td2) RUN(2, 2, true); else RUN(2, 1, true); }
    else             { if (td2) RUN(1, 2, true); else RUN(1, 1, true); }
  } else {
    if (tilesL >= 4)      { if (td2) RUN(4, 2, false); else RUN(4, 1, false); }
    else if (tilesL >= 2) { if (td2) RUN(2, 2, false); else RUN(2, 1, false); }
    else                  { if (td2) RUN(1, 2, false); else RUN(1, 1, false); }
  }
#undef RUN
}

}  // namespace bnpvi
