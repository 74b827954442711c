// The split statistic of /root/reference/src/var_inf.py:217-220 for the leaves that pass its mass gate only.
//
// split_nodes looks at a leaf's stdev = sqrt(sqrt(sum_n phi d^2) / sum_n phi) only behind
// mean(phi) >= 1e-2, so at most 100 leaves of any tree can need S2 = sum_n phi[n,l] ||z_n - alpha_l||^2 -- none
// on a balanced tree of hundreds of leaves.  The E-step kernels for large trees therefore carry S0' only
// (estep_rows1.h) and S2 comes from here: bnpvi_split_gate lists the leaves whose global mass passes the gate (on
// the device: no host round trip inside an iteration), bnpvi_leaf_s2 makes one pass over Z and over those leaves'
// phi columns.  With nothing listed the data-pass kernel's workgroups return at once.
//
// Data pass.  G = padded D / 4 lanes share a frame (one float4 of the row each).  The listed leaves are taken in
// rounds of G: in a round lane g of the group owns leaf cols[round * G + g]: every lane of the group computes
// its share of ||z - alpha_j||^2 for the round's leaves j in turn (alpha rows from LDS, direct fp32 differences,
// the group's sum by DPP / permlane), the owner keeps its own and adds phi * d^2 to its fp32 accumulator of that
// round.  A workgroup's rows are a contiguous range (2048 frames), its per-leaf sums go to an fp64 slab,
// the slabs are summed in fixed order: bit-reproducible.
#include "common.h"

namespace bnpvi {

constexpr int kS2Cap = 128;           // leaves per call (the reference's gate admits 100)

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// sum over the G lanes that share a row (G = 8 .. 64, aligned lane groups); every lane returns the sum
template <int G>
__device__ __forceinline__ float s2_group_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  if (G >= 16) v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x140, 0xf, 0xf, false));  // row_mirror
  if (G >= 32) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if (G >= 64) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}

// count[0] = leaves listed (<= cap), count[1] = leaves that pass the gate; cols ascending (DFS leaf order)
__global__ __launch_bounds__(256) void split_gate_kernel(const double* __restrict__ estats, int L, double min_mass,
                                                         int32_t* __restrict__ cols, int cap,
                                                         int32_t* __restrict__ count) {
  __shared__ int wave_tot[4];
  __shared__ int base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int l0 = 0; l0 < L; l0 += 256) {
    const int l = l0 + threadIdx.x;
    const bool pass = l < L && estats[2 * l] >= min_mass;
    const unsigned long long m = __ballot(pass);
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
    if (pass && pos < cap) cols[pos] = l;
    __syncthreads();
    if (threadIdx.x == 0) base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    count[0] = base < cap ? base : cap;
    count[1] = base;
  }
}

constexpr int kS2Rows = 2048;         // most frames per workgroup: fp32 partial sums stay short, >= 4 workgroups per CU at 2M frames
constexpr int kS2RowsMin = 256;       // fewest (small frame counts: enough workgroups to fill the chip)

template <int G>
__global__ __launch_bounds__(256) void leaf_s2_kernel(const float* __restrict__ Z, int ldz, int64_t N,
                                                      const float* __restrict__ phi, int ldphi,
                                                      const double* __restrict__ alpha, int D,
                                                      const int32_t* __restrict__ node_of_leaf,
                                                      const int32_t* __restrict__ cols,
                                                      const int32_t* __restrict__ count,
                                                      double* __restrict__ slab /* [workgroups][kS2Cap] */,
                                                      int rows /* frames per workgroup */) {
  constexpr int Dp = 4 * G;
  constexpr int kRounds = kS2Cap / G;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int J = count[0] < kS2Cap ? count[0] : kS2Cap;       // uniform over the grid
  if (J <= 0) return;
  float* a_lds = reinterpret_cast<float*>(smem_raw);         // [J][Dp] fp32 alpha rows of the listed leaves
  float* red = a_lds + (size_t)J * Dp;                       // [256 / G][kS2Cap] per row-group sums
  for (int e = threadIdx.x; e < J * Dp; e += blockDim.x) {
    const int j = e / Dp, d = e % Dp;
    a_lds[e] = d < D ? (float)alpha[(int64_t)node_of_leaf[cols[j]] * D + d] : 0.f;
  }
  __syncthreads();
  const int g = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  constexpr int kGroups = 256 / G;
  int mycol[kRounds];
#pragma unroll
  for (int q = 0; q < kRounds; ++q) mycol[q] = q * G + g < J ? cols[q * G + g] : -1;
  float acc[kRounds];
#pragma unroll
  for (int q = 0; q < kRounds; ++q) acc[q] = 0.f;
  const int64_t r0 = (int64_t)blockIdx.x * rows;
  const int64_t r1 = r0 + rows < N ? r0 + rows : N;
  // four rows of the group at a time, every load of the four issued before the first is used (one row at a time
  // left one 16-byte load per lane in flight: 3.3 TB/s); the sums run over the rows in the same order as before
  constexpr int kU = 4;
  for (int64_t n = r0 + grp; n < r1; n += kU * kGroups) {
    float4 z[kU];
    float w[kU][kRounds];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t nn = n + u * kGroups;
      const bool ok = nn < r1;
      const int64_t row = ok ? nn : r1 - 1;
      z[u] = *reinterpret_cast<const float4*>(Z + row * ldz + 4 * g);
      const float* prow = phi + row * ldphi;
#pragma unroll
      for (int q = 0; q < kRounds; ++q) {
        w[u][q] = 0.f;
        if (q * G < J) {                                       // uniform
          const float v = mycol[q] >= 0 ? prow[mycol[q]] : 0.f;
          w[u][q] = ok ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (n + u * kGroups >= r1) break;                        // uniform over the row group's lanes
#pragma unroll
      for (int q = 0; q < kRounds; ++q) {
        if (q * G < J) {                                       // uniform
          float mine = 0.f;
          const int jn = J - q * G < G ? J - q * G : G;
          for (int j = 0; j < jn; ++j) {
            const float4 m = *reinterpret_cast<const float4*>(a_lds + (size_t)(q * G + j) * Dp + 4 * g);
            const float dx = z[u].x - m.x, dy = z[u].y - m.y, dz = z[u].z - m.z, dw = z[u].w - m.w;
            const float d2 = s2_group_sum<G>(dx * dx + dy * dy + dz * dz + dw * dw);
            if (g == j) mine = d2;
          }
          acc[q] = fmaf(w[u][q], mine, acc[q]);
        }
      }
    }
  }
  // combine the row groups of the workgroup in fixed order, one fp64 slab per workgroup
#pragma unroll
  for (int q = 0; q < kRounds; ++q) red[grp * kS2Cap + q * G + g] = acc[q];
  __syncthreads();
  for (int j = threadIdx.x; j < kS2Cap; j += blockDim.x) {
    double t = 0.0;
    if (j < J)
      for (int w = 0; w < kGroups; ++w) t += (double)red[w * kS2Cap + j];
    slab[(int64_t)blockIdx.x * kS2Cap + j] = t;
  }
}

// s2[j] = sum over the workgroups' slabs in ascending order (j < count), 0 for the unlisted tail
__global__ __launch_bounds__(256) void leaf_s2_final_kernel(const double* __restrict__ slab, int nslab,
                                                            const int32_t* __restrict__ count, int cap,
                                                            double* __restrict__ s2) {
  __shared__ double sh[256];
  const int j = blockIdx.x;
  if (j >= cap) return;
  const int J = count[0] < kS2Cap ? count[0] : kS2Cap;
  double t = 0.0;
  if (j < J)
    for (int b = threadIdx.x; b < nslab; b += 256) t += slab[(int64_t)b * kS2Cap + j];
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) s2[j] = sh[0];
}

int launch_split_gate(bnpvi_ctx* ctx, const double* estats, int L, double min_mass, int32_t* cols, int cap,
                      int32_t* count, hipStream_t s) {
  (void)ctx;
  hipLaunchKernelGGL(split_gate_kernel, dim3(1), dim3(256), 0, s, estats, L, min_mass, cols, cap, count);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

int launch_leaf_s2(bnpvi_ctx* ctx, const bnpvi_tree* t, const float* Z, int ldz, int64_t N, int D,
                   const double* alpha, const float* phi, int ldphi, const int32_t* cols, const int32_t* count,
                   int cap, double* s2, hipStream_t s) {
  const int Dp = bnpvi_padded_dim(D);
  const int G = Dp / 4;
  // frames per workgroup: about four workgroups per CU, between kS2RowsMin and kS2Rows (a function of N and the
  // chip only, so the summation order is reproducible)
  int64_t rows = ceil_div(N > 0 ? N : 1, (int64_t)4 * ctx->num_cu);
  rows = rows < kS2RowsMin ? kS2RowsMin : (rows > kS2Rows ? kS2Rows : rows);
  const int64_t blocks = N > 0 ? ceil_div(N, rows) : 0;
  double* slab = nullptr;
  if (blocks > 0) {
    int rc = slab_workspace(ctx, (size_t)blocks * kS2Cap * sizeof(double), &slab);
    if (rc) return rc;
    // alpha rows of up to kS2Cap leaves + the row groups' partial sums
    const size_t lds = (size_t)kS2Cap * Dp * sizeof(float) + (size_t)(256 / G) * kS2Cap * sizeof(float);
#define BNPVI_S2(GG)                                                                                          \
  do {                                                                                                        \
    auto k = leaf_s2_kernel<GG>;                                                                              \
    if (lds > 48 * 1024)                                                                                      \
      BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, s, Z, ldz, N, phi, ldphi, alpha, D,         \
                       t->d_node_of_leaf, cols, count, slab, (int)rows);                                            \
  } while (0)
    switch (G) {
      case 8: BNPVI_S2(8); break;
      case 16: BNPVI_S2(16); break;
      case 32: BNPVI_S2(32); break;
      default: BNPVI_S2(64); break;
    }
#undef BNPVI_S2
    BNPVI_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(leaf_s2_final_kernel, dim3((unsigned)cap), dim3(256), 0, s, slab, (int)blocks, count, cap, s2);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

}  // namespace bnpvi

using namespace bnpvi;

extern "C" {

int bnpvi_split_max_columns(void) { return kS2Cap; }

int bnpvi_split_gate(bnpvi_ctx* ctx, const double* estats, int L, double min_mass, int32_t* cols, int cap,
                     int32_t* count, void* stream) {
  BNPVI_REQUIRE(ctx && estats && cols && count, "bnpvi_split_gate: NULL argument");
  BNPVI_REQUIRE(L >= 1 && cap >= 1 && cap <= kS2Cap, "bnpvi_split_gate: L=%d cap=%d (1..%d)", L, cap, kS2Cap);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_split_gate(ctx, estats, L, min_mass, cols, cap, count, (hipStream_t)stream);
}

int bnpvi_leaf_s2(bnpvi_ctx* ctx, const bnpvi_tree* tree, const float* Z, int ldz, int64_t N, int D,
                  const double* alpha, const float* phi, int ldphi, const int32_t* cols, const int32_t* count,
                  int cap, double* s2, void* stream) {
  BNPVI_REQUIRE(ctx && tree && alpha && cols && count && s2, "bnpvi_leaf_s2: NULL argument");
  BNPVI_REQUIRE(N == 0 || (Z && phi), "bnpvi_leaf_s2: NULL data pointer");
  BNPVI_REQUIRE(N >= 0 && N < (int64_t(1) << 33), "bnpvi_leaf_s2: N=%lld", (long long)N);
  const int Dp = bnpvi_padded_dim(D);
  BNPVI_REQUIRE(D >= 1 && Dp > 0 && ldz >= Dp && ldz % 4 == 0, "bnpvi_leaf_s2: D=%d ldz=%d", D, ldz);
  BNPVI_REQUIRE(ldphi == tree->Lp, "bnpvi_leaf_s2: ldphi=%d must equal the tree's padded leaf count %d", ldphi,
                tree->Lp);
  BNPVI_REQUIRE(cap >= 1 && cap <= kS2Cap, "bnpvi_leaf_s2: cap=%d (1..%d)", cap, kS2Cap);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_leaf_s2(ctx, tree, Z, ldz, N, D, alpha, phi, ldphi, cols, count, cap, s2, (hipStream_t)stream);
}

}  // extern "C"
