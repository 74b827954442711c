// E-step host dispatch and the DP-independent kernels (operand packing, the normalisation
// pass of the streamed regime, slab reduction).  The MFMA kernels live in estep_kernels.h and
// are instantiated per padded latent dim in estep_inst.hip.
#include "estep_kernels.h"

namespace bnpvi {

// ------------------------------------------------------------------------------------------
// Leaf operands.  Step 1 (one wave per padded leaf): fp32 row-major copy a32[Lp][DP] of the
// leaves' alpha (fp64, [K][D], pre-order), ||alpha||^2 of the fp32-rounded values, and the
// global max |alpha| (atomicMax on the bit pattern; non-negative floats order like ints).
// Step 2 (one thread per 16-byte fragment): scaled exact hi/lo f16 split in MFMA operand
// order: frag[((T*S+s)*2+part)*64+lane][e] = alpha[leaf 16T+(lane&15)][dim 32s+16(e>>2)+4(lane>>4)+(e&3)].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void leaf_rows_kernel(const double* __restrict__ alpha, int D, int DP,
                                                        const int32_t* __restrict__ node_of_leaf, int L,
                                                        int Lp, float* __restrict__ a32,
                                                        float* __restrict__ aavec,
                                                        int* __restrict__ amax_bits) {
  // one wave per padded leaf, lanes stride the dims
  const int lane = threadIdx.x & 63;
  const int leaf = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (leaf >= Lp) return;
  double aa = 0.0;
  float amax = 0.f;
  const double* a = leaf < L ? alpha + (int64_t)node_of_leaf[leaf] * D : nullptr;
  for (int d = lane; d < DP; d += 64) {
    const float f = (a && d < D) ? (float)a[d] : 0.f;
    a32[(int64_t)leaf * DP + d] = f;
    aa += (double)f * (double)f;
    amax = fmaxf(amax, fabsf(f));
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    aa += __shfl_xor(aa, off);
    amax = fmaxf(amax, __shfl_xor(amax, off));
  }
  if (lane == 0) {
    aavec[leaf] = (float)aa;
    atomicMax(amax_bits, __float_as_int(amax));
    atomicMax(amax_bits + 1, __float_as_int((float)aa));   // non-negative floats order like ints
  }
}

__global__ void pack_leaves_kernel(const float* __restrict__ a32, int DP, int Lp,
                                   const int* __restrict__ amax_bits, f16x8* __restrict__ apack,
                                   float* __restrict__ scal) {
  const int S = DP / 32;
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int ka = scale_exponent_bits(*amax_bits);
  if (u == 0) scal[0] = pow2f(-ka);
  if (u >= (int64_t)(Lp / kLT) * S * 2 * 64) return;
  const int lane = (int)(u & 63);
  const int part = (int)((u >> 6) & 1);
  const int64_t ts = u >> 7;
  const int s = (int)(ts % S);
  const int T = (int)(ts / S);
  const int leaf = kLT * T + (lane & 15);
  const int kg = lane >> 4;
  const float sa = pow2f(ka);
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

// Small trees: rows, norms, scale and packed fragments in one workgroup (leaf_operands_block,
// estep_kernels.h); also the tail of the fused mid-step kernel (tree_kernels.hip).
__global__ __launch_bounds__(1024) void leaf_operands_small_kernel(
    const double* __restrict__ alpha, int D, int DP, const int32_t* __restrict__ node_of_leaf, int L, int Lp,
    float* a32, float* __restrict__ aavec, f16x8* __restrict__ apack, float* __restrict__ scal) {
  leaf_operands_block(alpha, D, DP, node_of_leaf, L, Lp, a32, aavec, apack, scal);
}

// ------------------------------------------------------------------------------------------
// Streamed regime, pass B: combine the partials of a frame, normalise its row in place,
// accumulate S0' / S2.  One wave per row at a time, lanes stride the row in float4.
// Statistics: a leaf's column is always taken by the same lane of a wave, so every wave keeps its own
// fp32 table [Lp][2] in LDS (plain read-modify-write, rows in the wave's fixed order); the workgroup's waves are
// added in order into one fp64 slab per workgroup, the slabs by slab_reduce -- no atomics, same bits every run
// (the fused, row and column-window epilogues were already built that way).
// ------------------------------------------------------------------------------------------
template <bool HARD>
__global__ __launch_bounds__(256) void estep_normalize_kernel(
    float* __restrict__ phi, float* __restrict__ logphi, int32_t* __restrict__ assign,
    const float2* __restrict__ part, int nchunk, const float* __restrict__ prior,
    const int32_t* __restrict__ vid, int64_t N, int L, int Lp, float inv_half_sz,
    double* __restrict__ eslab /* [gridDim.x][L][2] one slab per workgroup, or null */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* st = reinterpret_cast<float*>(smem_raw);        // [waves][Lp][2] fp32 partials, one table per wave
  const bool want_stats = eslab != nullptr;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  if (want_stats) {
    for (int e = threadIdx.x; e < nw * 2 * Lp; e += blockDim.x) st[e] = 0.f;
    __syncthreads();
  }
  float* stw = st + (size_t)wave * 2 * Lp;
  for (int64_t n = (int64_t)blockIdx.x * nw + wave; n < N; n += (int64_t)gridDim.x * nw) {
    // combine partials (every lane redundantly walks a strided share, then wave-reduces)
    float m = -INFINITY;
    int best = 0x7fffffff;
    for (int c = lane; c < nchunk; c += 64) {
      const float2 p = part[n * nchunk + c];
      const int b = __float_as_int(p.y);
      if (p.x > m || (HARD && p.x == m && b < best)) {
        m = p.x;
        best = b;
      }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float om = __shfl_xor(m, off);
      const int ob = __shfl_xor(best, off);
      if (om > m || (HARD && om == m && ob < best)) {
        m = om;
        best = ob;
      }
    }
    float lse = 0.f;
    if (!HARD) {
      float ssum = 0.f;
      for (int c = lane; c < nchunk; c += 64) {
        const float2 p = part[n * nchunk + c];
        if (p.x != -INFINITY) ssum += p.y * __builtin_amdgcn_exp2f((p.x - m) * kLog2e);
      }
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) ssum += __shfl_xor(ssum, off);
      lse = __builtin_amdgcn_logf(ssum) * kLn2;
    }
    float* row = phi + n * Lp;
    float* lrow = logphi ? logphi + n * Lp : nullptr;
    const float* prow = HARD ? nullptr : prior + (int64_t)vid[n] * Lp;
    int amax = 0x7fffffff;
    float vmax = -INFINITY;
    for (int c4 = lane * 4; c4 < Lp; c4 += 256) {
      const float4 x = *reinterpret_cast<const float4*>(row + c4);
      const float xv[4] = {x.x, x.y, x.z, x.w};
      float pv[4], lv[4];
      float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!HARD) pr = *reinterpret_cast<const float4*>(prow + c4);
      const float prv[4] = {pr.x, pr.y, pr.z, pr.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (HARD) {
          pv[c] = (c4 + c == best) ? 1.f : 0.f;
          lv[c] = (c4 + c == best) ? 0.f : -INFINITY;
        } else {
          lv[c] = (xv[c] - m) - lse;
          pv[c] = __builtin_amdgcn_exp2f(lv[c] * kLog2e);
          if (xv[c] > vmax) {
            vmax = xv[c];
            amax = c4 + c;
          }
        }
        if (want_stats && c4 + c < L) {
          float dd;
          if (HARD)
            dd = -xv[c];                                   // logit was -d^2
          else {
            const float d = (prv[c] * kLn2 - xv[c]) * inv_half_sz;  // logit = prior*ln2 - half_sz*d
            dd = d * d;
          }
          if (pv[c] != 0.f) {
            stw[2 * (c4 + c)] += pv[c];
            stw[2 * (c4 + c) + 1] += pv[c] * dd;
          }
        }
      }
      *reinterpret_cast<float4*>(row + c4) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      if (lrow) *reinterpret_cast<float4*>(lrow + c4) = make_float4(lv[0], lv[1], lv[2], lv[3]);
    }
    if (assign) {
      if (HARD) {
        if (lane == 0) assign[n] = best;
      } else {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float om = __shfl_xor(vmax, off);
          const int ob = __shfl_xor(amax, off);
          if (om > vmax || (om == vmax && ob < amax)) {
            vmax = om;
            amax = ob;
          }
        }
        if (lane == 0) assign[n] = amax;
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    double* out = eslab + (int64_t)blockIdx.x * 2 * L;
    for (int e = threadIdx.x; e < 2 * L; e += blockDim.x) {
      double t = 0.0;
      for (int w = 0; w < nw; ++w) t += (double)st[(size_t)w * 2 * Lp + e];
      out[e] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Streamed regime, pass B for the production call (soft mode, no arg-max), in two kernels with no
// atomics: (1) per row, combine the chunk partials into (max, log-sum-exp); (2) normalise the
// matrix by COLUMN WINDOWS -- a lane owns 4 fixed leaves of a 256-leaf window and walks a block
// of rows, so the statistics S0' / S2 of its leaves stay in registers; the 4 waves of a
// workgroup are combined in fixed order and written as one fp64 slab per row block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void estep_rowstat_kernel(const float2* __restrict__ part, int nchunk,
                                                            int64_t N, float2* __restrict__ rowstat) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float m = -INFINITY;
  for (int c = lane; c < nchunk; c += 64) m = fmaxf(m, part[n * nchunk + c].x);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
  float ssum = 0.f;
  for (int c = lane; c < nchunk; c += 64) {
    const float2 p = part[n * nchunk + c];
    if (p.x != -INFINITY) ssum += p.y * __builtin_amdgcn_exp2f((p.x - m) * kLog2e);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) ssum += __shfl_xor(ssum, off);
  if (lane == 0) rowstat[n] = make_float2(m, __builtin_amdgcn_logf(ssum) * kLn2);
}

constexpr int kNormRows = 512;      // rows per workgroup (4 waves x 128)

template <bool STATS>
__global__ __launch_bounds__(256) void estep_normalize_cols_kernel(
    float* __restrict__ phi, float* __restrict__ logphi, const float2* __restrict__ rowstat,
    const float* __restrict__ prior, const int32_t* __restrict__ vid, int64_t N, int L, int Lp,
    float inv_half_sz, double* __restrict__ eslab /* [row blocks][L][2] */) {
  __shared__ float comb[4][256][8];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c4 = blockIdx.y * 256 + lane * 4;            // this lane's 4 leaves
  const bool col_ok = c4 < Lp;                           // Lp is a multiple of 4
  const int64_t r0 = (int64_t)blockIdx.x * kNormRows;
  const int64_t r1 = r0 + kNormRows < N ? r0 + kNormRows : N;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (col_ok) {
#pragma unroll 2
    for (int64_t n = r0 + wave; n < r1; n += 4) {
      const float2 rs = rowstat[n];
      const float4 x = *reinterpret_cast<const float4*>(phi + n * Lp + c4);
      float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (STATS) pr = *reinterpret_cast<const float4*>(prior + (int64_t)vid[n] * Lp + c4);
      const float xv[4] = {x.x, x.y, x.z, x.w};
      const float prv[4] = {pr.x, pr.y, pr.z, pr.w};
      float pv[4], lv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        lv[c] = (xv[c] - rs.x) - rs.y;
        pv[c] = __builtin_amdgcn_exp2f(lv[c] * kLog2e);
        if (STATS) {
          const float d = (prv[c] * kLn2 - xv[c]) * inv_half_sz;     // logit = prior*ln2 - half_sz*d
          s0[c] += pv[c];
          s2[c] += pv[c] != 0.f ? pv[c] * (d * d) : 0.f;             // padded leaves: 0 * inf
        }
      }
      *reinterpret_cast<float4*>(phi + n * Lp + c4) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      if (logphi) *reinterpret_cast<float4*>(logphi + n * Lp + c4) = make_float4(lv[0], lv[1], lv[2], lv[3]);
    }
  }
  if (STATS) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      comb[wave][lane][c] = s0[c];
      comb[wave][lane][4 + c] = s2[c];
    }
    __syncthreads();
    // thread t of the workgroup owns leaf (window, t): lane t/4, component t%4
    const int t = threadIdx.x;
    const int leaf = blockIdx.y * 256 + t;
    if (leaf < L) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += (double)comb[w][t >> 2][t & 3];
        b += (double)comb[w][t >> 2][4 + (t & 3)];
      }
      double* out = eslab + ((int64_t)blockIdx.x * L + leaf) * 2;
      out[0] = a;
      out[1] = b;
    }
  }
}

int estep_reduce_slabs(const EstepArgs& a, const double* eslab, int nslab) {
  const int elems = 2 * a.tree->L;
  return launch_slab_reduce(eslab, nslab, elems, a.estats, a.stream);
}

int estep_finish_streamed(const EstepArgs& a, const float2* part, int nchunk) {
  const int L = a.tree->L, Lp = a.tree->Lp;
  if (a.mode == BNPVI_ESTEP_SOFT && a.assign == nullptr) {
    float2* rowstat = const_cast<float2*>(part) + (size_t)a.N * nchunk;     // room reserved by run_streamed
    BNPVI_REQUIRE(ceil_div(a.N, 4) <= 0x7fffffffLL, "estep: N=%lld exceeds the streamed regime's grid", (long long)a.N);
    hipLaunchKernelGGL(estep_rowstat_kernel, dim3((unsigned)ceil_div(a.N, 4)), dim3(256), 0, a.stream, part,
                       nchunk, a.N, rowstat);
    BNPVI_LAUNCH_CHECK();
    // row blocks on gridDim.x (2^31 - 1 of them: N < 2^31 * kNormRows), the leaf windows on gridDim.y
    const int64_t rblocks = ceil_div(a.N, kNormRows);
    BNPVI_REQUIRE(rblocks <= 0x7fffffffLL && ceil_div(Lp, 256) <= 65535, "estep: %lld row blocks, %d leaf windows",
                  (long long)rblocks, (int)ceil_div(Lp, 256));
    dim3 grid((unsigned)rblocks, (unsigned)ceil_div(Lp, 256));
    if (a.estats) {
      int rc = a.ctx->slabs.ensure((size_t)rblocks * 2 * L * sizeof(double));
      if (rc) return rc;
      double* eslab = (double*)a.ctx->slabs.ptr;
      hipLaunchKernelGGL(estep_normalize_cols_kernel<true>, grid, dim3(256), 0, a.stream, a.phi, a.logphi,
                         rowstat, a.prior, a.vid, a.N, L, Lp, 1.f / a.half_sz, eslab);
      BNPVI_LAUNCH_CHECK();
      return estep_reduce_slabs(a, eslab, (int)rblocks);
    }
    hipLaunchKernelGGL(estep_normalize_cols_kernel<false>, grid, dim3(256), 0, a.stream, a.phi, a.logphi,
                       rowstat, a.prior, a.vid, a.N, L, Lp, 1.f / a.half_sz, (double*)nullptr);
    BNPVI_LAUNCH_CHECK();
    return BNPVI_OK;
  }
  // hard responsibilities, or the arg-max asked for: the general kernel.  With statistics every wave owns a table of
  // Lp x 2 floats in LDS, so large trees run fewer waves per workgroup (and one workgroup per CU).
  int nw = 4;
  size_t lds = 0;
  int per_cu = 8;
  if (a.estats) {
    const size_t table = (size_t)Lp * 8;
    BNPVI_REQUIRE(table <= (size_t)kMaxLdsBytes, "estep: L=%d too large for the stats epilogue", L);
    nw = (int)((size_t)kMaxLdsBytes / table);
    nw = nw > 4 ? 4 : nw;
    lds = (size_t)nw * table;
    per_cu = (int)((size_t)kMaxLdsBytes / lds);
    per_cu = per_cu > 8 ? 8 : per_cu;
  }
  int64_t nb = ceil_div(a.N, nw);
  int64_t capb = (int64_t)a.ctx->num_cu * per_cu;
  const int64_t slab_cap = ((int64_t)64 << 20) / ((int64_t)2 * L * (int64_t)sizeof(double));   // slabs <= 64 MiB
  if (a.estats && capb > slab_cap) capb = slab_cap > 0 ? slab_cap : 1;
  if (nb > capb) nb = capb;
  double* eslab = nullptr;
  if (a.estats) {
    int rc = a.ctx->slabs.ensure((size_t)nb * 2 * L * sizeof(double));
    if (rc) return rc;
    eslab = (double*)a.ctx->slabs.ptr;
  }
  if (a.mode == BNPVI_ESTEP_HARD) {
    auto k = estep_normalize_kernel<true>;
    if (lds > 48 * 1024)
      BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(64 * nw), lds, a.stream, a.phi, a.logphi, a.assign, part,
                       nchunk, a.prior, a.vid, a.N, L, Lp, 1.f / a.half_sz, eslab);
  } else {
    auto k = estep_normalize_kernel<false>;
    if (lds > 48 * 1024)
      BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(64 * nw), lds, a.stream, a.phi, a.logphi, a.assign, part,
                       nchunk, a.prior, a.vid, a.N, L, Lp, 1.f / a.half_sz, eslab);
  }
  BNPVI_LAUNCH_CHECK();
  if (a.estats) return estep_reduce_slabs(a, eslab, (int)nb);
  return BNPVI_OK;
}

template <> int estep_dispatch<32>(const EstepArgs&);
template <> int estep_dispatch<64>(const EstepArgs&);
template <> int estep_dispatch<128>(const EstepArgs&);
template <> int estep_dispatch<256>(const EstepArgs&);

int leaf_workspace(bnpvi_ctx* ctx, const bnpvi_tree* t, int DP, LeafWorkspace* w) {
  // the streamed regime reads whole chunks of the packed operands: pad to a chunk of tiles
  const int LpPack = round_up(t->Lp, kLT * kStreamNT);
  // workspace: [apack f16 hi/lo: LpPack*DP*4 B][a32: LpPack*DP*4 B][aa: LpPack*4 B][scal 16 B]
  const size_t img = (size_t)LpPack * DP * 4;
  int rc = ctx->pack.ensure(2 * img + (size_t)LpPack * 4 + 16);
  if (rc) return rc;
  char* base = (char*)ctx->pack.ptr;
  w->LpPack = LpPack;
  w->apack = (f16x8*)base;
  w->a32 = (float*)(base + img);
  w->aavec = (float*)(base + 2 * img);
  w->scal = w->aavec + LpPack;      // [0] = 2^-ka, [1] = bits of max|alpha|, [2] = max ||alpha||^2
  return BNPVI_OK;
}

int launch_estep(bnpvi_ctx* ctx, const bnpvi_tree* t, const float* Z, int ldz,
                 const int32_t* vid, int64_t N, int D, const double* alpha, const float* prior,
                 double sZ, int mode, float* phi, float* logphi, int32_t* assign, double* estats,
                 hipStream_t s) {
  const int DP = bnpvi_padded_dim(D);
  LeafWorkspace w;
  int rc = leaf_workspace(ctx, t, DP, &w);
  if (rc) return rc;
  const int LpPack = w.LpPack;
  f16x8* apack = w.apack;
  float *a32 = w.a32, *aavec = w.aavec, *scal = w.scal;
  int* amax_bits = (int*)(scal + 1);
  const bool ready = (mode & BNPVI_ESTEP_OPERANDS_READY) && ctx->ops_tree == t && ctx->ops_D == D &&
                     ctx->ops_alpha == alpha;
  const bool s0_only = (mode & BNPVI_ESTEP_STATS_S0) != 0;
  mode &= ~(BNPVI_ESTEP_OPERANDS_READY | BNPVI_ESTEP_STATS_S0);
  ctx->ops_tree = nullptr;                      // one use: alpha may change before the next call
  if (ready) {
    // packed by bnpvi_params_prior from this very alpha buffer
  } else
  if ((int64_t)LpPack * DP <= 64 * 1024) {
    hipLaunchKernelGGL(leaf_operands_small_kernel, dim3(1), dim3(1024), 0, s, alpha, D, DP, t->d_node_of_leaf,
                       t->L, LpPack, a32, aavec, apack, scal);
    BNPVI_LAUNCH_CHECK();
  } else {
    BNPVI_HIP(hipMemsetAsync(amax_bits, 0, 2 * sizeof(int), s));
    hipLaunchKernelGGL(leaf_rows_kernel, dim3((unsigned)ceil_div(LpPack, 4)), dim3(256), 0, s, alpha, D, DP,
                       t->d_node_of_leaf, t->L, LpPack, a32, aavec, amax_bits);
    BNPVI_LAUNCH_CHECK();
    const int64_t units = (int64_t)(LpPack / kLT) * (DP / 32) * 2 * 64;
    hipLaunchKernelGGL(pack_leaves_kernel, dim3((unsigned)ceil_div(units, 256)), dim3(256), 0, s, a32, DP,
                       LpPack, amax_bits, apack, scal);
    BNPVI_LAUNCH_CHECK();
  }
  if (N == 0) {
    if (estats) BNPVI_HIP(hipMemsetAsync(estats, 0, (size_t)2 * t->L * sizeof(double), s));
    return BNPVI_OK;
  }
  EstepArgs a{ctx, t, Z, ldz, vid, N, apack, aavec, a32, scal, prior, (float)(0.5 * sZ), mode,
              phi, logphi, assign, estats, s, s0_only};
  switch (DP) {
    case 32: return estep_dispatch<32>(a);
    case 64: return estep_dispatch<64>(a);
    case 128: return estep_dispatch<128>(a);
    case 256: return estep_dispatch<256>(a);
  }
  set_error("estep: unsupported padded dim %d", DP);
  return BNPVI_EUNSUPPORTED;
}

}  // namespace bnpvi
