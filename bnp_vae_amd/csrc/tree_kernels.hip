// Small per-node / per-video kernels of the VI step on the flattened tree (gfx950):
//   node parameter update     /root/reference/src/var_inf.py:265-297
//   gamma (stick) update      /root/reference/src/var_inf.py:395-416
//   stick log-prior table     /root/reference/src/var_inf.py:303-311, 359-381
//   weighted k-means seeding  /root/reference/src/var_inf.py:23-36
//   phi column surgery        /root/reference/src/var_inf.py:196, 232
// All of this is fp64 (the reference is NumPy fp64) except where noted; none of it is on the
// N*L*D critical path, so the kernels favour exactness and simplicity.
#include <math.h>
#include <string.h>

#include "common.h"
#include "estep_kernels.h"

namespace bnpvi {

// ------------------------------------------------------------------------------------------
// Node parameters: one launch per tree level, deepest level first.  A node at depth d reads
// alpha[parent] (depth d-1, not yet rewritten => the OLD value, as in the reference's
// post-order recursion) and alpha[children] (depth d+1, already rewritten => NEW values).
// ------------------------------------------------------------------------------------------
__global__ void node_params_level_kernel(const int32_t* __restrict__ level_nodes, int count,
                                         const int32_t* __restrict__ parent,
                                         const int32_t* __restrict__ leaf_of_node,
                                         const int32_t* __restrict__ child_ptr,
                                         const int32_t* __restrict__ child_idx,
                                         const double* __restrict__ stats, int W, int D, double sB,
                                         double sZ, const double* __restrict__ alpha0,
                                         double* alpha, double* __restrict__ sigma) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)count * D) return;
  const int k = level_nodes[e / D];
  const int d = (int)(e % D);
  const int p = parent[k];
  const double base = (p < 0) ? (alpha0 ? alpha0[d] : 0.0) : alpha[(int64_t)p * D + d];
  const int l = leaf_of_node[k];
  double s, a;
  if (l >= 0) {
    const double S0 = stats[(int64_t)l * W];
    const double S1 = stats[(int64_t)l * W + 1 + d];
    s = sB + S0 * sZ;
    a = (base * sB + S1 * sZ) / s;
  } else {
    const int c0 = child_ptr[k], c1 = child_ptr[k + 1];
    s = (1.0 + (double)(c1 - c0)) * sB;
    double acc = base;
    for (int c = c0; c < c1; ++c) acc += alpha[(int64_t)child_idx[c] * D + d];
    a = acc * sB / s;
  }
  alpha[(int64_t)k * D + d] = a;
  if (d == 0) sigma[k] = s;
}

// Small trees: every level in ONE launch by one workgroup (a level's writes are visible to the
// next level's reads through the workgroup barrier; alpha is not __restrict__ here).
__device__ __forceinline__ void node_params_tree_block(
    const int32_t* __restrict__ level_nodes, const int32_t* __restrict__ level_ptr, int nlevels,
    const int32_t* __restrict__ parent, const int32_t* __restrict__ leaf_of_node,
    const int32_t* __restrict__ child_ptr, const int32_t* __restrict__ child_idx,
    const double* __restrict__ stats, int W, int D, double sB, double sZ,
    const double* __restrict__ alpha0, double* alpha, double* __restrict__ sigma) {
  for (int lev = nlevels - 1; lev >= 0; --lev) {
    const int off = level_ptr[lev];
    const int count = level_ptr[lev + 1] - off;
    for (int e = threadIdx.x; e < count * D; e += blockDim.x) {
      const int k = level_nodes[off + e / D];
      const int d = e % D;
      const int p = parent[k];
      const double base = (p < 0) ? (alpha0 ? alpha0[d] : 0.0) : alpha[(int64_t)p * D + d];
      const int l = leaf_of_node[k];
      double s, a;
      if (l >= 0) {
        const double S0 = stats[(int64_t)l * W];
        const double S1 = stats[(int64_t)l * W + 1 + d];
        s = sB + S0 * sZ;
        a = (base * sB + S1 * sZ) / s;
      } else {
        const int c0 = child_ptr[k], c1 = child_ptr[k + 1];
        s = (1.0 + (double)(c1 - c0)) * sB;
        double acc = base;
        for (int c = c0; c < c1; ++c) acc += alpha[(int64_t)child_idx[c] * D + d];
        a = acc * sB / s;
      }
      alpha[(int64_t)k * D + d] = a;
      if (d == 0) sigma[k] = s;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void node_params_tree_kernel(
    const int32_t* __restrict__ level_nodes, const int32_t* __restrict__ level_ptr, int nlevels,
    const int32_t* __restrict__ parent, const int32_t* __restrict__ leaf_of_node,
    const int32_t* __restrict__ child_ptr, const int32_t* __restrict__ child_idx,
    const double* __restrict__ stats, int W, int D, double sB, double sZ,
    const double* __restrict__ alpha0, double* alpha, double* __restrict__ sigma) {
  node_params_tree_block(level_nodes, level_ptr, nlevels, parent, leaf_of_node, child_ptr, child_idx, stats, W,
                         D, sB, sZ, alpha0, alpha, sigma);
}

int launch_node_params(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* stats, int W, int D,
                       double sB, double sZ, const double* alpha0, double* alpha, double* sigma,
                       hipStream_t s) {
  (void)ctx;
  if ((int64_t)t->K * D <= 64 * 1024) {
    hipLaunchKernelGGL(node_params_tree_kernel, dim3(1), dim3(1024), 0, s, t->d_level_nodes, t->d_level_ptr,
                       t->nlevels, t->d_parent, t->d_leaf_of_node, t->d_child_ptr, t->d_child_idx, stats,
                       W, D, sB, sZ, alpha0, alpha, sigma);
    BNPVI_LAUNCH_CHECK();
    return BNPVI_OK;
  }
  for (int lev = t->nlevels - 1; lev >= 0; --lev) {
    const int off = t->level_ptr[lev];
    const int count = t->level_ptr[lev + 1] - off;
    if (count == 0) continue;
    const int64_t work = (int64_t)count * D;
    hipLaunchKernelGGL(node_params_level_kernel, dim3((unsigned)ceil_div(work, 256)), dim3(256), 0, s,
                       t->d_level_nodes + off, count, t->d_parent, t->d_leaf_of_node, t->d_child_ptr,
                       t->d_child_idx, stats, W, D, sB, sZ, alpha0, alpha, sigma);
    BNPVI_LAUNCH_CHECK();
  }
  return BNPVI_OK;
}

// ------------------------------------------------------------------------------------------
// In-place inclusive scan of a[0..n) (fp64, LDS) by one workgroup.  tmp: blockDim.x doubles.
// ------------------------------------------------------------------------------------------
__device__ void block_inclusive_scan(double* a, int n, double* tmp) {
  const int T = blockDim.x, t = threadIdx.x;
  const int seg = (n + T - 1) / T;
  const int b = t * seg, e = (b + seg < n) ? b + seg : n;
  double run = 0.0;
  for (int j = b; j < e; ++j) {
    run += a[j];
    a[j] = run;
  }
  tmp[t] = run;
  __syncthreads();
  for (int off = 1; off < T; off <<= 1) {
    double v = (t >= off) ? tmp[t - off] : 0.0;
    __syncthreads();
    tmp[t] += v;
    __syncthreads();
  }
  const double add = (t > 0) ? tmp[t - 1] : 0.0;
  for (int j = b; j < e; ++j) a[j] += add;
  __syncthreads();
}

template <bool DO_GAMMA, bool DO_PRIOR>
static int launch_stick_wave(const bnpvi_tree* t, const double* mass, double* gamma, const double* sigma,
                             int V, double gamma_hyper, double sZ, float* prior, double* on, double* before,
                             hipStream_t s);
constexpr int kWaveStickMaxK = 1024;   // trees up to this many nodes: one wave per video (below)

// gamma[v][k] = (1 + mass of k's leaves, GAMMA + mass of all leaves before k's first leaf).
// The reference's recursion (src/var_inf.py:395-413) hands every child "before(parent) + mass
// of earlier siblings"; with leaves in DFS order that is the exclusive prefix at leaf_lo[k].
__global__ void gamma_kernel(const double* __restrict__ mass, int L, int K,
                             const int32_t* __restrict__ leaf_lo,
                             const int32_t* __restrict__ leaf_hi, double gamma_hyper,
                             double* __restrict__ gamma) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* P = reinterpret_cast<double*>(smem_raw);   // [L] inclusive prefix
  double* tmp = P + L;                               // [blockDim.x]
  const int v = blockIdx.x;
  for (int l = threadIdx.x; l < L; l += blockDim.x) P[l] = mass[(int64_t)v * L + l];
  __syncthreads();
  block_inclusive_scan(P, L, tmp);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const int lo = leaf_lo[k], hi = leaf_hi[k];
    const double before = lo > 0 ? P[lo - 1] : 0.0;
    const double upto = hi > 0 ? P[hi - 1] : 0.0;
    double2 g;
    g.x = 1.0 + (upto - before);
    g.y = gamma_hyper + before;
    reinterpret_cast<double2*>(gamma)[(int64_t)v * K + k] = g;
  }
}

int launch_gamma_update(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* mass, int V,
                        double gamma_hyper, double* gamma, hipStream_t s) {
  (void)ctx;
  if (t->K <= kWaveStickMaxK)
    return launch_stick_wave<true, false>(t, mass, gamma, nullptr, V, gamma_hyper, 0.0, nullptr, nullptr,
                                          nullptr, s);
  const int T = 256;
  const size_t lds = (size_t)(t->L + T) * sizeof(double);
  BNPVI_REQUIRE(lds <= (size_t)kMaxLdsBytes, "gamma_update: L=%d too large for LDS", t->L);
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)gamma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(gamma_kernel, dim3(V), dim3(T), lds, s, mass, t->L, t->K, t->d_leaf_lo,
                     t->d_leaf_hi, gamma_hyper, gamma);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

// ------------------------------------------------------------------------------------------
// digamma for x >= 1 (arguments here are 1+mass, GAMMA+mass and their sum): upward recurrence
// to x >= 10, then the asymptotic series through x^-14 (truncation < 5e-17).
// ------------------------------------------------------------------------------------------
// 1/x for x >= 1 without the IEEE division sequence: hardware estimate + two Newton steps
// (relative error ~1e-16; the scaling / fix-up steps of a full division are not needed for
// normal, finite arguments).
__device__ __forceinline__ double recip_pos(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__device__ double digamma_pos(double x) {
#ifdef BNPVI_TABLATE_NODIGAMMA
  return x * 0.5;
#endif
  // x < 10: shift by 9 with ONE reciprocal, sum_{i<9} 1/(x+i) = P'(x)/P(x) for P = prod (x+i)
  // (a loop of up to nine reciprocals, run to the longest trip count in the wave, was a third of
  // the stick chain's time)
  double r = 0.0;
  if (x < 10.0) {
    double p = x, dp = 1.0;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
      const double f = x + (double)i;
      dp = fma(dp, f, p);
      p *= f;
    }
    r = -dp * recip_pos(p);
    x += 9.0;
  }
  const double rx = recip_pos(x);
  const double f = rx * rx;
  const double t =
      f * (-1.0 / 12.0 +
           f * (1.0 / 120.0 +
                f * (-1.0 / 252.0 +
                     f * (1.0 / 240.0 + f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 * rx + t;
}

// One workgroup per video.  e0[k] = psi(g0)-psi(g0+g1), e1[k] = psi(g1)-psi(g0+g1), root = 0.
//   on[l]     = sum of e0 over the path root->l (root excluded)          (src/var_inf.py:359-368)
//   before[l] = sum of e1 over all non-root nodes that precede l in pre-order and are not
//               ancestors of l                                           (src/var_inf.py:370-381)
//             = P[node(l)-1] - sum_{ancestors a} e1[a],  P = inclusive pre-order prefix of e1.
__global__ void prior_kernel(const double* __restrict__ gamma, const double* __restrict__ sigma,
                             const int32_t* __restrict__ parent,
                             const int32_t* __restrict__ node_of_leaf, int K, int L, int Lp,
                             double half_sz, float* __restrict__ prior, double* __restrict__ on_out,
                             double* __restrict__ before_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* e0 = reinterpret_cast<double*>(smem_raw);   // [K]
  double* P = e0 + K;                                 // [K]
  double* tmp = P + K;                                // [blockDim.x]
  const int v = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    if (k == 0) {
      e0[0] = 0.0;
      P[0] = 0.0;
    } else {
      const double2 g = reinterpret_cast<const double2*>(gamma)[(int64_t)v * K + k];
      const double ps = digamma_pos(g.x + g.y);
      e0[k] = digamma_pos(g.x) - ps;
      P[k] = digamma_pos(g.y) - ps;
    }
  }
  __syncthreads();
  block_inclusive_scan(P, K, tmp);

  double best = -INFINITY;
  for (int pass = 0; pass < 2; ++pass) {
    for (int l = threadIdx.x; l < Lp; l += blockDim.x) {
      if (l >= L) {
        if (pass == 1) prior[(int64_t)v * Lp + l] = -INFINITY;
        continue;
      }
      const int k = node_of_leaf[l];
      double on = 0.0, anc = 0.0;
      if (k > 0) {
        on = e0[k];
        int a = parent[k];
        while (a > 0) {
          on += e0[a];
          anc += P[a] - P[a - 1];
          a = parent[a];
        }
      }
      const double before = (k > 0 ? P[k - 1] : 0.0) - anc;
      const double val = on + before - half_sz / sigma[k];
      if (pass == 0) {
        best = fmax(best, val);
      } else {
        prior[(int64_t)v * Lp + l] = (float)((val - best) * 1.4426950408889634);   // log2 units
        if (on_out) on_out[(int64_t)v * L + l] = on;
        if (before_out) before_out[(int64_t)v * L + l] = before;
      }
    }
    if (pass == 0) {
      __syncthreads();
      tmp[threadIdx.x] = best;
      __syncthreads();
      for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) tmp[threadIdx.x] = fmax(tmp[threadIdx.x], tmp[threadIdx.x + off]);
        __syncthreads();
      }
      best = tmp[0];
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------
// Trees of up to kWaveStickMaxK nodes: ONE WAVE per video does the gamma-step and/or the prior
// table (no workgroup barriers; scans by lane segments + a shuffle scan).  LDS per wave:
// Pm[L] (leaf-mass prefix), e0[K], P[K].
// ------------------------------------------------------------------------------------------
constexpr int kStickWaves = 4;

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// in-place inclusive scan of a[0..n) in LDS by one wave
__device__ __forceinline__ void wave_inclusive_scan(double* a, int n, int lane) {
  const int seg = (n + 63) >> 6;
  const int b = lane * seg < n ? lane * seg : n;
  const int e = (b + seg < n) ? b + seg : n;
  double run = 0.0;
  for (int j = b; j < e; ++j) {
    run += a[j];
    a[j] = run;
  }
  double incl = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  const double add = incl - run;
  for (int j = b; j < e; ++j) a[j] += add;
  wave_sync_lds();
}

// One video by one wave.  leaf_sigma == nullptr: the leaf precisions come straight from
// the statistics (sigma_l = sB + sZ*S0_l, src/var_inf.py:268-272) -- used when the node-parameter
// update runs concurrently in the same launch.  ZERO_MASS: the mass row is zeroed once read.
// LDS per wave: Pm[L] (leaf-mass prefix, later the leaves' un-normalised log priors), e0[K], P[K].
// (A team of four waves per video was tried for K = 585 -- the chain is fp64 throughput, not latency:
// 158 us against 130 us for one wave per video, profiles/r02_notes.md.)
template <bool DO_GAMMA, bool DO_PRIOR, bool ZERO_MASS>
__device__ __forceinline__ void stick_wave_video(
    int v, int tid, double* Pm, double* e0, double* P, double* mass_rw,
    const double* __restrict__ mass, double* gamma, const double* __restrict__ leaf_sigma,
    const double* __restrict__ stats, int W, double sB, double sZ, const int32_t* __restrict__ parent,
    const int32_t* __restrict__ node_of_leaf, const int32_t* __restrict__ leaf_lo,
    const int32_t* __restrict__ leaf_hi, int K, int L, int Lp, double gamma_hyper, double half_sz,
    float* __restrict__ prior, double* __restrict__ on_out, double* __restrict__ before_out) {
  constexpr int T = 64;
  if (DO_GAMMA) {
    for (int l = tid; l < L; l += T) {
      Pm[l] = mass[(int64_t)v * L + l];
      if (ZERO_MASS) mass_rw[(int64_t)v * L + l] = 0.0;
    }
    wave_sync_lds();
    wave_inclusive_scan(Pm, L, tid);
  }
  for (int k = tid; k < K; k += T) {
    double2 g;
    if (DO_GAMMA) {
      const int lo = leaf_lo[k], hi = leaf_hi[k];
      const double before = lo > 0 ? Pm[lo - 1] : 0.0;
      const double upto = hi > 0 ? Pm[hi - 1] : 0.0;
      g.x = 1.0 + (upto - before);
      g.y = gamma_hyper + before;
      reinterpret_cast<double2*>(gamma)[(int64_t)v * K + k] = g;
    } else {
      g = reinterpret_cast<const double2*>(gamma)[(int64_t)v * K + k];
    }
    if (DO_PRIOR) {
      if (k == 0) {
        e0[0] = 0.0;
        P[0] = 0.0;
      } else {
        const double ps = digamma_pos(g.x + g.y);
        e0[k] = digamma_pos(g.x) - ps;
        P[k] = digamma_pos(g.y) - ps;
      }
    }
  }
  if (!DO_PRIOR) return;
  wave_sync_lds();                       // also: every read of the mass prefix Pm is done
  wave_inclusive_scan(P, K, tid);

#ifdef BNPVI_TABLATE_NOPRIORPASS
  for (int l = tid; l < Lp; l += T)
    prior[(int64_t)v * Lp + l] = l < L ? (float)P[node_of_leaf[l] > 0 ? 1 : 0] : -INFINITY;
  return;
#endif
  // pass 0: un-normalised log prior of every leaf (kept in Pm) and its maximum; pass 1: the table
  double best = -INFINITY;
  for (int l = tid; l < L; l += T) {
    const int k = node_of_leaf[l];
    double on = 0.0, anc = 0.0;
    if (k > 0) {
      on = e0[k];
      int a = parent[k];
      while (a > 0) {
        on += e0[a];
        anc += P[a] - P[a - 1];
        a = parent[a];
      }
    }
    const double before = (k > 0 ? P[k - 1] : 0.0) - anc;
    const double sig = leaf_sigma ? leaf_sigma[k] : sB + sZ * stats[(int64_t)l * W];
    const double val = on + before - half_sz / sig;
    best = fmax(best, val);
    Pm[l] = val;                         // read back by the same thread only
    if (on_out) on_out[(int64_t)v * L + l] = on;
    if (before_out) before_out[(int64_t)v * L + l] = before;
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) best = fmax(best, __shfl_xor(best, off));
  for (int l = tid; l < Lp; l += T)
    prior[(int64_t)v * Lp + l] = l < L ? (float)((Pm[l] - best) * 1.4426950408889634) : -INFINITY;   // log2 units
}

__host__ __device__ inline size_t stick_wave_lds(int K, int L) {
  return (size_t)(L + 2 * K) * sizeof(double);
}

template <bool DO_GAMMA, bool DO_PRIOR>
__global__ __launch_bounds__(kStickWaves * 64) void stick_wave_kernel(
    const double* __restrict__ mass, double* gamma, const double* __restrict__ sigma,
    const int32_t* __restrict__ parent, const int32_t* __restrict__ node_of_leaf,
    const int32_t* __restrict__ leaf_lo, const int32_t* __restrict__ leaf_hi, int V, int K, int L, int Lp,
    double gamma_hyper, double half_sz, float* __restrict__ prior, double* __restrict__ on_out,
    double* __restrict__ before_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double* Pm = reinterpret_cast<double*>(smem_raw) + (size_t)wave * (L + 2 * K);
  const int v = blockIdx.x * kStickWaves + wave;
  if (v >= V) return;
  stick_wave_video<DO_GAMMA, DO_PRIOR, false>(v, lane, Pm, Pm + L, Pm + L + K, nullptr, mass, gamma, sigma,
                                              nullptr, 0, 0.0, 0.0, parent, node_of_leaf, leaf_lo, leaf_hi, K,
                                              L, Lp, gamma_hyper, half_sz, prior, on_out, before_out);
}

// ------------------------------------------------------------------------------------------
// Small trees: the whole middle of an iteration in ONE launch.  Workgroup 0 updates the node
// parameters (all levels) and then packs the leaf operands of the coming E-step from the new
// alpha; every other workgroup runs the stick chain (gamma + prior table) for 16 videos, one
// per wave.  The two parts are independent (the prior's 1/sigma_l term is taken from the
// statistics), so no workgroup waits for another.
// ------------------------------------------------------------------------------------------
constexpr int kMidWaves = 16;

struct MidArgs {
  // tree
  const int32_t *level_nodes, *level_ptr, *parent, *leaf_of_node, *node_of_leaf, *child_ptr, *child_idx,
      *leaf_lo, *leaf_hi;
  int nlevels, K, L, Lp;
  int internal_levels;     // the node update's level loop starts at internal_levels - 1: deeper levels hold only leaves
  // node parameters
  const double* stats;
  int W, D;
  double sB, sZ;
  const double* alpha0;
  double* alpha;
  double* sigma;
  // stick chain
  double* mass;
  int V;
  double gamma_hyper;
  double* gamma;
  float* prior;
  // leaf operands
  int DP, LpPack;
  int stick_waves;         // waves per stick workgroup that take a video (LDS: L + 2K doubles each)
  int block0_lds;          // workgroup 0 keeps alpha and the tree in LDS (it fits)
  float* a32;
  float* aavec;
  f16x8* apack;
  float* scal;
};

// Workgroup 0 when K*D doubles + the tree's index arrays fit in LDS: ONE round of global loads
// (index arrays, old alpha, the leaves' statistics -- none depends on another), after which the
// level recursion and the operand packing run out of LDS.  Same dataflow as
// node_params_tree_block: a node reads its parent's OLD alpha and its children's NEW alpha;
// leaves (no children) therefore go first at any depth, then internal nodes deepest level first.
__host__ __device__ inline size_t midstep_block0_lds(int K, int L, int D, int nlevels) {
  return (size_t)K * D * sizeof(double) + (size_t)(5 * K + L + nlevels + 4) * sizeof(int32_t);
}

__device__ __forceinline__ void params_operands_lds_block(const MidArgs& a, char* smem) {
  const int K = a.K, D = a.D, L = a.L, W = a.W, T = blockDim.x, tid = threadIdx.x;
  double* al = reinterpret_cast<double*>(smem);                       // [K][D]
  int32_t* parent = reinterpret_cast<int32_t*>(al + (size_t)K * D);   // [K]
  int32_t* leaf_of_node = parent + K;                                  // [K]
  int32_t* child_ptr = leaf_of_node + K;                               // [K+1]
  int32_t* child_idx = child_ptr + K + 1;                              // [K]
  int32_t* level_nodes = child_idx + K;                                // [K]
  int32_t* level_ptr = level_nodes + K;                                // [nlevels+1]
  int32_t* node_of_leaf = level_ptr + a.nlevels + 1;                   // [L]
  __shared__ float red[2][kMidWaves];

  // ---- round 1: every global read of the node update
  for (int i = tid; i < K; i += T) {
    parent[i] = a.parent[i];
    leaf_of_node[i] = a.leaf_of_node[i];
    child_ptr[i] = a.child_ptr[i];
    child_idx[i] = i < K - 1 ? a.child_idx[i] : 0;      // K-1 edges
    level_nodes[i] = a.level_nodes[i];
  }
  if (tid == 0) child_ptr[K] = a.child_ptr[K];
  if (tid <= a.nlevels) level_ptr[tid] = a.level_ptr[tid];
  for (int i = tid; i < L; i += T) node_of_leaf[i] = a.node_of_leaf[i];
  for (int e = tid; e < K * D; e += T) al[e] = a.alpha[e];
  constexpr int kPre = 4;                                 // leaf statistics held in registers
  double s0[kPre], s1[kPre];
#pragma unroll
  for (int i = 0; i < kPre; ++i) {
    const int e = tid + i * T;
    s0[i] = s1[i] = 0.0;
    if (e < L * D) {
      s0[i] = a.stats[(int64_t)(e / D) * W];
      s1[i] = a.stats[(int64_t)(e / D) * W + 1 + e % D];
    }
  }
  __syncthreads();

  // ---- leaves
  auto leaf_update = [&](int e, double S0, double S1) {
    const int l = e / D, d = e % D;
    const int k = node_of_leaf[l];
    const int p = parent[k];
    const double base = (p < 0) ? (a.alpha0 ? a.alpha0[d] : 0.0) : al[p * D + d];
    const double s = a.sB + S0 * a.sZ;
    const double v = (base * a.sB + S1 * a.sZ) / s;
    a.alpha[(int64_t)k * D + d] = v;
    if (d == 0) a.sigma[k] = s;
    return v;
  };
  // new leaf values are written to LDS after a barrier: a leaf's parent may (in a one-node tree
  // only) be absent, but another leaf never reads a leaf -- the barrier is for the fixed-count part
  double nv[kPre];
#pragma unroll
  for (int i = 0; i < kPre; ++i) {
    const int e = tid + i * T;
    nv[i] = e < L * D ? leaf_update(e, s0[i], s1[i]) : 0.0;
  }
  for (int e = tid + kPre * T; e < L * D; e += T) {
    const int l = e / D;
    const double v = leaf_update(e, a.stats[(int64_t)l * W], a.stats[(int64_t)l * W + 1 + e % D]);
    al[node_of_leaf[l] * D + e % D] = v;     // no reader of a leaf's old value exists
  }
#pragma unroll
  for (int i = 0; i < kPre; ++i) {
    const int e = tid + i * T;
    if (e < L * D) al[node_of_leaf[e / D] * D + e % D] = nv[i];
  }
  __syncthreads();

  // ---- internal nodes, deepest level first
#ifdef BNPVI_TABLATE_B0_NOLEVELS
  if (K < 0)
#endif
  for (int lev = a.internal_levels - 1; lev >= 0; --lev) {
    const int off = level_ptr[lev];
    const int count = level_ptr[lev + 1] - off;
    bool any = false;
    for (int e = tid; e < count * D; e += T) {
      const int k = level_nodes[off + e / D];
      if (leaf_of_node[k] >= 0) continue;
      any = true;
      const int d = e % D;
      const int p = parent[k];
      const double base = (p < 0) ? (a.alpha0 ? a.alpha0[d] : 0.0) : al[p * D + d];
      const int c0 = child_ptr[k], c1 = child_ptr[k + 1];
      const double s = (1.0 + (double)(c1 - c0)) * a.sB;
      double acc = base;
      for (int c = c0; c < c1; ++c) acc += al[child_idx[c] * D + d];
      const double v = acc * a.sB / s;
      al[k * D + d] = v;
      a.alpha[(int64_t)k * D + d] = v;
      if (d == 0) a.sigma[k] = s;
    }
    (void)any;
    __syncthreads();
  }

  // ---- leaf operands of the E-step from the LDS copy (layout: leaf_operands_block)
  const int DP = a.DP, Lp = a.LpPack;
  const int lane = tid & 63, wave = tid >> 6, nw = T >> 6;
#ifdef BNPVI_TABLATE_B0_NOPACKALL
  if (K > 0) return;
#endif
  // fp32 rows for the E-step's exact-recompute path: coalesced
  for (int e = tid; e < Lp * DP; e += T) {
    const int leaf = e / DP, d = e % DP;
    a.a32[e] = (leaf < L && d < D) ? (float)al[node_of_leaf[leaf] * D + d] : 0.f;
  }
  // ||alpha||^2 and max|alpha|: FOUR threads per leaf, each walking a quarter of the row in LDS (one thread per
  // leaf took 1.9 us of this workgroup's ~16 us chain at config 2; a wave per leaf with a 6-step fp64 shuffle
  // reduction per row was slower still); the walk starts at a leaf-dependent dimension so that the lanes of a
  // wave hit different banks
  float amax = 0.f, aamax = 0.f;
  {
    const int Q = (D + 3) >> 2;
    for (int u = tid; u < 4 * Lp; u += T) {              // T is a multiple of 4: a leaf's four threads share a wave
      const int leaf = u >> 2, part = u & 3;
      double aa = 0.0;
#ifdef BNPVI_TABLATE_B0_NOAA
      if (leaf < 0) {
#else
      if (leaf < L) {
#endif
        const double* row = al + node_of_leaf[leaf] * D;
        int q = leaf % Q;
        for (int it = 0; it < Q; ++it) {
          const int d = part * Q + q;
          if (d < D) {
            const float f = (float)row[d];
            aa = fma((double)f, (double)f, aa);
            amax = fmaxf(amax, fabsf(f));
          }
          q = q + 1 == Q ? 0 : q + 1;
        }
      }
      aa += __shfl_xor(aa, 1);
      aa += __shfl_xor(aa, 2);
      if (part == 0) a.aavec[leaf] = (float)aa;
      aamax = fmaxf(aamax, (float)aa);
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) aamax = fmaxf(aamax, __shfl_xor(aamax, off));
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  if (lane == 0) {
    red[0][wave] = amax;
    red[1][wave] = aamax;
  }
  __syncthreads();
  amax = 0.f;
  aamax = 0.f;
  for (int w = 0; w < nw; ++w) {
    amax = fmaxf(amax, red[0][w]);
    aamax = fmaxf(aamax, red[1][w]);
  }
  const int ka = scale_exponent_bits(__float_as_int(amax));
  if (tid == 0) {
    a.scal[0] = pow2f(-ka);
    a.scal[1] = amax;
    a.scal[2] = aamax;
  }
  const int S = DP / 32;
  const float sa = pow2f(ka);
  for (int u = tid; u < (Lp / kLT) * S * 2 * 64; u += T) {
    const int ln = u & 63;
    const int part = (u >> 6) & 1;
    const int ts = u >> 7;
    const int s = ts % S, Tl = ts / S;
    const int leaf = kLT * Tl + (ln & 15);
    const int kg = ln >> 4;
    const double* row = leaf < L ? al + node_of_leaf[leaf] * D : nullptr;
    f16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int dim = 32 * s + 16 * (e >> 2) + 4 * kg + (e & 3);
      const float v = ((row && dim < D) ? (float)row[dim] : 0.f) * sa;
      const _Float16 h = (_Float16)v;
      out[e] = part == 0 ? h : (_Float16)(v - (float)h);
    }
    a.apack[u] = out;
  }
}

constexpr size_t kMidBlock0LdsMax = 96 * 1024;

__global__ __launch_bounds__(kMidWaves * 64) void midstep_small_kernel(MidArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (blockIdx.x == 0) {
#ifdef BNPVI_TABLATE_NOBLOCK0
    return;
#endif
    if (a.block0_lds) {
      params_operands_lds_block(a, smem_raw);
      return;
    }
    node_params_tree_block(a.level_nodes, a.level_ptr, a.nlevels, a.parent, a.leaf_of_node, a.child_ptr,
                           a.child_idx, a.stats, a.W, a.D, a.sB, a.sZ, a.alpha0, a.alpha, a.sigma);
    leaf_operands_block(a.alpha, a.D, a.DP, a.node_of_leaf, a.L, a.LpPack, a.a32, a.aavec, a.apack, a.scal);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (wave >= a.stick_waves) return;
  double* Pm = reinterpret_cast<double*>(smem_raw) + (size_t)wave * (a.L + 2 * a.K);
  const int v = (blockIdx.x - 1) * a.stick_waves + wave;
  if (v >= a.V) return;
  stick_wave_video<true, true, true>(v, lane, Pm, Pm + a.L, Pm + a.L + a.K, a.mass, a.mass, a.gamma, nullptr,
                                     a.stats, a.W, a.sB, a.sZ, a.parent, a.node_of_leaf, a.leaf_lo, a.leaf_hi,
                                     a.K, a.L, a.Lp, a.gamma_hyper, 0.5 * a.sZ, a.prior, nullptr, nullptr);
}

int launch_params_prior(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* stats, int W, int D, double sB,
                        double sZ, const double* alpha0, double* alpha, double* sigma, double* mass, int V,
                        double gamma_hyper, double* gamma, float* prior, hipStream_t s) {
  const int DP = bnpvi_padded_dim(D);
  // stick workgroups: as many of the 16 waves take a video as fit 64 KB of LDS (L + 2K doubles each)
  const size_t per_wave = stick_wave_lds(t->K, t->L);
  int stick_waves = (int)((size_t)64 * 1024 / per_wave);
  stick_waves = stick_waves > kMidWaves ? kMidWaves : stick_waves;
  size_t lds = (size_t)stick_waves * per_wave;
  const bool small = (int64_t)t->K * D <= 64 * 1024 && stick_waves >= 1 && t->K <= kWaveStickMaxK &&
                     (int64_t)round_up(t->Lp, kLT * kStreamNT) * DP <= 64 * 1024;
  ctx->ops_tree = nullptr;
  if (!small) {
    int rc = launch_node_params(ctx, t, stats, W, D, sB, sZ, alpha0, alpha, sigma, s);
    if (rc) return rc;
    return launch_gamma_prior(ctx, t, mass, sigma, V, gamma_hyper, sZ, gamma, prior, s);
  }
  LeafWorkspace w;
  int rc = leaf_workspace(ctx, t, DP, &w);
  if (rc) return rc;
  MidArgs a;
  a.level_nodes = t->d_level_nodes; a.level_ptr = t->d_level_ptr; a.parent = t->d_parent;
  a.leaf_of_node = t->d_leaf_of_node; a.node_of_leaf = t->d_node_of_leaf; a.child_ptr = t->d_child_ptr;
  a.child_idx = t->d_child_idx; a.leaf_lo = t->d_leaf_lo; a.leaf_hi = t->d_leaf_hi;
  a.nlevels = t->nlevels; a.K = t->K; a.L = t->L; a.Lp = t->Lp; a.internal_levels = t->internal_levels;
  a.stats = stats; a.W = W; a.D = D; a.sB = sB; a.sZ = sZ; a.alpha0 = alpha0; a.alpha = alpha; a.sigma = sigma;
  a.mass = mass; a.V = V; a.gamma_hyper = gamma_hyper; a.gamma = gamma; a.prior = prior;
  a.stick_waves = stick_waves;
  a.DP = DP; a.LpPack = w.LpPack; a.a32 = w.a32; a.aavec = w.aavec; a.apack = w.apack; a.scal = w.scal;
  const size_t lds0 = midstep_block0_lds(t->K, t->L, D, t->nlevels);
  a.block0_lds = lds0 <= kMidBlock0LdsMax && t->nlevels < kMidWaves * 64;
  if (a.block0_lds && lds0 > lds) lds = lds0;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)midstep_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(midstep_small_kernel, dim3(1 + (unsigned)ceil_div(V, stick_waves)), dim3(kMidWaves * 64), lds,
                     s, a);
  BNPVI_LAUNCH_CHECK();
  ctx->ops_tree = t;
  ctx->ops_alpha = alpha;
  ctx->ops_D = D;
  ctx->mass_clean = mass;
  ctx->mass_clean_elems = (size_t)V * t->L;
  ctx->mass_clean_stream = s;
  return BNPVI_OK;
}

template <bool DO_GAMMA, bool DO_PRIOR>
static int launch_stick_wave(const bnpvi_tree* t, const double* mass, double* gamma, const double* sigma,
                             int V, double gamma_hyper, double sZ, float* prior, double* on, double* before,
                             hipStream_t s) {
  const size_t lds = (size_t)kStickWaves * stick_wave_lds(t->K, t->L);
  auto k = stick_wave_kernel<DO_GAMMA, DO_PRIOR>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3((unsigned)ceil_div(V, kStickWaves)), dim3(kStickWaves * 64), lds, s, mass, gamma,
                     sigma, t->d_parent, t->d_node_of_leaf, t->d_leaf_lo, t->d_leaf_hi, V, t->K, t->L, t->Lp,
                     gamma_hyper, 0.5 * sZ, prior, on, before);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

int launch_gamma_prior(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* mass, const double* sigma, int V,
                       double gamma_hyper, double sZ, double* gamma, float* prior, hipStream_t s) {
  if (t->K <= kWaveStickMaxK)
    return launch_stick_wave<true, true>(t, mass, gamma, sigma, V, gamma_hyper, sZ, prior, nullptr, nullptr, s);
  int rc = launch_gamma_update(ctx, t, mass, V, gamma_hyper, gamma, s);
  if (rc) return rc;
  return launch_prior_table(ctx, t, gamma, sigma, V, sZ, prior, nullptr, nullptr, s);
}

int launch_prior_table(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* gamma,
                       const double* sigma, int V, double sZ, float* prior, double* on,
                       double* before, hipStream_t s) {
  (void)ctx;
  if (t->K <= kWaveStickMaxK)
    return launch_stick_wave<false, true>(t, nullptr, const_cast<double*>(gamma), sigma, V, 0.0, sZ, prior,
                                          on, before, s);
  const int T = 256;
  const size_t lds = (size_t)(2 * t->K + T) * sizeof(double);
  BNPVI_REQUIRE(lds <= (size_t)kMaxLdsBytes, "prior_table: K=%d too large for LDS (max ~10000)", t->K);
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)prior_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(prior_kernel, dim3(V), dim3(T), lds, s, gamma, sigma, t->d_parent,
                     t->d_node_of_leaf, t->K, t->L, t->Lp, 0.5 * sZ, prior, on, before);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

// ------------------------------------------------------------------------------------------
// Weighted farthest-point seeding (src/var_inf.py:23-36).
// ------------------------------------------------------------------------------------------
struct Best {
  float score;
  long long row;
};

__device__ inline bool better(float s, long long r, float bs, long long br) {
  return (s > bs) || (s == bs && r < br);   // np.argmax: first index wins ties
}

// ------------------------------------------------------------------------------------------
// Batched seeding: the leaves that split in one split_nodes walk are seeded TOGETHER, one pass over
// Z per centre index for all of them (J phi columns at a time) instead of one pass per centre per
// leaf -- Z (the big operand) is read num_centres - 1 times in all, not J * (num_centres - 1) times.
// ------------------------------------------------------------------------------------------
struct Best64 {
  double score;
  long long row;
};

__device__ inline bool better64(double s, long long r, double bs, long long br) {
  return (s > bs) || (s == bs && r < br);
}

// fp32 phi at or above this makes a frame a candidate for the fp64 ordering below.  At most one leaf of a frame
// can hold more than half of its mass, so every row is summed at most once per pass.
constexpr float kDominant = 0.5f;

// First centres (src/var_inf.py:29-30): arg-max of each of the J phi columns, lowest row on ties.  The
// reference takes it over fp64 responsibilities; the device matrix is fp32, where frames whose fp64 phi differ
// by less than 6e-8 read the same -- all the frames at exactly 1.0 on well-separated data, but also two frames
// at 0.99976906 and 0.99976901 (found by the regime-walk trajectory test).  For a frame dominated by the leaf,
// phi = 1 / (1 + c) with c = the frame's mass on all OTHER leaves, which fp32 holds to full RELATIVE precision:
// the error of that phi is ~1e-7 * c instead of ~6e-8.  So next to the fp32 arg-max every column tracks the
// arg-max of the fp64 1 / (1 + c) over its dominated frames (fp32 phi >= 0.5; c summed in fp64 over the row
// without the column itself: no cancellation), and the final kernel takes that winner whenever there is one --
// the true maximum is then among those frames.  Below 0.5 the fp32 arg-max stands (its ties are near-ties of
// the reference's own arithmetic).  Assumes rows that sum to 1 (phi straight from an E-step): callers pass
// L = 0 otherwise.  One wave per row, lane j < J owns column cols[j].  L == 0: the columns are plain weights,
// not rows of responsibilities -- fp32 arg-max only.
__global__ __launch_bounds__(256) void kmeans_first_multi_kernel(const float* __restrict__ phi, int ldphi,
                                                                 int64_t N, int L,
                                                                 const int32_t* __restrict__ cols, int J,
                                                                 Best* __restrict__ part32,
                                                                 Best64* __restrict__ part64) {
  __shared__ Best sh32[4][64];
  __shared__ Best64 sh64[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mycol = lane < J ? cols[lane] : 0;
  Best b32;
  b32.score = -1.f;
  b32.row = 0x7fffffffffffffffLL;
  Best64 b64;
  b64.score = -1.0;
  b64.row = 0x7fffffffffffffffLL;
  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
    const float* row = phi + n * ldphi;
    const float wn = lane < J ? row[mycol] : -1.f;
    if (lane < J && better(wn, n, b32.score, b32.row)) {
      b32.score = wn;
      b32.row = n;
    }
    unsigned long long mask = __ballot(L > 0 && lane < J && wn >= kDominant);
    while (mask != 0ull) {                                     // wave-uniform
      const int jl = __ffsll((long long)mask) - 1;
      mask &= mask - 1ull;
      const int col = __builtin_amdgcn_readlane(mycol, jl);
      double c = 0.0;
      for (int l = lane; l < L; l += 64)
        if (l != col) c += (double)row[l];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
      const double s = 1.0 / (1.0 + c);
      if (lane == jl && better64(s, n, b64.score, b64.row)) {
        b64.score = s;
        b64.row = n;
      }
    }
  }
  sh32[wave][lane] = b32;
  sh64[wave][lane] = b64;
  __syncthreads();
  if (wave == 0 && lane < J) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const Best o = sh32[w][lane];
      if (better(o.score, o.row, b32.score, b32.row)) b32 = o;
      const Best64 o64 = sh64[w][lane];
      if (better64(o64.score, o64.row, b64.score, b64.row)) b64 = o64;
    }
    part32[(int64_t)blockIdx.x * J + lane] = b32;
    part64[(int64_t)blockIdx.x * J + lane] = b64;
  }
}

// one workgroup per column j: reduce the partials; the fp64 winner over the dominated frames stands when there is one
__global__ __launch_bounds__(256) void kmeans_final_multi_kernel(const Best* __restrict__ part32,
                                                                 const Best64* __restrict__ part64, int nparts,
                                                                 int J, int64_t* __restrict__ best_row,
                                                                 double* __restrict__ best_score) {
  __shared__ Best sh32[256];
  __shared__ Best64 sh64[256];
  const int j = blockIdx.x;
  Best b32;
  b32.score = -1.f;
  b32.row = 0x7fffffffffffffffLL;
  Best64 b64;
  b64.score = -1.0;
  b64.row = 0x7fffffffffffffffLL;
  for (int b = threadIdx.x; b < nparts; b += blockDim.x) {
    const Best o = part32[(int64_t)b * J + j];
    if (better(o.score, o.row, b32.score, b32.row)) b32 = o;
    if (part64 != nullptr) {
      const Best64 o64 = part64[(int64_t)b * J + j];
      if (better64(o64.score, o64.row, b64.score, b64.row)) b64 = o64;
    }
  }
  sh32[threadIdx.x] = b32;
  sh64[threadIdx.x] = b64;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const Best o = sh32[threadIdx.x + off];
      if (better(o.score, o.row, sh32[threadIdx.x].score, sh32[threadIdx.x].row)) sh32[threadIdx.x] = o;
      const Best64 o64 = sh64[threadIdx.x + off];
      if (better64(o64.score, o64.row, sh64[threadIdx.x].score, sh64[threadIdx.x].row)) sh64[threadIdx.x] = o64;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const Best w32 = sh32[0];
    const Best64 w64 = sh64[0];
    if (part64 != nullptr && w64.score >= 0.0) {
      best_row[j] = w64.row;
      best_score[j] = w64.score;
    } else {
      // all scores 0 (e.g. all weights 0): normalize() falls back to uniform => index 0
      best_row[j] = (w32.score > 0.f) ? w32.row : 0;
      best_score[j] = (double)w32.score;
    }
  }
}

// ------------------------------------------------------------------------------------------
// First centres with the reference's OWN rounding of the responsibilities.  The reference normalises a row as
// phi = exp(logit - max) / phi_sum with phi_sum = get_phi_sum(root) (src/var_inf.py:336-343, 391-392): an fp64 sum
// NESTED like the tree (every internal node adds its children in order, starting from 0).  For a frame its leaf
// dominates, exp(0) = 1 sits somewhere inside that nest and the other leaves' tiny terms are absorbed or not
// depending on WHERE they meet it: a frame with five terms of 5e-17 each reads phi = 1.0 exactly when they are
// added to the 1 one by one, 1 - 2.2e-16 when they meet each other first.  np.argmax then takes the lowest row
// among equal doubles, so on well-separated data (many rows at or near 1.0) the choice of the first centre hangs on
// that rounding -- a flat 1 / (1 + c) picked another frame in 2 of 29 splits of a random trajectory
// (tests/test_gpu_round3.py::test_random_trajectories_against_oracle).  Here the nest is replayed: per dominated
// row one wave writes x_l = phi_l / phi_col (fp64; the column itself exactly 1) into an LDS image of the tree's
// nodes and sums it level by level, children in order, one lane per internal node; score = 1 / phi_sum as the
// reference divides.  The fp32 phi_l carry the kernels' relative error (~3e-5), which moves a sum across a rounding
// boundary only when it sits within that distance of one.  The later factors of weighted_kmeans (weights / sum,
// * 1/N, normalize()) are monotone and cannot be replayed bit for bit (np.sum over all frames); they can only merge
// neighbouring doubles, which is left as a near-tie.
// ------------------------------------------------------------------------------------------
struct NestTree {
  const int32_t *node_of_leaf, *leaf_of_node, *child_ptr, *child_idx, *level_ptr, *level_nodes;
  int K, L, internal_levels;
};

// LDS_TREE: the tree's index arrays are copied into LDS once per workgroup (L + 4K + levels ints behind the four
// node images).  Read from global memory they put three or four dependent L2 round trips into every level of every
// dominated row: 2.15 ms per 2M frames at K = 80, against 0.72 ms for the flat rule this kernel replaced.
__host__ __device__ inline size_t nest_tree_ints(int K, int L, int internal_levels) {
  return (size_t)L + 4 * (size_t)K + 1 + (size_t)internal_levels + 1;
}

template <bool LDS_TREE>
__global__ __launch_bounds__(256) void kmeans_first_nested_kernel(const float* __restrict__ phi, int ldphi, int64_t N,
                                                                  NestTree tg, const int32_t* __restrict__ cols, int J,
                                                                  Best* __restrict__ part32,
                                                                  Best64* __restrict__ part64) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ Best sh32[4][64];
  __shared__ Best64 sh64[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* val = reinterpret_cast<double*>(smem_raw) + (size_t)wave * tg.K;    // this wave's image of the nodes
  NestTree t = tg;
  if (LDS_TREE) {
    int32_t* base = reinterpret_cast<int32_t*>(reinterpret_cast<double*>(smem_raw) + (size_t)4 * tg.K);
    int32_t* nol = base;
    int32_t* lofn = nol + tg.L;
    int32_t* cptr = lofn + tg.K;
    int32_t* cidx = cptr + tg.K + 1;
    int32_t* lnodes = cidx + tg.K;
    int32_t* lptr = lnodes + tg.K;
    for (int i = threadIdx.x; i < tg.L; i += 256) nol[i] = tg.node_of_leaf[i];
    for (int i = threadIdx.x; i < tg.K; i += 256) {
      lofn[i] = tg.leaf_of_node[i];
      cptr[i] = tg.child_ptr[i];
      cidx[i] = i < tg.K - 1 ? tg.child_idx[i] : 0;                           // K - 1 edges
      lnodes[i] = tg.level_nodes[i];
    }
    if (threadIdx.x == 0) cptr[tg.K] = tg.child_ptr[tg.K];
    for (int i = threadIdx.x; i <= tg.internal_levels; i += 256) lptr[i] = tg.level_ptr[i];
    __syncthreads();
    t.node_of_leaf = nol; t.leaf_of_node = lofn; t.child_ptr = cptr; t.child_idx = cidx;
    t.level_nodes = lnodes; t.level_ptr = lptr;
  }
  const int mycol = lane < J ? cols[lane] : 0;
  Best b32;
  b32.score = -1.f;
  b32.row = 0x7fffffffffffffffLL;
  Best64 b64;
  b64.score = -1.0;
  b64.row = 0x7fffffffffffffffLL;
  for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
    const float* row = phi + n * ldphi;
    const float wn = lane < J ? row[mycol] : -1.f;
    if (lane < J && better(wn, n, b32.score, b32.row)) {
      b32.score = wn;
      b32.row = n;
    }
    unsigned long long mask = __ballot(lane < J && wn >= kDominant);
    while (mask != 0ull) {                                     // wave-uniform; at most one leaf dominates a row
      const int jl = __ffsll((long long)mask) - 1;
      mask &= mask - 1ull;
      const int col = __builtin_amdgcn_readlane(mycol, jl);
      const double pc = (double)__uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(wn), jl));
      // Screen: a wave meets its rows in ascending order, so a row only matters if it BEATS the column's best so far.
      // 1 / phi_sum <= 1, and it equals 1 / (1 + c / pc), c = the other leaves' mass, up to the nest's rounding
      // (a few 2^-53) and the fp32 sum's (< 64 * 2^-24 of c): rows that cannot reach the best skip the nest.
      {
        float part = 0.f;
        for (int l = lane; l < t.L; l += 64) part += l == col ? 0.f : row[l];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) part += __shfl_xor(part, off);
        const double rest = (double)part / pc;
        const double bestj = __shfl(b64.score, jl);
        if (bestj >= 1.0 || 1.0 / (1.0 + rest) + 1e-5 * rest + 1e-12 < bestj) continue;
      }
      for (int l = lane; l < t.L; l += 64) val[t.node_of_leaf[l]] = l == col ? 1.0 : (double)row[l] / pc;
      wave_sync_lds();
      for (int lev = t.internal_levels - 1; lev >= 0; --lev) {
        for (int i = t.level_ptr[lev] + lane; i < t.level_ptr[lev + 1]; i += 64) {
          const int k = t.level_nodes[i];
          if (t.leaf_of_node[k] < 0) {
            double acc = 0.0;
            for (int c = t.child_ptr[k]; c < t.child_ptr[k + 1]; ++c) acc += val[t.child_idx[c]];
            val[k] = acc;
          }
        }
        wave_sync_lds();
      }
      const double s = 1.0 / val[0];
      wave_sync_lds();                                         // everybody has read val[0] before the next row's image
      if (lane == jl && better64(s, n, b64.score, b64.row)) {
        b64.score = s;
        b64.row = n;
      }
    }
  }
  sh32[wave][lane] = b32;
  sh64[wave][lane] = b64;
  __syncthreads();
  if (wave == 0 && lane < J) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const Best o = sh32[w][lane];
      if (better(o.score, o.row, b32.score, b32.row)) b32 = o;
      const Best64 o64 = sh64[w][lane];
      if (better64(o64.score, o64.row, b64.score, b64.row)) b64 = o64;
    }
    part32[(int64_t)blockIdx.x * J + lane] = b32;
    part64[(int64_t)blockIdx.x * J + lane] = b64;
  }
}

constexpr size_t kNestLdsMax = 96 * 1024;      // 4 waves x K doubles: trees of up to 3072 nodes

int launch_kmeans_first_nested(bnpvi_ctx* ctx, const bnpvi_tree* t, const float* phi, int ldphi, int64_t N,
                               const int32_t* cols, int J, int64_t* best_row, double* best_score, hipStream_t s) {
  size_t lds = (size_t)4 * t->K * sizeof(double);
  if (lds > kNestLdsMax) return BNPVI_EUNSUPPORTED;
  const size_t with_tree = lds + nest_tree_ints(t->K, t->L, t->internal_levels) * sizeof(int32_t);
  const bool lds_tree = with_tree <= (size_t)kMaxLdsBytes - 16384;           // room for the kernel's static tables (8 KB)
  if (lds_tree) lds = with_tree;
  int64_t blocks = ceil_div(N, (int64_t)4);
  const int64_t cap = (int64_t)ctx->num_cu * 8;
  if (blocks > cap) blocks = cap;
  int rc = ctx->red.ensure((size_t)blocks * J * (sizeof(Best) + sizeof(Best64)));
  if (rc) return rc;
  Best64* p64 = (Best64*)ctx->red.ptr;          // 16-byte records first
  Best* p32 = (Best*)(p64 + blocks * J);
  NestTree nt{t->d_node_of_leaf, t->d_leaf_of_node, t->d_child_ptr, t->d_child_idx, t->d_level_ptr, t->d_level_nodes,
              t->K, t->L, t->internal_levels};
  auto kern = lds_tree ? kmeans_first_nested_kernel<true> : kmeans_first_nested_kernel<false>;
  if (lds > 48 * 1024)
    BNPVI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, phi, ldphi, N, nt, cols, J, p32, p64);
  BNPVI_LAUNCH_CHECK();
  hipLaunchKernelGGL(kmeans_final_multi_kernel, dim3(J), dim3(256), 0, s, p32, p64, (int)blocks, J, best_row,
                     best_score);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

// sum over the G lanes that share a row (G = 8 .. 64, aligned lane groups): DPP inside a 16-lane row,
// permlane swaps across rows.  Every lane of the group returns the same bits.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  if (G >= 16) v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x140, 0xf, 0xf, false));  // row_mirror
  if (G >= 32) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if (G >= 64) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}

// Later centres (src/var_inf.py:31-35) for J columns at once.  G = Dp / 4 lanes share a row, one float4 of
// it each; lane j < J of the group owns column cols[j]: its running min squared distance dist2[n][j],
// its score w * sqrt(dist2) and its arg-max.  centres: [J][Dp] fp32.
template <int G>
__global__ __launch_bounds__(256) void kmeans_pass_multi_kernel(const float* __restrict__ Z, int ldz, int64_t N,
                                                                const float* __restrict__ phi, int ldphi,
                                                                const int32_t* __restrict__ cols, int J,
                                                                const float* __restrict__ centres, int first,
                                                                float* __restrict__ dist2,
                                                                Best* __restrict__ partial) {
  __shared__ Best sh[256];
  constexpr int Dp = 4 * G;
  const int g = threadIdx.x % G;
  constexpr int64_t rows_per_block = 256 / G;
  const int mycol = g < J ? cols[g] : 0;
  Best mine;
  mine.score = -1.f;
  mine.row = 0x7fffffffffffffffLL;
  for (int64_t n = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / G; n < N;
       n += (int64_t)gridDim.x * rows_per_block) {
    const float4 z = *reinterpret_cast<const float4*>(Z + n * ldz + 4 * g);
    float wn = 0.f, dold = 0.f;
    if (g < J) {
      wn = phi[n * ldphi + mycol];
      if (!first) dold = dist2[n * J + g];
    }
    float myd = 0.f;
    for (int j = 0; j < J; ++j) {
      const float4 m = *reinterpret_cast<const float4*>(centres + j * Dp + 4 * g);
      const float dx = z.x - m.x, dy = z.y - m.y, dz = z.z - m.z, dw = z.w - m.w;
      const float d2 = group_sum<G>(dx * dx + dy * dy + dz * dz + dw * dw);
      if (g == j) myd = d2;
    }
    if (g < J) {
      const float dm = first ? myd : fminf(dold, myd);
      dist2[n * J + g] = dm;
      const float score = wn * __builtin_amdgcn_sqrtf(dm);
      if (better(score, n, mine.score, mine.row)) {
        mine.score = score;
        mine.row = n;
      }
    }
  }
  sh[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 128; off >= G; off >>= 1) {                   // threads off apart own the same column
    if ((int)threadIdx.x < off) {
      const Best o = sh[threadIdx.x + off];
      if (better(o.score, o.row, sh[threadIdx.x].score, sh[threadIdx.x].row)) sh[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < J) partial[(int64_t)blockIdx.x * J + threadIdx.x] = sh[threadIdx.x];
}

// The same pass with ONE LANE PER ROW (Dp <= 128).  The kernel above spends more on adding a row's partial sums
// across its G lanes (a five-step butterfly per row and centre) than on the differences themselves: 0.87 ms per
// 2M x 128 frames and 24 columns.  Here a lane keeps its whole row in registers, the centres are wave-uniform
// (scalar loads), the squared differences run down the row in eight independent fp32 chains added pairwise at the
// end, and nothing crosses lanes until the wave's arg-max per column (one butterfly per column and 64 rows).
// A wave's running best per column lives in LDS (no register array indexed by a runtime j).
template <int DP>
__global__ __launch_bounds__(256) void kmeans_pass_rows_kernel(const float* __restrict__ Z, int ldz, int64_t N,
                                                               const float* __restrict__ phi, int ldphi,
                                                               const int32_t* __restrict__ cols, int J,
                                                               const float* __restrict__ centres, int first,
                                                               float* __restrict__ dist2,
                                                               Best* __restrict__ partial) {
  __shared__ Best sh[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (lane < J) {
    sh[wave][lane].score = -1.f;
    sh[wave][lane].row = 0x7fffffffffffffffLL;
  }
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64; base < N; base += (int64_t)gridDim.x * 256) {
    const int64_t n = base + lane;
    const bool ok = n < N;
    const int64_t nr = ok ? n : N - 1;
    float z[DP];
    {
      const float4* zr = reinterpret_cast<const float4*>(Z + nr * ldz);
#pragma unroll
      for (int q = 0; q < DP / 4; ++q) {
        const float4 v = zr[q];
        z[4 * q] = v.x;
        z[4 * q + 1] = v.y;
        z[4 * q + 2] = v.z;
        z[4 * q + 3] = v.w;
      }
    }
    const float* prow = phi + nr * ldphi;
    float* drow = dist2 + nr * J;
    for (int j = 0; j < J; ++j) {                                 // wave-uniform
      const float* c = centres + (int64_t)j * DP;
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const float df = z[d] - c[d];
        acc[d & 7] = fmaf(df, df, acc[d & 7]);
      }
      const float myd = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
      const float wn = prow[cols[j]];
      const float dm = first ? myd : fminf(drow[j], myd);
      if (ok) drow[j] = dm;
      float score = ok ? wn * __builtin_amdgcn_sqrtf(dm) : -1.f;
      long long row = ok ? n : 0x7fffffffffffffffLL;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float os = __shfl_xor(score, off);
        const long long orow = __shfl_xor(row, off);
        if (better(os, orow, score, row)) {
          score = os;
          row = orow;
        }
      }
      if (lane == 0 && better(score, row, sh[wave][j].score, sh[wave][j].row)) {
        sh[wave][j].score = score;
        sh[wave][j].row = row;
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < J) {
    Best b = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const Best o = sh[w][threadIdx.x];
      if (better(o.score, o.row, b.score, b.row)) b = o;
    }
    partial[(int64_t)blockIdx.x * J + threadIdx.x] = b;
  }
}

int launch_kmeans_pass_multi(bnpvi_ctx* ctx, const float* Z, int ldz, int64_t N, int D, const float* phi,
                             int ldphi, int L, const int32_t* cols, int J, const float* centres, int first,
                             float* dist2, int64_t* best_row, double* best_score, hipStream_t s) {
  const int Dp = bnpvi_padded_dim(D);
  const int G = Dp / 4;                         // 8 .. 64
  int64_t blocks;
  if (centres == nullptr) {
    blocks = ceil_div(N, (int64_t)4);
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    int rc = ctx->red.ensure((size_t)blocks * J * (sizeof(Best) + sizeof(Best64)));
    if (rc) return rc;
    Best64* p64 = (Best64*)ctx->red.ptr;        // 16-byte records first
    Best* p32 = (Best*)(p64 + blocks * J);
    hipLaunchKernelGGL(kmeans_first_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, phi, ldphi, N, L, cols,
                       J, p32, p64);
    BNPVI_LAUNCH_CHECK();
    hipLaunchKernelGGL(kmeans_final_multi_kernel, dim3(J), dim3(256), 0, s, p32, p64, (int)blocks, J, best_row,
                       best_score);
    BNPVI_LAUNCH_CHECK();
    return BNPVI_OK;
  }
  const char* kenv = dev_getenv("BNPVI_KMEANS_PASS");                  // development switch: "groups" = the kernel above
  const bool by_rows = Dp <= 128 && (((uintptr_t)Z) & 15) == 0 && ldz % 4 == 0 && !(kenv && strcmp(kenv, "groups") == 0);
  const int64_t rows_per_block = by_rows ? 256 : 256 / G;
  blocks = ceil_div(N, rows_per_block);
  const int64_t cap = (int64_t)ctx->num_cu * 8;
  if (blocks > cap) blocks = cap;
  int rc = ctx->red.ensure((size_t)blocks * J * sizeof(Best));
  if (rc) return rc;
  Best* part = (Best*)ctx->red.ptr;
  if (by_rows) {
#define KMR(DD)                                                                                              \
  hipLaunchKernelGGL(kmeans_pass_rows_kernel<DD>, dim3((unsigned)blocks), dim3(256), 0, s, Z, ldz, N, phi, ldphi, \
                     cols, J, centres, first, dist2, part)
    switch (Dp) {
      case 32: KMR(32); break;
      case 64: KMR(64); break;
      default: KMR(128); break;
    }
#undef KMR
    BNPVI_LAUNCH_CHECK();
    hipLaunchKernelGGL(kmeans_final_multi_kernel, dim3(J), dim3(256), 0, s, part, (const Best64*)nullptr,
                       (int)blocks, J, best_row, best_score);
    BNPVI_LAUNCH_CHECK();
    return BNPVI_OK;
  }
#define KMM(GG)                                                                                          \
  hipLaunchKernelGGL(kmeans_pass_multi_kernel<GG>, dim3((unsigned)blocks), dim3(256), 0, s, Z, ldz, N, phi, \
                     ldphi, cols, J, centres, first, dist2, part)
  switch (G) {
    case 8: KMM(8); break;
    case 16: KMM(16); break;
    case 32: KMM(32); break;
    default: KMM(64); break;
  }
#undef KMM
  BNPVI_LAUNCH_CHECK();
  hipLaunchKernelGGL(kmeans_final_multi_kernel, dim3(J), dim3(256), 0, s, part, (const Best64*)nullptr,
                     (int)blocks, J, best_row, best_score);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

// ------------------------------------------------------------------------------------------
// out[n][j] = scale[j] * in[n][src[j]]  (split: phi_parent/num_splits; merge: drop columns)
// ------------------------------------------------------------------------------------------
// A thread owns FOUR consecutive output columns (one 16-byte store per row) and walks rows: no per-element
// division, the columns' sources and scales stay in registers when a row has at most 256 quads.  (The first
// version -- one element per thread, a 64-bit division each -- ran at 1.2 TB/s: 1.3 ms per 2M x 96 matrix.)
__global__ __launch_bounds__(256) void remap_columns_kernel(const float* __restrict__ in, int ld_in,
                                                            float* __restrict__ out, int ld_out, int64_t N,
                                                            const int32_t* __restrict__ src,
                                                            const float* __restrict__ scale) {
  const int Q = ld_out >> 2;                       // ld_out is a multiple of 4 (checked by the launcher)
  if (Q <= 256) {
    const int rpb = 256 / Q;                       // rows per workgroup and trip
    const int rl = (int)threadIdx.x / Q, q = (int)threadIdx.x % Q;
    if (rl >= rpb) return;
    int sj[4];
    float sc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sj[c] = src[4 * q + c];
      sc[c] = scale[4 * q + c];
    }
    for (int64_t n = (int64_t)blockIdx.x * rpb + rl; n < N; n += (int64_t)gridDim.x * rpb) {
      const float* row = in + n * ld_in;
      float4 v;
      v.x = sj[0] >= 0 ? sc[0] * row[sj[0]] : 0.f;
      v.y = sj[1] >= 0 ? sc[1] * row[sj[1]] : 0.f;
      v.z = sj[2] >= 0 ? sc[2] * row[sj[2]] : 0.f;
      v.w = sj[3] >= 0 ? sc[3] * row[sj[3]] : 0.f;
      *reinterpret_cast<float4*>(out + n * ld_out + 4 * q) = v;
    }
    return;
  }
  for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
    const float* row = in + n * ld_in;
    for (int q = threadIdx.x; q < Q; q += 256) {
      const int4 s4 = *reinterpret_cast<const int4*>(src + 4 * q);
      const float4 c4 = *reinterpret_cast<const float4*>(scale + 4 * q);
      float4 v;
      v.x = s4.x >= 0 ? c4.x * row[s4.x] : 0.f;
      v.y = s4.y >= 0 ? c4.y * row[s4.y] : 0.f;
      v.z = s4.z >= 0 ? c4.z * row[s4.z] : 0.f;
      v.w = s4.w >= 0 ? c4.w * row[s4.w] : 0.f;
      *reinterpret_cast<float4*>(out + n * ld_out + 4 * q) = v;
    }
  }
}

int launch_remap_columns(bnpvi_ctx* ctx, const float* in, int ld_in, float* out, int ld_out,
                         int64_t N, int Lnew, const int32_t* src_host, const float* scale_host,
                         hipStream_t s) {
  int rc = ctx->cols.ensure((size_t)ld_out * 8);
  if (rc) return rc;
  std::vector<int32_t> src(ld_out, -1);
  std::vector<float> scale(ld_out, 0.f);
  for (int j = 0; j < Lnew; ++j) {
    src[j] = src_host[j];
    scale[j] = scale_host[j];
  }
  int32_t* dsrc = (int32_t*)ctx->cols.ptr;
  float* dscale = (float*)((char*)ctx->cols.ptr + (size_t)ld_out * 4);
  // pageable-host copies: the runtime stages them before returning, so the vectors may die
  BNPVI_HIP(hipMemcpyAsync(dsrc, src.data(), (size_t)ld_out * 4, hipMemcpyHostToDevice, s));
  BNPVI_HIP(hipMemcpyAsync(dscale, scale.data(), (size_t)ld_out * 4, hipMemcpyHostToDevice, s));
  BNPVI_HIP(hipStreamSynchronize(s));
  if (N == 0) return BNPVI_OK;
  const int Q = ld_out / 4;
  int64_t blocks = Q <= 256 ? ceil_div(N, (int64_t)(256 / Q)) : N;
  const int64_t cap = (int64_t)ctx->num_cu * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(remap_columns_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, ld_in, out,
                     ld_out, N, dsrc, dscale);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

// ------------------------------------------------------------------------------------------
// gamma re-layout after tree surgery: out[v][k] = in[v][src[k]] or the default (1, GAMMA)
// ------------------------------------------------------------------------------------------
__global__ void gamma_remap_kernel(const double2* __restrict__ in, int K_old, double2* __restrict__ out,
                                   int K_new, int V, const int32_t* __restrict__ src,
                                   double gamma_hyper) {
  const int64_t total = (int64_t)V * K_new;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = e / K_new;
    const int k = (int)(e % K_new);
    const int sk = src[k];
    double2 g;
    g.x = 1.0;
    g.y = gamma_hyper;
    if (sk >= 0) g = in[v * K_old + sk];
    out[e] = g;
  }
}

int launch_gamma_remap(bnpvi_ctx* ctx, const double* in, int K_old, double* out, int K_new, int V,
                       const int32_t* src_host, double gamma_hyper, hipStream_t s) {
  int rc = ctx->cols.ensure((size_t)K_new * 4);
  if (rc) return rc;
  int32_t* dsrc = (int32_t*)ctx->cols.ptr;
  BNPVI_HIP(hipMemcpyAsync(dsrc, src_host, (size_t)K_new * 4, hipMemcpyHostToDevice, s));
  BNPVI_HIP(hipStreamSynchronize(s));
  int64_t blocks = ceil_div((int64_t)V * K_new, 256);
  const int64_t cap = (int64_t)ctx->num_cu * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(gamma_remap_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const double2*)in,
                     K_old, (double2*)out, K_new, V, dsrc, gamma_hyper);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

// ------------------------------------------------------------------------------------------
// Fixed-order reduction of per-workgroup fp64 slabs: 32 outputs x 32 slab groups per block.
// Group g sums slabs g, g+32, ... in ascending order, the 32 group sums are combined in order
// (fixed summation tree => bit-reproducible).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const double* __restrict__ slab, int nslab,
                                                           int64_t elems, double* __restrict__ out) {
  __shared__ double sh[32][33];
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t e = (int64_t)blockIdx.x * 32 + lane;
  double acc = 0.0;
  if (e < elems) {
    int g = grp;
    for (; g + 96 < nslab; g += 128) {     // 4 independent loads in flight per thread
      const double a0 = slab[(int64_t)g * elems + e];
      const double a1 = slab[(int64_t)(g + 32) * elems + e];
      const double a2 = slab[(int64_t)(g + 64) * elems + e];
      const double a3 = slab[(int64_t)(g + 96) * elems + e];
      acc += a0;
      acc += a1;
      acc += a2;
      acc += a3;
    }
    for (; g < nslab; g += 32) acc += slab[(int64_t)g * elems + e];
  }
  sh[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && e < elems) {
    double t = sh[0][lane];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += sh[k][lane];
    out[e] = t;
  }
}

// ------------------------------------------------------------------------------------------
// ||alpha_k - z_n||_2 for the nodes k on the root -> leaf path of frame n (the ranking key of
// src/create_hierarchy.py:50-51).  One wave per frame: walk up from the assigned leaf, lanes
// over the dimensions, fp64 sums of fp64 differences (the reference is NumPy fp64).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void path_distances_kernel(
    const float* __restrict__ Z, int ldz, int64_t N, int D, const double* __restrict__ alpha,
    const int32_t* __restrict__ assign, const int32_t* __restrict__ node_of_leaf,
    const int32_t* __restrict__ parent, const int32_t* __restrict__ depth, int L, int max_depth,
    int32_t* __restrict__ node_out, double* __restrict__ dist_out) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  int leaf = assign[n];
  leaf = leaf < 0 ? 0 : (leaf >= L ? L - 1 : leaf);
  int k = node_of_leaf[leaf];
  const int dleaf = depth[k];
  for (int j = lane; j < max_depth; j += 64)
    if (j > dleaf) {
      node_out[n * max_depth + j] = -1;
      dist_out[n * max_depth + j] = 0.0;
    }
  for (int j = dleaf; j >= 0 && k >= 0; --j) {
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) {
      const double df = alpha[(int64_t)k * D + d] - (double)Z[n * ldz + d];
      acc = fma(df, df, acc);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
    if (lane == 0 && j < max_depth) {
      node_out[n * max_depth + j] = k;
      dist_out[n * max_depth + j] = sqrt(acc);
    }
    k = parent[k];
  }
}

int launch_path_distances(bnpvi_ctx* ctx, const bnpvi_tree* t, const float* Z, int ldz, int64_t N, int D,
                          const double* alpha, const int32_t* assign, int max_depth, int32_t* node_out,
                          double* dist_out, hipStream_t s) {
  (void)ctx;                             // N < 2^33 (check_common): the grid fits
  hipLaunchKernelGGL(path_distances_kernel, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, s, Z, ldz, N, D, alpha,
                     assign, t->d_node_of_leaf, t->d_parent, t->d_depth, t->L, max_depth, node_out, dist_out);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

int launch_slab_reduce(const double* slab, int nslab, int64_t elems, double* out, hipStream_t s) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(elems, 32)), dim3(1024), 0, s, slab,
                     nslab, elems, out);
  BNPVI_LAUNCH_CHECK();
  return BNPVI_OK;
}

}  // namespace bnpvi
