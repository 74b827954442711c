// Shared host-side declarations for libbnpvi (gfx950 only; no CUDA/portability layer).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "../../include/bnpvi.h"

namespace bnpvi {

void set_error(const char* fmt, ...);

#define BNPVI_HIP(call)                                                              \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      ::bnpvi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                         __FILE__, __LINE__);                                        \
      return BNPVI_EHIP;                                                             \
    }                                                                                \
  } while (0)

#define BNPVI_REQUIRE(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::bnpvi::set_error(__VA_ARGS__);           \
      return BNPVI_EINVAL;                       \
    }                                            \
  } while (0)

#define BNPVI_LAUNCH_CHECK()                                                           \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) {                                                            \
      ::bnpvi::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),    \
                         __FILE__, __LINE__);                                          \
      return BNPVI_EHIP;                                                               \
    }                                                                                  \
  } while (0)

constexpr int kTile = 32;          // MFMA 32x32 tile edge
constexpr int kNumCU = 256;        // MI355X
constexpr int kMaxLdsBytes = 160 * 1024;

// Development switches (kernel-path overrides used by the tools/ scripts) exist only in builds made
// with -DBNPVI_DEV_KNOBS; the shipped library never reads the environment on a launch path.
inline const char* dev_getenv(const char* name) {
#ifdef BNPVI_DEV_KNOBS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grow-only device scratch buffer.
struct Scratch {
  void* ptr = nullptr;
  size_t bytes = 0;
  int ensure(size_t need);
  void release();
};

// HIP-event timer for one kernel kind: events are recorded on the launch stream right around
// the kernel (bench.py's roofline numbers come from here, see bnpvi_profile_read).
struct KernelTimer {
  static constexpr int kMax = 1024;
  std::vector<hipEvent_t> ev;   // start/stop pairs
  int used = 0;
  bool begin(hipStream_t s);    // returns false when full or event creation failed
  void end(hipStream_t s);
  int read(double* total_ms, int* count);
  void release();
};

}  // namespace bnpvi

struct bnpvi_ctx {
  int device = 0;
  int num_cu = bnpvi::kNumCU;
  bnpvi::Scratch slabs;      // partial-sum slabs (fp64)
  bnpvi::Scratch pack;       // packed leaf parameters for the E-step (fp32)
  bnpvi::Scratch part;       // per-frame softmax partials of the streamed E-step
  bnpvi::Scratch zpack;      // frame-side operand fragments of the streamed E-step (as large as Z)
  bnpvi::Scratch red;        // arg-max partials of the k-means pass
  bnpvi::Scratch cols;       // column maps of remap_columns
  void* small = nullptr;     // 256 bytes of device scalars, allocated with the context (max|Z| of the wide M-step)
  bnpvi::KernelTimer timer[2];   // BNPVI_PROF_ESTEP / BNPVI_PROF_MSTEP main kernels
  bool profiling = false;
  // leaf operands packed ahead of the E-step by bnpvi_params_prior (one use)
  const bnpvi_tree* ops_tree = nullptr;
  const double* ops_alpha = nullptr;
  int ops_D = 0;
  // per-video mass buffer that bnpvi_params_prior left zeroed (bnpvi_leaf_stats skips its memset)
  // the per-video mass buffer the last bnpvi_params_prior left zeroed (pointer, element count, stream): the
  // next bnpvi_leaf_stats on exactly that buffer, size and stream skips its memset
  const double* mass_clean = nullptr;
  size_t mass_clean_elems = 0;
  hipStream_t mass_clean_stream = nullptr;
  // this device's address of the static no-allocation scratch array (noalloc.hip), resolved once in
  // bnpvi_ctx_create; while slab_override is set the statistics kernels put their partial-sum
  // slabs there instead of in `slabs` (bnpvi_leaf_stats_noalloc)
  void* static_scratch = nullptr;
  size_t static_scratch_bytes = 0;
  void* slab_override = nullptr;
  size_t slab_override_bytes = 0;
};

struct bnpvi_tree {
  bnpvi_ctx* ctx = nullptr;
  int K = 0, L = 0, Lp = 0, nlevels = 0;
  int internal_levels = 0;   // levels 0 .. internal_levels-1 hold at least one internal node (deeper ones only leaves)
  // host copies (pre-order node numbering, DFS leaf numbering)
  std::vector<int32_t> parent, depth, leaf_of_node, node_of_leaf, leaf_lo, leaf_hi;
  std::vector<int32_t> child_ptr, child_idx, level_ptr, level_nodes;
  // device copies
  int32_t *d_parent = nullptr, *d_depth = nullptr, *d_leaf_of_node = nullptr,
          *d_node_of_leaf = nullptr, *d_leaf_lo = nullptr, *d_leaf_hi = nullptr,
          *d_child_ptr = nullptr, *d_child_idx = nullptr, *d_level_nodes = nullptr,
          *d_level_ptr = nullptr;
  int32_t* d_block = nullptr;   // the ONE allocation all device copies live in (uploaded with one copy)
};

// kernels' launchers (one per .hip file)
namespace bnpvi {

int launch_leaf_stats(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                      const int32_t* vid, int64_t N, int D, int L, int V, int want_q,
                      double* stats, double* mass, hipStream_t s);

int launch_node_params(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* stats, int W, int D,
                       double sB, double sZ, const double* alpha0, double* alpha, double* sigma,
                       hipStream_t s);

int launch_gamma_update(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* mass, int V,
                        double gamma_hyper, double* gamma, hipStream_t s);

int launch_prior_table(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* gamma,
                       const double* sigma, int V, double sZ, float* prior, double* on,
                       double* before, hipStream_t s);

// gamma-step + prior table in one launch (per-video work, no collective in between)
int launch_gamma_prior(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* mass, const double* sigma, int V,
                       double gamma_hyper, double sZ, double* gamma, float* prior, hipStream_t s);

// node parameters + gamma + prior table (+ leaf operands for the next E-step) ; small trees: one launch
int launch_params_prior(bnpvi_ctx* ctx, const bnpvi_tree* t, const double* stats, int W, int D, double sB,
                        double sZ, const double* alpha0, double* alpha, double* sigma, double* mass, int V,
                        double gamma_hyper, double* gamma, float* prior, hipStream_t s);

int launch_estep(bnpvi_ctx* ctx, const bnpvi_tree* t, const float* Z, int ldz,
                 const int32_t* vid, int64_t N, int D, const double* alpha, const float* prior,
                 double sZ, int mode, float* phi, float* logphi, int32_t* assign,
                 double* estats, hipStream_t s);

int launch_path_distances(bnpvi_ctx* ctx, const bnpvi_tree* t, const float* Z, int ldz, int64_t N, int D,
                          const double* alpha, const int32_t* assign, int max_depth, int32_t* node_out,
                          double* dist_out, hipStream_t s);
int launch_kmeans_first_nested(bnpvi_ctx* ctx, const bnpvi_tree* t, const float* phi, int ldphi, int64_t N,
                               const int32_t* cols, int J, int64_t* best_row, double* best_score, hipStream_t s);
int launch_kmeans_pass_multi(bnpvi_ctx* ctx, const float* Z, int ldz, int64_t N, int D, const float* phi,
                             int ldphi, int L, const int32_t* cols, int J, const float* centres, int first,
                             float* dist2, int64_t* best_row, double* best_score, hipStream_t s);

int launch_remap_columns(bnpvi_ctx* ctx, const float* in, int ld_in, float* out, int ld_out,
                         int64_t N, int Lnew, const int32_t* src_host, const float* scale_host,
                         hipStream_t s);

int launch_gamma_remap(bnpvi_ctx* ctx, const double* in, int K_old, double* out, int K_new, int V,
                       const int32_t* src_host, double gamma_hyper, hipStream_t s);

// Partial-sum slab workspace of one statistics launch: the context's grow-only buffer, or -- inside
// an entry point that may not allocate -- the region that entry point provides (an error when the
// launch needs more than it holds).
int slab_workspace(bnpvi_ctx* ctx, size_t need, double** out);
// Most slabs of `slab_bytes` each a launch may use (unbounded with the context's own buffer).  A
// launcher that would use more frames its waves' chunks longer instead, up to kMaxChunk frames.
inline int64_t slab_cap(const bnpvi_ctx* ctx, size_t slab_bytes) {
  if (!ctx->slab_override) return INT64_MAX;
  return (int64_t)(ctx->slab_override_bytes / (slab_bytes ? slab_bytes : 1));
}

// noalloc.hip: address of the static scratch array on the CURRENT device (hipGetSymbolAddress)
int static_scratch_address(void** ptr, size_t* bytes);

// out[e] = sum_g slab[g][e], g ascending within 8 interleaved groups, groups combined in order
// (fixed summation tree => bit-reproducible)
int launch_slab_reduce(const double* slab, int nslab, int64_t elems, double* out, hipStream_t s);

}  // namespace bnpvi
