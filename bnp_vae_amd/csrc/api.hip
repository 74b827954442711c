// C-ABI entry points of libbnpvi (declared in include/bnpvi.h): argument validation, the
// per-device context, and the host-side flattening of the Node hierarchy into SoA arrays.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace bnpvi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int Scratch::ensure(size_t need) {
  if (need <= bytes) return BNPVI_OK;
  if (ptr) {
    BNPVI_HIP(hipFree(ptr));
    ptr = nullptr;
    bytes = 0;
  }
  size_t want = need + need / 4;
  hipError_t e = hipMalloc(&ptr, want);
  if (e != hipSuccess) {
    ptr = nullptr;
    set_error("hipMalloc(%zu) for workspace failed: %s", want, hipGetErrorString(e));
    return BNPVI_ENOMEM;
  }
  bytes = want;
  return BNPVI_OK;
}

void Scratch::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  bytes = 0;
}

int slab_workspace(bnpvi_ctx* ctx, size_t need, double** out) {
  if (ctx->slab_override) {
    if (need > ctx->slab_override_bytes) {
      set_error("this launch needs %zu bytes of partial-sum workspace; the no-allocation entry point has %zu "
                "(use the allocating entry point, or fewer frames / a smaller tree per call)",
                need, ctx->slab_override_bytes);
      return BNPVI_EINVAL;
    }
    *out = (double*)ctx->slab_override;
    return BNPVI_OK;
  }
  int rc = ctx->slabs.ensure(need);
  if (rc) return rc;
  *out = (double*)ctx->slabs.ptr;
  return BNPVI_OK;
}

bool KernelTimer::begin(hipStream_t s) {
  if (used >= kMax) return false;
  if ((int)ev.size() < 2 * (used + 1)) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return false;
    if (hipEventCreate(&b) != hipSuccess) {
      (void)hipEventDestroy(a);
      return false;
    }
    ev.push_back(a);
    ev.push_back(b);
  }
  return hipEventRecord(ev[2 * used], s) == hipSuccess;
}

void KernelTimer::end(hipStream_t s) {
  (void)hipEventRecord(ev[2 * used + 1], s);
  ++used;
}

int KernelTimer::read(double* total_ms, int* count) {
  double tot = 0.0;
  for (int i = 0; i < used; ++i) {
    BNPVI_HIP(hipEventSynchronize(ev[2 * i + 1]));
    float ms = 0.f;
    BNPVI_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *count = used;
  used = 0;
  return BNPVI_OK;
}

void KernelTimer::release() {
  for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  ev.clear();
  used = 0;
}

}  // namespace bnpvi

using namespace bnpvi;

extern "C" {

int bnpvi_abi_version(void) { return BNPVI_ABI_VERSION; }

const char* bnpvi_last_error(void) { return g_err; }

int bnpvi_padded_dim(int D) {
  // the E-step K-loop consumes 32 dims per MFMA step: kernels are instantiated for these widths
  const int sizes[] = {32, 64, 128, 256};
  for (int s : sizes)
    if (D <= s) return s;
  return -1;
}

int bnpvi_padded_leaves(int L) { return L <= 0 ? -1 : round_up(L, kTile); }

int bnpvi_ctx_create(int device, bnpvi_ctx** out) {
  BNPVI_REQUIRE(out != nullptr, "bnpvi_ctx_create: out is NULL");
  int n = 0;
  BNPVI_HIP(hipGetDeviceCount(&n));
  BNPVI_REQUIRE(device >= 0 && device < n, "bnpvi_ctx_create: device %d out of range (%d visible)",
                device, n);
  BNPVI_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  BNPVI_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("libbnpvi is built for gfx950 (MI355X) only; device %d is %s", device,
              prop.gcnArchName);
    return BNPVI_EUNSUPPORTED;
  }
  bnpvi_ctx* c = new bnpvi_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : kNumCU;
  // set-up for bnpvi_leaf_stats_noalloc: this device's copy of the static scratch array, resolved
  // here (device current) so that the entry point itself never has to
  int rc = static_scratch_address(&c->static_scratch, &c->static_scratch_bytes);
  if (rc) {
    delete c;
    return rc;
  }
  if (hipMalloc(&c->small, 256) != hipSuccess) {      // device scalars: allocated here so that no entry point has to
    (void)hipGetLastError();
    c->small = nullptr;                               // (the wide M-step then keeps its exact split)
  }
  *out = c;
  return BNPVI_OK;
}

int bnpvi_ctx_destroy(bnpvi_ctx* ctx) {
  if (!ctx) return BNPVI_OK;
  (void)hipSetDevice(ctx->device);
  ctx->slabs.release();
  ctx->pack.release();
  ctx->part.release();
  ctx->zpack.release();
  ctx->red.release();
  ctx->cols.release();
  if (ctx->small) (void)hipFree(ctx->small);
  ctx->timer[0].release();
  ctx->timer[1].release();
  delete ctx;
  return BNPVI_OK;
}

int bnpvi_profile_enable(bnpvi_ctx* ctx, int on) {
  BNPVI_REQUIRE(ctx != nullptr, "bnpvi_profile_enable: ctx is NULL");
  ctx->profiling = on != 0;
  ctx->timer[0].used = ctx->timer[1].used = 0;
  return BNPVI_OK;
}

int bnpvi_profile_read(bnpvi_ctx* ctx, int which, double* total_ms, int* count) {
  BNPVI_REQUIRE(ctx && total_ms && count, "bnpvi_profile_read: NULL argument");
  BNPVI_REQUIRE(which == BNPVI_PROF_ESTEP || which == BNPVI_PROF_MSTEP, "bnpvi_profile_read: which=%d",
                which);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return ctx->timer[which].read(total_ms, count);
}

int bnpvi_tree_create(bnpvi_ctx* ctx, int K, const int32_t* parent, const uint8_t* is_leaf,
                      bnpvi_tree** out) {
  BNPVI_REQUIRE(ctx && parent && is_leaf && out, "bnpvi_tree_create: NULL argument");
  BNPVI_REQUIRE(K >= 1, "bnpvi_tree_create: K=%d", K);
  BNPVI_REQUIRE(parent[0] == -1, "bnpvi_tree_create: node 0 must be the root (parent -1)");
  bnpvi_tree* t = new bnpvi_tree();
  t->ctx = ctx;
  t->K = K;
  t->parent.assign(parent, parent + K);
  t->depth.assign(K, 0);
  std::vector<int> nchild(K, 0);
  for (int k = 1; k < K; ++k) {
    int p = parent[k];
    if (p < 0 || p >= k || is_leaf[p]) {
      set_error("bnpvi_tree_create: node %d has invalid parent %d (pre-order, non-leaf parents)", k, p);
      delete t;
      return BNPVI_EINVAL;
    }
    // pre-order: the parent must be node k-1 or one of its ancestors
    int a = k - 1;
    while (a != p && a > 0) a = parent[a];
    if (a != p) {
      set_error("bnpvi_tree_create: nodes are not in pre-order at node %d", k);
      delete t;
      return BNPVI_EINVAL;
    }
    t->depth[k] = t->depth[p] + 1;
    nchild[p]++;
  }
  for (int k = 0; k < K; ++k) {
    if (!is_leaf[k] && nchild[k] == 0) {
      set_error("bnpvi_tree_create: internal node %d has no children", k);
      delete t;
      return BNPVI_EINVAL;
    }
  }
  // children CSR (children in order of appearance == Node.children order)
  t->child_ptr.assign(K + 1, 0);
  for (int k = 0; k < K; ++k) t->child_ptr[k + 1] = t->child_ptr[k] + nchild[k];
  t->child_idx.assign(K > 1 ? K - 1 : 0, 0);
  {
    std::vector<int> fill(t->child_ptr.begin(), t->child_ptr.end() - 1);
    for (int k = 1; k < K; ++k) t->child_idx[fill[parent[k]]++] = k;
  }
  // DFS leaf numbering and [lo,hi) leaf range per node
  t->leaf_of_node.assign(K, -1);
  t->leaf_lo.assign(K, 0);
  t->leaf_hi.assign(K, 0);
  int L = 0;
  for (int k = 0; k < K; ++k) {
    t->leaf_lo[k] = L;
    if (is_leaf[k]) {
      t->leaf_of_node[k] = L;
      t->node_of_leaf.push_back(k);
      ++L;
    }
  }
  for (int k = K - 1; k >= 0; --k) {
    if (is_leaf[k]) {
      t->leaf_hi[k] = t->leaf_lo[k] + 1;
    } else {
      int last = t->child_idx[t->child_ptr[k + 1] - 1];
      t->leaf_hi[k] = t->leaf_hi[last];
    }
  }
  t->L = L;
  t->Lp = round_up(L, kTile);
  // level lists
  int maxd = 0;
  for (int k = 0; k < K; ++k) maxd = t->depth[k] > maxd ? t->depth[k] : maxd;
  t->nlevels = maxd + 1;
  t->level_ptr.assign(t->nlevels + 1, 0);
  for (int k = 0; k < K; ++k) t->level_ptr[t->depth[k] + 1]++;
  for (int d = 0; d < t->nlevels; ++d) t->level_ptr[d + 1] += t->level_ptr[d];
  t->level_nodes.assign(K, 0);
  {
    std::vector<int> fill(t->level_ptr.begin(), t->level_ptr.end() - 1);
    for (int k = 0; k < K; ++k) t->level_nodes[fill[t->depth[k]]++] = k;
  }
  t->internal_levels = 0;
  for (int k = 0; k < K; ++k)
    if (t->leaf_of_node[k] < 0 && t->depth[k] + 1 > t->internal_levels) t->internal_levels = t->depth[k] + 1;
  BNPVI_HIP(hipSetDevice(ctx->device));
  // One allocation and one copy for the ten index arrays (a tree is rebuilt after every split / merge walk: ten
  // allocations, ten blocking copies and, at destruction, ten hipFree -- each a device-wide wait -- were ~0.3 ms of
  // such a call).  Every array starts on a 64-byte boundary.
  const std::vector<int32_t>* src[10] = {&t->parent, &t->depth, &t->leaf_of_node, &t->node_of_leaf, &t->leaf_lo,
                                         &t->leaf_hi, &t->child_ptr, &t->child_idx, &t->level_nodes, &t->level_ptr};
  int32_t** dst[10] = {&t->d_parent, &t->d_depth, &t->d_leaf_of_node, &t->d_node_of_leaf, &t->d_leaf_lo,
                       &t->d_leaf_hi, &t->d_child_ptr, &t->d_child_idx, &t->d_level_nodes, &t->d_level_ptr};
  size_t off[10], total = 0;
  for (int a = 0; a < 10; ++a) {
    off[a] = total;
    total += (src[a]->size() + 15) / 16 * 16 + 16;     // never empty: every array keeps its own address
  }
  std::vector<int32_t> host(total, 0);
  for (int a = 0; a < 10; ++a)
    if (!src[a]->empty()) memcpy(host.data() + off[a], src[a]->data(), src[a]->size() * sizeof(int32_t));
  if (hipMalloc((void**)&t->d_block, total * sizeof(int32_t)) != hipSuccess) {
    set_error("bnpvi_tree_create: hipMalloc(%zu) failed", total * sizeof(int32_t));
    delete t;
    return BNPVI_ENOMEM;
  }
  if (hipMemcpy(t->d_block, host.data(), total * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("bnpvi_tree_create: upload of the index arrays failed");
    bnpvi_tree_destroy(t);
    return BNPVI_EHIP;
  }
  for (int a = 0; a < 10; ++a) *dst[a] = t->d_block + off[a];
  *out = t;
  return BNPVI_OK;
}

int bnpvi_tree_destroy(bnpvi_tree* t) {
  if (!t) return BNPVI_OK;
  if (t->d_block) (void)hipFree(t->d_block);
  delete t;
  return BNPVI_OK;
}

int bnpvi_tree_num_leaves(const bnpvi_tree* t) { return t ? t->L : -1; }

int bnpvi_tree_leaf_nodes(const bnpvi_tree* t, int32_t* out_host) {
  BNPVI_REQUIRE(t && out_host, "bnpvi_tree_leaf_nodes: NULL argument");
  memcpy(out_host, t->node_of_leaf.data(), sizeof(int32_t) * t->L);
  return BNPVI_OK;
}

static int check_common(const char* fn, bnpvi_ctx* ctx, int64_t N, int D, int ldz) {
  BNPVI_REQUIRE(ctx != nullptr, "%s: ctx is NULL", fn);
  // 2^33 frames: every per-frame grid (N/4 workgroups at the smallest) stays below 2^31; 288 GB of HBM
  // hold < 2^31 frames of the smallest Z + phi anyway
  BNPVI_REQUIRE(N >= 0 && N < (int64_t(1) << 33), "%s: N=%lld (at most 2^33 - 1 frames)", fn, (long long)N);
  int Dp = bnpvi_padded_dim(D);
  BNPVI_REQUIRE(D >= 1 && Dp > 0, "%s: D=%d unsupported (1..256)", fn, D);
  BNPVI_REQUIRE(ldz >= Dp && ldz % 4 == 0,
                "%s: ldz=%d must be >= bnpvi_padded_dim(D)=%d and a multiple of 4", fn, ldz, Dp);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return BNPVI_OK;
}

int bnpvi_leaf_stats(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                     const int32_t* vid, int64_t N, int D, int L, int V, int want_q,
                     double* stats, double* mass, void* stream) {
  int rc = check_common("bnpvi_leaf_stats", ctx, N, D, ldz);
  if (rc) return rc;
  BNPVI_REQUIRE(Z && phi && stats, "bnpvi_leaf_stats: NULL device pointer");
  BNPVI_REQUIRE(L >= 1 && ldphi == bnpvi_padded_leaves(L),
                "bnpvi_leaf_stats: ldphi=%d must equal bnpvi_padded_leaves(L=%d)", ldphi, L);
  BNPVI_REQUIRE(mass == nullptr || (vid != nullptr && V >= 1),
                "bnpvi_leaf_stats: mass requested without vid / V");
  return launch_leaf_stats(ctx, Z, ldz, phi, ldphi, vid, N, D, L, V, want_q, stats, mass,
                           (hipStream_t)stream);
}

int bnpvi_node_params(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* stats, int W, int D,
                      double sB, double sZ, const double* alpha0, double* alpha, double* sigma,
                      void* stream) {
  BNPVI_REQUIRE(ctx && tree && stats && alpha && sigma, "bnpvi_node_params: NULL argument");
  BNPVI_REQUIRE(D >= 1 && W >= 1 + D, "bnpvi_node_params: stats_width=%d < 1+D=%d", W, 1 + D);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_node_params(ctx, tree, stats, W, D, sB, sZ, alpha0, alpha, sigma,
                            (hipStream_t)stream);
}

int bnpvi_gamma_update(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* mass, int V,
                       double gamma_hyper, double* gamma, void* stream) {
  BNPVI_REQUIRE(ctx && tree && mass && gamma, "bnpvi_gamma_update: NULL argument");
  BNPVI_REQUIRE(V >= 1, "bnpvi_gamma_update: V=%d", V);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_gamma_update(ctx, tree, mass, V, gamma_hyper, gamma, (hipStream_t)stream);
}

int bnpvi_prior_table(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* gamma,
                      const double* sigma, int V, double sZ, float* prior, double* on,
                      double* before, void* stream) {
  BNPVI_REQUIRE(ctx && tree && gamma && sigma && prior, "bnpvi_prior_table: NULL argument");
  BNPVI_REQUIRE(V >= 1, "bnpvi_prior_table: V=%d", V);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_prior_table(ctx, tree, gamma, sigma, V, sZ, prior, on, before,
                            (hipStream_t)stream);
}

int bnpvi_gamma_prior(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* mass, const double* sigma,
                      int V, double gamma_hyper, double sZ, double* gamma, float* prior, void* stream) {
  BNPVI_REQUIRE(ctx && tree && mass && sigma && gamma && prior, "bnpvi_gamma_prior: NULL argument");
  BNPVI_REQUIRE(V >= 1, "bnpvi_gamma_prior: V=%d", V);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_gamma_prior(ctx, tree, mass, sigma, V, gamma_hyper, sZ, gamma, prior, (hipStream_t)stream);
}

int bnpvi_params_prior(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* stats, int W, int D, double sB,
                       double sZ, const double* alpha0, double* alpha, double* sigma, double* mass, int V,
                       double gamma_hyper, double* gamma, float* prior, void* stream) {
  BNPVI_REQUIRE(ctx && tree && stats && alpha && sigma && mass && gamma && prior,
                "bnpvi_params_prior: NULL argument");
  BNPVI_REQUIRE(D >= 1 && bnpvi_padded_dim(D) > 0 && W >= 1 + D && V >= 1,
                "bnpvi_params_prior: D=%d stats_width=%d V=%d", D, W, V);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_params_prior(ctx, tree, stats, W, D, sB, sZ, alpha0, alpha, sigma, mass, V, gamma_hyper, gamma,
                             prior, (hipStream_t)stream);
}

int bnpvi_estep(bnpvi_ctx* ctx, const bnpvi_tree* tree, const float* Z, int ldz,
                const int32_t* vid, int64_t N, int D, const double* alpha, const float* prior,
                double sZ, int mode, float* phi, float* logphi, int32_t* assign, double* estats,
                void* stream) {
  int rc = check_common("bnpvi_estep", ctx, N, D, ldz);
  if (rc) return rc;
  BNPVI_REQUIRE(tree && Z && alpha && phi, "bnpvi_estep: NULL argument");
  const int base_mode = mode & ~(BNPVI_ESTEP_OPERANDS_READY | BNPVI_ESTEP_STATS_S0);
  BNPVI_REQUIRE(base_mode == BNPVI_ESTEP_SOFT || base_mode == BNPVI_ESTEP_HARD, "bnpvi_estep: mode=%d", mode);
  BNPVI_REQUIRE(base_mode == BNPVI_ESTEP_HARD || (prior && vid),
                "bnpvi_estep: soft mode needs prior and vid");
  return launch_estep(ctx, tree, Z, ldz, vid, N, D, alpha, prior, sZ, mode, phi, logphi, assign,
                      estats, (hipStream_t)stream);
}

int bnpvi_kmeans_pass_multi(bnpvi_ctx* ctx, const float* Z, int ldz, int64_t N, int D, const float* phi,
                            int ldphi, int L, const int32_t* cols, int J, const float* centres, int first,
                            float* dist2, int64_t* best_row, double* best_score, void* stream) {
  int rc = check_common("bnpvi_kmeans_pass_multi", ctx, N, D, ldz);
  if (rc) return rc;
  BNPVI_REQUIRE(Z && phi && cols && best_row && best_score, "bnpvi_kmeans_pass_multi: NULL argument");
  BNPVI_REQUIRE(centres == nullptr || dist2 != nullptr, "bnpvi_kmeans_pass_multi: dist2 is NULL");
  const int G = bnpvi_padded_dim(D) / 4;
  BNPVI_REQUIRE(N >= 1 && L >= 0 && ldphi >= 1 && ldphi >= L, "bnpvi_kmeans_pass_multi: N=%lld L=%d ldphi=%d",
                (long long)N, L, ldphi);
  BNPVI_REQUIRE(J >= 1 && J <= bnpvi_kmeans_max_columns(D),
                "bnpvi_kmeans_pass_multi: J=%d columns, at most %d per call at D=%d (%d lanes share a row)", J,
                bnpvi_kmeans_max_columns(D), D, G);
  return launch_kmeans_pass_multi(ctx, Z, ldz, N, D, phi, ldphi, L, cols, J, centres, first, dist2, best_row,
                                  best_score, (hipStream_t)stream);
}

int bnpvi_kmeans_first_nested(bnpvi_ctx* ctx, const bnpvi_tree* tree, const float* phi, int ldphi, int64_t N,
                              const int32_t* cols, int J, int64_t* best_row, double* best_score, void* stream) {
  BNPVI_REQUIRE(ctx && tree && phi && cols && best_row && best_score, "bnpvi_kmeans_first_nested: NULL argument");
  BNPVI_REQUIRE(N >= 1 && ldphi >= tree->L, "bnpvi_kmeans_first_nested: N=%lld ldphi=%d (L=%d)", (long long)N, ldphi,
                tree->L);
  BNPVI_REQUIRE(J >= 1 && J <= 64, "bnpvi_kmeans_first_nested: J=%d columns, at most 64 per call", J);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_kmeans_first_nested(ctx, tree, phi, ldphi, N, cols, J, best_row, best_score, (hipStream_t)stream);
}

int bnpvi_kmeans_max_columns(int D) {
  const int G = bnpvi_padded_dim(D) / 4;
  return G < 64 ? G : 64;
}

int bnpvi_path_distances(bnpvi_ctx* ctx, const bnpvi_tree* tree, const float* Z, int ldz, int64_t N, int D,
                         const double* alpha, const int32_t* assign, int max_depth, int32_t* node_out,
                         double* dist_out, void* stream) {
  int rc = check_common("bnpvi_path_distances", ctx, N, D, ldz);
  if (rc) return rc;
  BNPVI_REQUIRE(tree && Z && alpha && assign && node_out && dist_out, "bnpvi_path_distances: NULL argument");
  int deepest = 0;
  for (int l = 0; l < tree->L; ++l) {
    const int d = tree->depth[tree->node_of_leaf[l]];
    deepest = d > deepest ? d : deepest;
  }
  BNPVI_REQUIRE(N >= 1 && max_depth >= deepest + 1 && max_depth <= 64,
                "bnpvi_path_distances: max_depth=%d, the deepest leaf needs %d (at most 64)", max_depth,
                deepest + 1);
  return launch_path_distances(ctx, tree, Z, ldz, N, D, alpha, assign, max_depth, node_out, dist_out,
                               (hipStream_t)stream);
}

int bnpvi_remap_columns(bnpvi_ctx* ctx, const float* in, int ld_in, float* out, int ld_out,
                        int64_t N, int Lnew, const int32_t* src_host, const float* scale_host,
                        void* stream) {
  BNPVI_REQUIRE(ctx && in && out && src_host && scale_host, "bnpvi_remap_columns: NULL argument");
  BNPVI_REQUIRE(Lnew >= 1 && ld_out == bnpvi_padded_leaves(Lnew) && ld_in >= 1 && in != out,
                "bnpvi_remap_columns: Lnew=%d ld_out=%d ld_in=%d (out of place only)", Lnew, ld_out,
                ld_in);
  for (int j = 0; j < Lnew; ++j)
    BNPVI_REQUIRE(src_host[j] >= 0 && src_host[j] < ld_in, "bnpvi_remap_columns: src[%d]=%d", j,
                  src_host[j]);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_remap_columns(ctx, in, ld_in, out, ld_out, N, Lnew, src_host, scale_host,
                              (hipStream_t)stream);
}

int bnpvi_gamma_remap(bnpvi_ctx* ctx, const double* in, int K_old, double* out, int K_new, int V,
                      const int32_t* src_host, double gamma_hyper, void* stream) {
  BNPVI_REQUIRE(ctx && in && out && src_host, "bnpvi_gamma_remap: NULL argument");
  BNPVI_REQUIRE(K_old >= 1 && K_new >= 1 && V >= 1 && in != out,
                "bnpvi_gamma_remap: K_old=%d K_new=%d V=%d (out of place only)", K_old, K_new, V);
  for (int k = 0; k < K_new; ++k)
    BNPVI_REQUIRE(src_host[k] < K_old, "bnpvi_gamma_remap: src[%d]=%d", k, src_host[k]);
  BNPVI_HIP(hipSetDevice(ctx->device));
  return launch_gamma_remap(ctx, in, K_old, out, K_new, V, src_host, gamma_hyper,
                            (hipStream_t)stream);
}

}  // extern "C"
