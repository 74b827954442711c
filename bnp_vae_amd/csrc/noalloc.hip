// bnpvi_leaf_stats_noalloc: the M-step data pass for callers that cannot allocate device memory
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It runs
// the same kernels as bnpvi_leaf_stats; the only device memory those kernels use beyond the
// caller's inputs and outputs -- the per-workgroup partial-sum slabs, [workgroups][L][W] doubles,
// which a second kernel sums in a fixed order -- lives in the static array below instead of in the
// context's hipMalloc'ed workspace.  Every GPU the library runs on holds its own copy of the array
// (kScratchBytes) from the moment the code object is loaded until the process ends.
#include "common.h"

namespace bnpvi {

constexpr size_t kScratchBytes = size_t(32) << 20;       // 32 MiB: config 2 (1M frames, L = D = 64) needs 16.3 MB

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// `scratch` is a host-side stand-in here, not a device address: the address on the CURRENT device
// comes from hipGetSymbolAddress, once per device (bnpvi_ctx_create calls this with its device
// current and keeps the result in the context; every use below goes through that pointer).
int static_scratch_address(void** ptr, size_t* bytes) {
  void* p = nullptr;
  BNPVI_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  if (p == nullptr) {
    set_error("hipGetSymbolAddress returned a null address for the static scratch array");
    return BNPVI_EHIP;
  }
  *ptr = p;
  *bytes = kScratchBytes;
  return BNPVI_OK;
}

}  // namespace bnpvi

using namespace bnpvi;

extern "C" int bnpvi_leaf_stats_noalloc(bnpvi_ctx* ctx, const float* Z, int ldz, const float* phi, int ldphi,
                                        const int32_t* vid, int64_t N, int D, int L, int V, int want_q,
                                        double* stats, double* mass, void* stream) {
  BNPVI_REQUIRE(ctx != nullptr, "bnpvi_leaf_stats_noalloc: ctx is NULL");
  BNPVI_REQUIRE(N >= 0 && N < (int64_t(1) << 33), "bnpvi_leaf_stats_noalloc: N=%lld", (long long)N);
  const int Dp = bnpvi_padded_dim(D);
  BNPVI_REQUIRE(D >= 1 && Dp > 0, "bnpvi_leaf_stats_noalloc: D=%d unsupported (1..256)", D);
  BNPVI_REQUIRE(ldz >= Dp && ldz % 4 == 0,
                "bnpvi_leaf_stats_noalloc: ldz=%d must be >= bnpvi_padded_dim(D)=%d and a multiple of 4", ldz, Dp);
  BNPVI_REQUIRE(Z && phi && stats, "bnpvi_leaf_stats_noalloc: NULL device pointer");
  BNPVI_REQUIRE(L >= 1 && ldphi == bnpvi_padded_leaves(L),
                "bnpvi_leaf_stats_noalloc: ldphi=%d must equal bnpvi_padded_leaves(L=%d)", ldphi, L);
  BNPVI_REQUIRE(mass == nullptr || (vid != nullptr && V >= 1), "bnpvi_leaf_stats_noalloc: mass requested without vid / V");
  BNPVI_REQUIRE(ctx->static_scratch != nullptr, "bnpvi_leaf_stats_noalloc: the context has no static scratch address");
  BNPVI_HIP(hipSetDevice(ctx->device));
  // the kernels take the slab base as an argument, exactly as in bnpvi_leaf_stats; here it is the
  // per-device address of `scratch`.  Every slab element a launch reads it has written in the same
  // launch, so nothing has to start at zero (the per-video masses, an OUTPUT of the caller, are
  // cleared with hipMemsetAsync -- ALWAYS here: the "bnpvi_params_prior left this buffer zeroed"
  // shortcut of bnpvi_leaf_stats is decided from host-side context state, which a captured graph
  // would freeze at capture time and then accumulate into the masses on every replay).  The
  // HIP-event timers of the measurement hooks create events on first use, so they stay off for
  // this call.
  ctx->mass_clean = nullptr;
  const bool was_profiling = ctx->profiling;
  ctx->profiling = false;
  ctx->slab_override = ctx->static_scratch;
  ctx->slab_override_bytes = ctx->static_scratch_bytes;
  const int rc = launch_leaf_stats(ctx, Z, ldz, phi, ldphi, vid, N, D, L, V, want_q, stats, mass, (hipStream_t)stream);
  ctx->slab_override = nullptr;
  ctx->slab_override_bytes = 0;
  ctx->profiling = was_profiling;
  return rc;
}
