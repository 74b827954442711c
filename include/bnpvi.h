/*
 * bnpvi.h -- C ABI of the MI355X-native (gfx950) variational-inference step of bnp-vae.
 *
 * One shared library (libbnpvi.so, built from bnp_vae_amd/csrc/ by hipcc) exports exactly the
 * entry points below.  They are what the Python drop-in `bnp_vae_amd.var_inf.VarInf` binds
 * through ctypes (bnp_vae_amd/_lib.py), and what any other host (C, C++, a cgo/JNI stub)
 * would bind: plain pointers and sizes, no torch types.  The reference has no FFI of its
 * own (it is NumPy); each entry point cites the reference routine whose
 * data pass it replaces as /root/reference/src/<file>:<lines>.
 *
 * Conventions
 *   - every function returns 0 (BNPVI_OK) or a BNPVI_E* code; bnpvi_last_error() gives a
 *     thread-local message for the last failure.
 *   - pointers named *_dev are DEVICE pointers owned by the caller (hipMalloc / torch);
 *     *_host are host pointers.  `stream` is a hipStream_t passed as void* (NULL = default
 *     stream).  Calls only enqueue work; results are ready after the stream is synchronised.
 *   - latents Z: float32, row-major, one row per frame, frames of one video contiguous
 *     (src/var_inf.py:72-82), row stride ldz >= bnpvi_padded_dim(D) with columns
 *     D..padded-1 ZERO.
 *   - responsibilities phi: float32, row-major [N][ldphi], ldphi = bnpvi_padded_leaves(L),
 *     column j = j-th leaf in depth-first order (the order every reference recursion visits
 *     leaves); padded columns hold 0.
 *   - vid: int32 [N], index of the frame's video in [0, V) (src/var_inf.py:318-322 does
 *     this lookup with list.index per frame).
 *   - nodes are numbered in PRE-ORDER (the order of get_nodes_list, src/var_inf.py:465-472).
 *   - sizes: 1 <= D <= 256; N < 2^33 frames in every entry point (each regime's grids and 64-bit row
 *     offsets hold up to there; BNPVI_EINVAL beyond).  Trees: any K for the M-step and the tree kernels; the
 *     E-step's streamed regime (more than 64 leaves with hard mode / arg-max, or leaf operands beyond the
 *     LDS) needs ceil(padded L / 256) <= 65535 leaf windows.  The no-allocation entry point has its own,
 *     smaller limit (see there).
 */
#ifndef BNPVI_H
#define BNPVI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNPVI_ABI_VERSION 1

enum {
  BNPVI_OK = 0,
  BNPVI_EINVAL = 1,       /* bad argument / shape the kernels do not support */
  BNPVI_EHIP = 2,         /* a HIP runtime call failed */
  BNPVI_ENOMEM = 3,
  BNPVI_EUNSUPPORTED = 4
};

/* E-step modes */
enum {
  BNPVI_ESTEP_SOFT = 0,   /* softmax responsibilities, src/var_inf.py:313-357,383-393 */
  BNPVI_ESTEP_HARD = 1,   /* one-hot nearest leaf, src/var_inf.py:126-132 (first-call init);
                             same rule as compute_acc_frame_avg_dynamic.py:49-52 */
  BNPVI_ESTEP_OPERANDS_READY = 0x100, /* OR-ed into the mode: the leaf operands of THIS alpha_dev
                             were prepared by the preceding bnpvi_params_prior (ignored, and
                             the operands rebuilt, when that is not the case) */
  BNPVI_ESTEP_STATS_S0 = 0x200        /* OR-ed into the mode: the caller reads only S0' of estats_dev; S2 may be
                             left 0 (it is, by the one-pass kernel for 64 < L <= 512 leaves) -- the split test's
                             S2 then comes from bnpvi_split_gate + bnpvi_leaf_s2 for the few leaves that pass
                             the reference's mass gate (src/var_inf.py:217) */
};

typedef struct bnpvi_ctx bnpvi_ctx;     /* per-device workspace (partial-sum slabs, scratch) */
typedef struct bnpvi_tree bnpvi_tree;   /* flattened Node hierarchy on the device (SoA) */

int bnpvi_abi_version(void);
const char* bnpvi_last_error(void);

/* D padded to what the MFMA k-loop consumes; L padded to a whole 32-leaf MFMA tile. */
int bnpvi_padded_dim(int D);
int bnpvi_padded_leaves(int L);

/* ---- context ----------------------------------------------------------------------- */
int bnpvi_ctx_create(int device, bnpvi_ctx** out);
int bnpvi_ctx_destroy(bnpvi_ctx* ctx);

/* ---- tree: replaces the Node object graph of src/node.py:6-17 on the device ----------
 * parent_host[k] = pre-order index of node k's parent (-1 for the root, k=0);
 * is_leaf_host[k] = Node.isLeafNode.  Children order = order of appearance in pre-order
 * (= Node.children order).  Derived arrays (depth, DFS leaf numbering, leaf ranges
 * [lo,hi) per node, children CSR, level lists) are built here and uploaded. */
int bnpvi_tree_create(bnpvi_ctx* ctx, int K, const int32_t* parent_host,
                      const uint8_t* is_leaf_host, bnpvi_tree** out);
int bnpvi_tree_destroy(bnpvi_tree* tree);
int bnpvi_tree_num_leaves(const bnpvi_tree* tree);
/* copies the pre-order index of every leaf (DFS leaf order) into out_host[L] */
int bnpvi_tree_leaf_nodes(const bnpvi_tree* tree, int32_t* out_host);

/* ---- M-step data pass: responsibility-weighted sufficient statistics ------------------
 * Replaces the per-leaf NumPy passes of compute_sigma_node / compute_alpha_node
 * (src/var_inf.py:265-297: sum(phi), sum(phi*Z)) and the per-video sums of
 * compute_gamma_node (src/var_inf.py:400-403).
 *   stats_dev  : double [L][W], W = 1 + D (+ D when want_q):  [S0 | S1[0..D) | Q[0..D)]
 *                S0 = sum_n phi[n,l];  S1 = sum_n phi[n,l] z_n;  Q = sum_n phi[n,l] z_n^2
 *                (Q feeds the split test, src/var_inf.py:219-220).  This is the buffer a
 *                multi-GPU caller all-reduces (RCCL, sum, fp64) before bnpvi_node_params.
 *   mass_dev   : double [V][L] or NULL:  m[v,l] = sum_{n in video v} phi[n,l]
 * fp32 MFMA products, fp32 partial sums over <= 8192 frames, fp64 accumulation above that;
 * partial slabs are combined in a fixed order (bit-reproducible run to run).  The products are exact
 * (operands split into three bf16 pieces, six matrix products) except for trees whose padded leaves x padded
 * dims reach 512 x 256, where the matrix pipe bounds the pass: there the operands are scaled by powers of two
 * and split into two f16 pieces good to 2^-22 (three products, one extra read of Z for max|Z|); measured
 * against fp64 the statistics are as close as with the exact split (the fp32 partial sums dominate both). */
int bnpvi_leaf_stats(bnpvi_ctx* ctx, const float* Z_dev, int ldz, const float* phi_dev,
                     int ldphi, const int32_t* vid_dev, int64_t N, int D, int L, int V,
                     int want_q, double* stats_dev, double* mass_dev, void* stream);

/* ---- the same data pass for callers that cannot allocate ----------------------------------
 * No counterpart in the reference.  For callers that cannot call hipMalloc: code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation.  Same arguments and kernels as
 * bnpvi_leaf_stats; the results are bit-identical to it while the launch geometry is the same, and equal to
 * fp32 partial-sum rounding (about 2e-6 relative, still reproducible run to run) when the 32 MiB cap makes
 * the launch use fewer workgroups with longer frame chunks.  The per-video masses are always cleared by the
 * call itself (no reliance on bnpvi_params_prior having left them zeroed: safe to replay from a captured
 * graph).  The partial-sum slabs of the launch live in a
 * static device array of the library (32 MiB per GPU, held until the process ends) instead of the
 * context's grow-only workspace, and the call makes no hipMalloc / hipFree / allocating library
 * call.  A launch that needs more than the array holds (workgroups x L x (1 + D (+ D)) doubles;
 * config 2 with 1M frames needs 16.3 MB) is rejected with BNPVI_EINVAL and a message naming both
 * sizes.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams,
 * two contexts, two threads) race -- serialise them.  The context must have been created with
 * bnpvi_ctx_create (it resolves the array's address on its device). */
int bnpvi_leaf_stats_noalloc(bnpvi_ctx* ctx, const float* Z_dev, int ldz, const float* phi_dev,
                             int ldphi, const int32_t* vid_dev, int64_t N, int D, int L, int V,
                             int want_q, double* stats_dev, double* mass_dev, void* stream);

/* ---- M-step node update: src/var_inf.py:265-297 (post-order Gauss-Seidel) --------------
 * sigma[k] : leaf  sB + sZ*S0 ; internal (1+#children)*sB
 * alpha[k] : leaf  (alpha_old[parent]*sB + sZ*S1)/sigma ; internal
 *            (alpha_old[parent or alpha0] + sum_children alpha_new)*sB/sigma
 * alpha_dev: double [K][D] in/out (pre-order); sigma_dev: double [K] out;
 * alpha0_dev: double [D] (ALPHA of src/util.py:23) or NULL for zeros. */
int bnpvi_node_params(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* stats_dev,
                      int stats_width, int D, double sigma_b_sqrinv, double sigma_z_sqrinv,
                      const double* alpha0_dev, double* alpha_dev, double* sigma_dev,
                      void* stream);

/* ---- gamma-step: src/var_inf.py:395-416 ------------------------------------------------
 * gamma_dev: double [V][K][2] out;  gamma0 = 1 + mass(subtree), gamma1 = GAMMA + mass of
 * everything before the node among its ancestors' earlier siblings (prefix over DFS leaves).*/
int bnpvi_gamma_update(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* mass_dev, int V,
                       double gamma_hyper, double* gamma_dev, void* stream);

/* ---- stick-breaking log-prior table: src/var_inf.py:303-311,359-381 --------------------
 * prior_dev : float [V][ldp], ldp = bnpvi_padded_leaves(L).  Row v, column l holds
 *             on[l][v] + before[l][v] - 0.5*sZ/sigma[leaf l]  minus the row maximum
 *             (softmax over leaves is shift-invariant per frame; the shift is applied in
 *             fp64 before rounding to fp32), in LOG2 units (multiplied by log2 e: the E-step
 *             exponentiates with exp2).  Padded columns hold -inf.  The table is private to
 *             the (bnpvi_prior_table | bnpvi_gamma_prior, bnpvi_estep) pair.
 * on_dev / before_dev: optional double [V][L] unshifted tables (NULL to skip) -- the two
 *             dicts compute_gamma_sum_on / compute_gamma_sum_before return. */
int bnpvi_prior_table(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* gamma_dev,
                      const double* sigma_dev, int V, double sigma_z_sqrinv, float* prior_dev,
                      double* on_dev, double* before_dev, void* stream);

/* ---- gamma-step + prior table in one launch: src/var_inf.py:395-416 then :359-381 --------
 * Same results as bnpvi_gamma_update followed by bnpvi_prior_table (both outputs are written);
 * one kernel instead of two when the caller runs them back to back, as one iteration of
 * update_variational_parameters does (src/var_inf.py:136-137). */
int bnpvi_gamma_prior(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* mass_dev,
                      const double* sigma_dev, int V, double gamma_hyper, double sigma_z_sqrinv,
                      double* gamma_dev, float* prior_dev, void* stream);

/* ---- middle of one iteration in one call: src/var_inf.py:265-297 then :395-416, :359-381 --
 * bnpvi_node_params + bnpvi_gamma_update + bnpvi_prior_table with the same results.  For small
 * trees all of it is ONE kernel launch, which also prepares the leaf operands of the coming
 * E-step (pass BNPVI_ESTEP_OPERANDS_READY to bnpvi_estep with the same alpha_dev).
 * mass_dev is CONSUMED: it is zeroed once read, which is the state bnpvi_leaf_stats needs it
 * in.  The context remembers (pointer, V * L, stream) of that buffer and the NEXT bnpvi_leaf_stats
 * skips its memset only when all three match; any bnpvi_leaf_stats with a mass buffer forgets it.
 * A caller that writes into the buffer in between, or frees it and allocates another one of the same
 * size at the same address with other contents on the same stream, breaks that contract -- call
 * bnpvi_gamma_prior (which does not record the buffer) or zero it yourself in that case. */
int bnpvi_params_prior(bnpvi_ctx* ctx, const bnpvi_tree* tree, const double* stats_dev,
                       int stats_width, int D, double sigma_b_sqrinv, double sigma_z_sqrinv,
                       const double* alpha0_dev, double* alpha_dev, double* sigma_dev,
                       double* mass_dev, int V, double gamma_hyper, double* gamma_dev,
                       float* prior_dev, void* stream);

/* ---- E-step: src/var_inf.py:313-357,383-393 ---------------------------------------------
 * logit[n,l] = prior[vid[n],l] - 0.5*sZ*||z_n - alpha_l||_2   (UN-squared norm)
 * phi = softmax over leaves.  alpha_dev: double [K][D] (pre-order).
 *   logphi_dev : float [N][ldphi] or NULL     (log responsibilities)
 *   assign_dev : int32 [N] or NULL            (arg-max leaf, first index on ties)
 *   estats_dev : double [L][2] or NULL        S0' = sum_n phi, S2 = sum_n phi*||z-alpha||^2 (left 0 under
 *                                             BNPVI_ESTEP_STATS_S0 where the kernel does not carry it)
 *                                             (mean(phi) and stdev of src/var_inf.py:194,217-220)
 * mode BNPVI_ESTEP_HARD ignores prior and writes the one-hot nearest-leaf matrix.
 * Workspace (owned by the context, grown on demand, freed by bnpvi_ctx_destroy): the packed leaf operands; for
 * trees whose operands do not fit the LDS (thousands of leaves) also per-(frame, 128-leaf chunk) softmax
 * partials and, in soft mode, an image of Z in matrix-operand order (N * ldz * 4 bytes). */
int bnpvi_estep(bnpvi_ctx* ctx, const bnpvi_tree* tree, const float* Z_dev, int ldz,
                const int32_t* vid_dev, int64_t N, int D, const double* alpha_dev,
                const float* prior_dev, double sigma_z_sqrinv, int mode, float* phi_dev,
                float* logphi_dev, int32_t* assign_dev, double* estats_dev, void* stream);

/* ---- split statistic behind the mass gate: src/var_inf.py:217-220 ------------------------------
 * split_nodes computes a leaf's stdev = sqrt(sqrt(S2) / S0') only when mean(phi) >= 1e-2, which at most 100
 * leaves can pass.  For trees whose E-step kernel carries S0' only (BNPVI_ESTEP_STATS_S0):
 *   bnpvi_split_gate : cols_dev[0 .. count) = the leaves l (ascending) with estats_dev[2 l] >= min_mass
 *                      (estats_dev: the E-step's [L][2] table, S0' summed over all ranks; min_mass = 1e-2 * the
 *                      global frame count for the reference's gate); count_dev[0] = min(number passing, cap),
 *                      count_dev[1] = number passing (int32 [2]); cap <= bnpvi_split_max_columns() = 128.
 *   bnpvi_leaf_s2    : s2_dev[j] = sum_n phi[n, cols[j]] * ||z_n - alpha_{cols[j]}||^2 for j < count_dev[0]
 *                      (direct fp32 differences, fp64 above 2048 frames, fixed summation order), 0 for
 *                      count_dev[0] <= j < cap.  Rank-local sums: a multi-GPU caller all-reduces s2_dev.
 * Both read count / cols on the device: no host round trip between the E-step and the statistic. */
int bnpvi_split_max_columns(void);
int bnpvi_split_gate(bnpvi_ctx* ctx, const double* estats_dev, int L, double min_mass, int32_t* cols_dev, int cap,
                     int32_t* count_dev, void* stream);
int bnpvi_leaf_s2(bnpvi_ctx* ctx, const bnpvi_tree* tree, const float* Z_dev, int ldz, int64_t N, int D,
                  const double* alpha_dev, const float* phi_dev, int ldphi, const int32_t* cols_dev,
                  const int32_t* count_dev, int cap, double* s2_dev, void* stream);

/* ---- weighted farthest-point seeding: src/var_inf.py:23-36, 221-224 ------------------------
 * weighted_kmeans picks num_splits centres among the frames: the first is the arg-max of the weights
 * (a leaf's phi column, :29-30), each later one the arg-max of w[n] * min_c ||z_n - c|| over the
 * centres c chosen so far (:31-35; the reference starts the running minimum from a uniform 1/N, i.e.
 * the first distance replaces it).  One call = one pass = one new centre, for J phi columns at once:
 * split_nodes seeds every splitting leaf (:224), one leaf after the other; the seedings are independent,
 * so the leaves of one walk share the passes over Z.  cols_dev[J] = the leaves' phi columns (int32,
 * device); J <= bnpvi_kmeans_max_columns(D) (the padded D / 4 lanes that share a row each own a column).
 *   centres_dev == NULL : first centres = arg-max of each column (phi_dev [N][ldphi]), lowest row on
 *                         ties (np.argmax).  With L > 0 the rows of phi_dev are responsibilities over L
 *                         leaves that sum to 1: among the frames the column dominates (fp32 phi >= 0.5)
 *                         the arg-max is taken the fp64 way, over 1 / (1 + c), c = the row's mass on the
 *                         other L - 1 columns -- frames that tie in fp32 (all those at 1.0, or two at
 *                         0.99976906) are ordered as the reference's fp64 phi orders them; below 0.5 and
 *                         with L = 0 (plain weights, or rows that do not sum to 1): fp32 arg-max.
 *                         dist2_dev is not touched.
 *   centres_dev != NULL : float [J][padded D], column j's newest centre (e.g. gathered rows of Z).
 *                         dist2_dev [N][J] is the running min squared distance (overwritten when
 *                         first != 0); score = w * sqrt(dist2).
 * best_row_dev[J] (int64) and best_score_dev[J] (double) receive each column's arg-max; all-zero
 * scores give row 0 (normalize()'s uniform fallback, :418-425). */
int bnpvi_kmeans_pass_multi(bnpvi_ctx* ctx, const float* Z_dev, int ldz, int64_t N, int D,
                            const float* phi_dev, int ldphi, int L, const int32_t* cols_dev, int J,
                            const float* centres_dev, int first, float* dist2_dev,
                            int64_t* best_row_dev, double* best_score_dev, void* stream);
int bnpvi_kmeans_max_columns(int D);
/* First centres (src/var_inf.py:29-30) for J <= 64 columns of a responsibility matrix that came straight from
 * bnpvi_estep on `tree`, with the reference's own rounding: for every frame a listed leaf dominates (fp32
 * phi >= 0.5) the fp64 sum get_phi_sum(root) of src/var_inf.py:336-343 is replayed in the TREE'S nesting order
 * (x_l = phi_l / phi_col, the column itself exactly 1) and the score is 1 / that sum, as the reference's
 * normalisation divides (:345-349) -- on well-separated data, where many rows read 1.0 or a few ulps below, which
 * of them np.argmax sees first depends on where in that nest the tiny terms are absorbed.  Below 0.5: fp32
 * arg-max.  Lowest row on ties.  Returns BNPVI_EUNSUPPORTED for trees of more than 3072 nodes (the caller then
 * uses bnpvi_kmeans_pass_multi with centres_dev == NULL). */
int bnpvi_kmeans_first_nested(bnpvi_ctx* ctx, const bnpvi_tree* tree, const float* phi_dev, int ldphi,
                              int64_t N, const int32_t* cols_dev, int J, int64_t* best_row_dev,
                              double* best_score_dev, void* stream);

/* ---- distances of every frame to the nodes on the path of its leaf ---------------------------
 * The per-node ranking of src/create_hierarchy.py:40-53 ("top-K closest frames" of a node among
 * the frames assigned to the leaves below it) and the nearest-leaf distance of
 * src/compute_acc_frame_avg_dynamic.py:49-52 need ||alpha_k - z_n||_2 for the nodes k on the
 * root -> leaf path of frame n only.  assign_dev[N] = DFS leaf index of each frame (the arg-min
 * output of bnpvi_estep in HARD mode).  For j < max_depth:
 *   node_dev[n*max_depth + j] = pre-order index of the path node at depth j (root = depth 0),
 *                               -1 below the leaf;
 *   dist_dev[n*max_depth + j] = ||alpha_node - z_n||_2 in fp64 (direct differences), 0 where -1.
 * max_depth must cover the deepest leaf (depth + 1 entries). */
int bnpvi_path_distances(bnpvi_ctx* ctx, const bnpvi_tree* tree, const float* Z_dev, int ldz,
                         int64_t N, int D, const double* alpha_dev, const int32_t* assign_dev,
                         int max_depth, int32_t* node_dev, double* dist_dev, void* stream);

/* ---- column surgery for split / merge: src/var_inf.py:196,232 ---------------------------
 * out[n][j] = scale_host[j] * in[n][src_host[j]]  for j < Lnew; padded columns = 0.
 * split: children take phi_parent/num_splits; merge: pruned leaves' columns are dropped
 * (rows are NOT renormalised, as in the reference). */
int bnpvi_remap_columns(bnpvi_ctx* ctx, const float* in_dev, int ld_in, float* out_dev,
                        int ld_out, int64_t N, int Lnew, const int32_t* src_host,
                        const float* scale_host, void* stream);

/* ---- measurement hooks (no reference counterpart; used by bench.py) ----------------------
 * With profiling on, bnpvi_estep and bnpvi_leaf_stats record a HIP event pair on the launch
 * stream right around their main kernel (the fused / logits E-step kernel, the R^T.Z stats
 * kernel); bnpvi_profile_read waits for them, returns the summed kernel time and the number
 * of launches since the last read, and resets the counter (at most 1024 launches are kept). */
enum { BNPVI_PROF_ESTEP = 0, BNPVI_PROF_MSTEP = 1 };
int bnpvi_profile_enable(bnpvi_ctx* ctx, int on);
int bnpvi_profile_read(bnpvi_ctx* ctx, int which, double* total_ms, int* count);

/* ---- gamma re-layout after split / merge: src/node.py:15-16, src/var_inf.py:208 -----------
 * out[v][k] = in[v][src_host[k]] for src_host[k] >= 0, else the Node.__init__ default
 * (1, GAMMA) that a freshly split child carries until the next gamma-step; a node that
 * absorbed its only child takes that child's gamma (deepcopy at src/var_inf.py:208). */
int bnpvi_gamma_remap(bnpvi_ctx* ctx, const double* in_dev, int K_old, double* out_dev,
                      int K_new, int V, const int32_t* src_host, double gamma_hyper,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BNPVI_H */
