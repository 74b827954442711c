"""Device-side state of the VI step and the launch sequence of one iteration.

`DeviceVI` owns the device buffers (torch is used for memory, streams and
torch.distributed only) and calls the hand-written gfx950 kernels through the C ABI
(include/bnpvi.h).  One instance per process / GPU.  With a `torch.distributed` process
group, frames are sharded by sample on video boundaries and the per-leaf sufficient
statistics `[S0 | S1 (| Q)]` are combined with ONE all-reduce (RCCL over xGMI) per
iteration; everything else is rank-local.

Reference mapping: the body of one iteration is /root/reference/src/var_inf.py:135-137
(compute_sigma_alpha -> compute_gamma -> compute_phi).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from . import dist as vdist
from ._lib import lib, check


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


class FlatTree(object):
    """Pre-order SoA view of a Node hierarchy (host side) + its device twin."""

    def __init__(self, ctx, parent, is_leaf):
        self.parent = np.ascontiguousarray(parent, dtype=np.int32)
        self.is_leaf = np.ascontiguousarray(is_leaf, dtype=np.uint8)
        self.K = int(self.parent.shape[0])
        h = C.c_void_p()
        check(lib.bnpvi_tree_create(ctx, self.K, self.parent.ctypes.data_as(C.c_void_p),
                                    self.is_leaf.ctypes.data_as(C.c_void_p), C.byref(h)),
              "bnpvi_tree_create")
        self.handle = h
        self.L = lib.bnpvi_tree_num_leaves(h)
        self.Lp = _lib.padded_leaves(self.L)
        self.node_of_leaf = np.zeros(self.L, dtype=np.int32)
        check(lib.bnpvi_tree_leaf_nodes(h, self.node_of_leaf.ctypes.data_as(C.c_void_p)))
        self.depth = np.zeros(self.K, dtype=np.int32)          # pre-order: a parent precedes its children
        for k in range(1, self.K):
            self.depth[k] = self.depth[self.parent[k]] + 1

    def close(self):
        if self.handle is not None:
            lib.bnpvi_tree_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceVI(object):
    def __init__(self, device=None, group=None, sigma_b_sqrinv=1.0, sigma_z_sqrinv=1.0,
                 gamma_hyper=10.0, alpha_hyper=None):
        if not torch.cuda.is_available():
            raise RuntimeError("bnp_vae_amd needs a ROCm GPU (MI355X / gfx950); none is visible "
                               "and there is no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index)
        self.group = group
        self.sB, self.sZ, self.GAMMA = float(sigma_b_sqrinv), float(sigma_z_sqrinv), float(gamma_hyper)
        self.alpha_hyper = None if alpha_hyper is None else np.asarray(alpha_hyper, dtype=np.float64)
        h = C.c_void_p()
        check(lib.bnpvi_ctx_create(self.device.index, C.byref(h)), "bnpvi_ctx_create")
        self.ctx = h
        self.tree = None
        self.N = self.D = self.V = 0
        self.Z = self.vid = self.phi = None
        self.alpha = self.sigma = self.gamma = self.mass = self.prior = None
        self.stats = self.estats = self.alpha0 = None
        self.gate_cols = self.gate_count = self.gate_s2 = None   # split gate: leaves listed, their S2
        self.s2_gated = False        # the last E-step left S2 to gated_s2()
        self.want_q = False
        self.row_offset = 0          # global index of this rank's first frame

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if self.tree is not None:
            self.tree.close()
            self.tree = None
        if self.ctx is not None:
            lib.bnpvi_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def synchronize(self):
        torch.cuda.current_stream(self.device).synchronize()

    # ------------------------------------------------------------------ data
    def set_latents(self, Z, vid, V):
        """Z: (N, D) float32 host array (or device tensor); vid: (N,) int video index."""
        N, D = int(Z.shape[0]), int(Z.shape[1])
        Dp = _lib.padded_dim(D)
        if self.Z is None or tuple(self.Z.shape) != (N, Dp):
            self.Z = torch.zeros((N, Dp), dtype=torch.float32, device=self.device)
        if isinstance(Z, torch.Tensor):
            self.Z[:, :D].copy_(Z.to(torch.float32))
        else:
            self.Z[:, :D].copy_(torch.from_numpy(np.ascontiguousarray(Z, dtype=np.float32)))
        if isinstance(vid, torch.Tensor):
            self.vid = vid.to(device=self.device, dtype=torch.int32).contiguous()
        else:
            self.vid = torch.from_numpy(np.ascontiguousarray(vid, dtype=np.int32)).to(self.device)
        self.N, self.D, self.V, self.Dp = N, D, int(V), Dp
        if self.alpha_hyper is not None:
            self.alpha0 = torch.from_numpy(self.alpha_hyper).to(self.device)

    def set_tree(self, parent, is_leaf, alpha=None, sigma=None):
        """(Re)build the flattened tree; optional node parameters as (K, D) / (K,) fp64."""
        old = self.tree
        self.tree = FlatTree(self.ctx, parent, is_leaf)
        if old is not None:
            old.close()
        K = self.tree.K
        if alpha is not None:
            self.alpha = torch.from_numpy(np.ascontiguousarray(alpha, dtype=np.float64)).to(self.device)
        if sigma is not None:
            self.sigma = torch.from_numpy(np.ascontiguousarray(sigma, dtype=np.float64)).to(self.device)
        if self.alpha is None or self.alpha.shape[0] != K:
            raise ValueError("set_tree: alpha for %d nodes required" % K)
        if self.sigma is None or self.sigma.shape[0] != K:
            self.sigma = torch.ones(K, dtype=torch.float64, device=self.device)
        return self.tree

    def alloc_phi(self):
        self.phi = torch.zeros((self.N, self.tree.Lp), dtype=torch.float32, device=self.device)
        return self.phi

    def set_phi(self, phi_host):
        """phi_host: (N, L) array, DFS leaf order."""
        L, Lp = self.tree.L, self.tree.Lp
        assert phi_host.shape == (self.N, L), (phi_host.shape, self.N, L)
        self.phi = torch.zeros((self.N, Lp), dtype=torch.float32, device=self.device)
        self.phi[:, :L].copy_(torch.from_numpy(np.ascontiguousarray(phi_host, dtype=np.float32)))

    def default_gamma(self):
        K, V = self.tree.K, self.V
        g = torch.empty((V, K, 2), dtype=torch.float64, device=self.device)
        g[:, :, 0] = 1.0
        g[:, :, 1] = self.GAMMA
        self.gamma = g

    def set_gamma(self, gamma_host):
        """gamma_host: (K, V, 2) fp64 (Node.gamma stacked in pre-order)."""
        g = np.ascontiguousarray(np.transpose(gamma_host, (1, 0, 2)), dtype=np.float64)
        self.gamma = torch.from_numpy(g).to(self.device)

    # ------------------------------------------------------------------ kernels
    def leaf_stats(self, want_q=False, want_mass=True):
        """M-step data pass (src/var_inf.py:265-297 sums, :400-403 per-video sums)."""
        L, D = self.tree.L, self.D
        W = 1 + D * (2 if want_q else 1)
        if self.stats is None or tuple(self.stats.shape) != (L, W):
            self.stats = self._empty((L, W), torch.float64)
        if want_mass and (self.mass is None or tuple(self.mass.shape) != (self.V, L)):
            self.mass = self._empty((self.V, L), torch.float64)
        if self.N == 0 or self.V == 0:          # a rank without frames (fewer videos than ranks)
            self.stats.zero_()
            self.want_q = want_q
            return self.stats
        check(lib.bnpvi_leaf_stats(self.ctx, _ptr(self.Z), self.Dp, _ptr(self.phi), self.tree.Lp,
                                   _ptr(self.vid), self.N, D, L, self.V, 1 if want_q else 0,
                                   _ptr(self.stats), _ptr(self.mass if want_mass else None),
                                   self._stream()), "bnpvi_leaf_stats")
        self.want_q = want_q
        return self.stats

    def allreduce_stats(self):
        """The single collective of an iteration (RCCL all-reduce, sum, fp64)."""
        vdist.allreduce_sum_(self.stats, self.group)

    def node_params(self):
        """sigma / alpha update on the tree (src/var_inf.py:265-297)."""
        check(lib.bnpvi_node_params(self.ctx, self.tree.handle, _ptr(self.stats),
                                    int(self.stats.shape[1]), self.D, self.sB, self.sZ,
                                    _ptr(self.alpha0), _ptr(self.alpha), _ptr(self.sigma),
                                    self._stream()), "bnpvi_node_params")

    def gamma_update(self):
        """src/var_inf.py:395-416 from the per-video leaf masses."""
        K = self.tree.K
        if self.gamma is None or tuple(self.gamma.shape) != (self.V, K, 2):
            self.gamma = self._empty((self.V, K, 2), torch.float64)
        if self.V == 0:
            return
        check(lib.bnpvi_gamma_update(self.ctx, self.tree.handle, _ptr(self.mass), self.V,
                                     self.GAMMA, _ptr(self.gamma), self._stream()),
              "bnpvi_gamma_update")

    def prior_table(self, want_tables=False):
        """src/var_inf.py:359-381 (+ the 1/sigma term of :315-317 folded in)."""
        Lp, L = self.tree.Lp, self.tree.L
        if self.prior is None or tuple(self.prior.shape) != (self.V, Lp):
            self.prior = self._empty((self.V, Lp), torch.float32)
        on = before = None
        if want_tables:
            on = self._empty((self.V, L), torch.float64)
            before = self._empty((self.V, L), torch.float64)
        if self.V == 0:
            return on, before
        check(lib.bnpvi_prior_table(self.ctx, self.tree.handle, _ptr(self.gamma), _ptr(self.sigma),
                                    self.V, self.sZ, _ptr(self.prior), _ptr(on), _ptr(before),
                                    self._stream()), "bnpvi_prior_table")
        return on, before

    def gamma_prior(self):
        """gamma-step and prior table back to back in one launch (src/var_inf.py:136-137)."""
        K, Lp = self.tree.K, self.tree.Lp
        if self.gamma is None or tuple(self.gamma.shape) != (self.V, K, 2):
            self.gamma = self._empty((self.V, K, 2), torch.float64)
        if self.prior is None or tuple(self.prior.shape) != (self.V, Lp):
            self.prior = self._empty((self.V, Lp), torch.float32)
        if self.V == 0:
            return
        check(lib.bnpvi_gamma_prior(self.ctx, self.tree.handle, _ptr(self.mass), _ptr(self.sigma), self.V,
                                    self.GAMMA, self.sZ, _ptr(self.gamma), _ptr(self.prior),
                                    self._stream()), "bnpvi_gamma_prior")

    # leaves beyond which the E-step statistics carry S0' only and S2 is computed for the gated leaves alone: the
    # reference's gate mean(phi) >= 1e-2 (src/var_inf.py:217) admits at most 100 leaves
    S0_ONLY_FROM = 100

    def estep(self, mode=_lib.ESTEP_SOFT, want_logphi=False, want_assign=False, want_stats=True,
              operands_ready=False, s0_only=None):
        """src/var_inf.py:313-357, 383-393; HARD = one-hot nearest leaf (:126-132).
        s0_only (default: trees of more than 100 leaves): estats[:, 1] is left to `gated_s2`."""
        Lp, L = self.tree.Lp, self.tree.L
        if self.phi is None or tuple(self.phi.shape) != (self.N, Lp):
            self.phi = self._empty((self.N, Lp), torch.float32)
        logphi = self._empty((self.N, Lp), torch.float32) if want_logphi else None
        assign = self._empty((self.N,), torch.int32) if want_assign else None
        if want_stats and (self.estats is None or tuple(self.estats.shape) != (L, 2)):
            self.estats = self._empty((L, 2), torch.float64)
        if s0_only is None:
            s0_only = L > self.S0_ONLY_FROM
        s0_only = bool(s0_only and want_stats and mode == _lib.ESTEP_SOFT)
        if want_stats:
            self.s2_gated = s0_only
        if self.N == 0:
            if want_stats:
                self.estats.zero_()
            return logphi, assign
        check(lib.bnpvi_estep(self.ctx, self.tree.handle, _ptr(self.Z), self.Dp, _ptr(self.vid),
                              self.N, self.D, _ptr(self.alpha),
                              _ptr(self.prior if mode == _lib.ESTEP_SOFT else None), self.sZ,
                              int(mode) | (_lib.ESTEP_OPERANDS_READY if operands_ready else 0) |
                              (_lib.ESTEP_STATS_S0 if s0_only else 0),
                              _ptr(self.phi), _ptr(logphi), _ptr(assign),
                              _ptr(self.estats if want_stats else None), self._stream()),
              "bnpvi_estep")
        return logphi, assign

    def path_distances(self, assign, max_depth=None):
        """||alpha_k - z_n|| (fp64) for the nodes k on the root -> leaf path of every frame's
        assigned leaf (src/create_hierarchy.py:50): (node [N, max_depth] int32, -1 below the
        leaf; dist [N, max_depth] float64)."""
        if max_depth is None:
            max_depth = int(self.tree.depth.max()) + 1
        node = self._empty((self.N, max_depth), torch.int32)
        dist = self._empty((self.N, max_depth), torch.float64)
        check(lib.bnpvi_path_distances(self.ctx, self.tree.handle, _ptr(self.Z), self.Dp, self.N, self.D,
                                       _ptr(self.alpha), _ptr(assign), int(max_depth), _ptr(node), _ptr(dist),
                                       self._stream()), "bnpvi_path_distances")
        return node, dist

    def allreduce_estats(self):
        vdist.allreduce_sum_(self.estats, self.group)

    def gated_s2(self, n_total, gate=1e-2):
        """S2 = sum phi * ||z - alpha||^2 for the leaves whose GLOBAL mass passes the split gate
        mean(phi) >= gate (src/var_inf.py:217-220); call after `allreduce_estats`.  Everything stays on the
        device: gate_cols [128] (leaf indices), gate_count [2] (listed, passing), gate_s2 [128]."""
        cap = int(lib.bnpvi_split_max_columns())
        if self.gate_cols is None:
            self.gate_cols = torch.zeros((cap,), dtype=torch.int32, device=self.device)
            self.gate_count = torch.zeros((2,), dtype=torch.int32, device=self.device)
            self.gate_s2 = torch.zeros((cap,), dtype=torch.float64, device=self.device)
        check(lib.bnpvi_split_gate(self.ctx, _ptr(self.estats), self.tree.L, float(gate) * float(n_total),
                                   _ptr(self.gate_cols), cap, _ptr(self.gate_count), self._stream()),
              "bnpvi_split_gate")
        check(lib.bnpvi_leaf_s2(self.ctx, self.tree.handle, _ptr(self.Z), self.Dp, self.N, self.D, _ptr(self.alpha),
                                _ptr(self.phi), self.tree.Lp, _ptr(self.gate_cols), _ptr(self.gate_count), cap,
                                _ptr(self.gate_s2), self._stream()), "bnpvi_leaf_s2")
        vdist.allreduce_sum_(self.gate_s2, self.group)

    def params_prior(self):
        """node parameters + gamma-step + prior table (+ leaf operands) in one call."""
        K, Lp = self.tree.K, self.tree.Lp
        if self.gamma is None or tuple(self.gamma.shape) != (self.V, K, 2):
            self.gamma = self._empty((self.V, K, 2), torch.float64)
        if self.prior is None or tuple(self.prior.shape) != (self.V, Lp):
            self.prior = self._empty((self.V, Lp), torch.float32)
        if self.V == 0:                           # no videos here: node parameters only
            self.node_params()
            return
        check(lib.bnpvi_params_prior(self.ctx, self.tree.handle, _ptr(self.stats), int(self.stats.shape[1]),
                                     self.D, self.sB, self.sZ, _ptr(self.alpha0), _ptr(self.alpha),
                                     _ptr(self.sigma), _ptr(self.mass), self.V, self.GAMMA, _ptr(self.gamma),
                                     _ptr(self.prior), self._stream()), "bnpvi_params_prior")

    def iteration(self, want_estats=True, n_total=None):
        """One VI iteration: M-step -> gamma-step -> prior table -> E-step (var_inf.py:135-137):
        statistics pass, the single all-reduce, one mid-step call, E-step with the split / merge statistics
        (for trees of more than 100 leaves: S0' from the E-step, S2 of the gated leaves from `gated_s2`;
        n_total = frames over all ranks, default this rank's)."""
        self.leaf_stats(want_q=False, want_mass=True)
        self.allreduce_stats()
        self.params_prior()
        self.estep(want_stats=want_estats, operands_ready=True)
        if want_estats:
            self.allreduce_estats()                  # what split_nodes / merge_nodes read (no-op on one rank)
            if self.s2_gated:
                self.gated_s2(self.N if n_total is None else n_total)

    # ------------------------------------------------------------------ split / merge support
    def kmeans_seed(self, col, num_centres, responsibilities=True):
        """Weighted farthest-point seeding on phi column `col` (src/var_inf.py:23-36).
        Returns (chosen rows -- global indices under a process group --, centres (num_centres, D) fp64)."""
        rows, centres = self.kmeans_seed_multi([col], num_centres, responsibilities)
        return rows[0], centres[0]

    def kmeans_seed_multi(self, cols, num_centres, responsibilities=True):
        """`kmeans_seed` for several phi columns at once -- the leaves that split in one split_nodes walk
        (src/var_inf.py:221-224 seeds them one after the other; the seedings are independent).  One pass over
        Z per centre index for all of them; the arg-max rows feed the next pass's centres through a device
        gather, the host waits once at the end (single process) or exchanges one small table per pass.

        First centres = arg-max of the column.  Frames whose fp64 phi differ by less than fp32 resolution tie in
        the fp32 matrix (all those at exactly 1.0 on well-separated data) while the reference's fp64
        phi = 1 / (1 + c) still orders them; among the frames the leaf dominates the kernel takes the arg-max
        over that fp64 form (`responsibilities=False`: the columns are plain weights, or rows that do not sum
        to 1 -- fp32 arg-max only).

        Returns (rows [J][num_centres] -- global indices under a process group --, centres
        [J][num_centres][D] fp64)."""
        J = len(cols)
        D, Dp, N = self.D, self.Dp, self.N
        jmax = int(lib.bnpvi_kmeans_max_columns(D))
        if J > jmax:
            rows, cen = [], []
            for lo in range(0, J, jmax):
                r, c = self.kmeans_seed_multi(cols[lo:lo + jmax], num_centres, responsibilities)
                rows.extend(r)
                cen.extend(c)
            return rows, np.asarray(cen)
        multi = vdist.is_distributed(self.group)
        cols_t = torch.as_tensor(np.asarray(cols, dtype=np.int32), device=self.device)
        best_row = torch.zeros((J,), dtype=torch.int64, device=self.device)
        best_score = torch.full((J,), -1.0, dtype=torch.float64, device=self.device)
        dist2 = self._empty((max(N, 1), J), torch.float32) if num_centres > 1 else None
        centres = torch.zeros((J, Dp), dtype=torch.float32, device=self.device)
        picked = []                                                  # per pass: [J, Dp] fp32 rows of Z
        rows = []                                                    # per pass: [J] (global) row indices
        for k in range(num_centres):
            nested = False
            if N > 0 and k == 0 and responsibilities:
                # first centres with the reference's nested fp64 row sums (trees of up to 3072 nodes)
                rc = lib.bnpvi_kmeans_first_nested(self.ctx, self.tree.handle, _ptr(self.phi), int(self.phi.shape[1]),
                                                   N, _ptr(cols_t), J, _ptr(best_row), _ptr(best_score),
                                                   self._stream())
                if rc == 0:
                    nested = True
                elif rc != _lib.EUNSUPPORTED:
                    check(rc, "bnpvi_kmeans_first_nested")
            if N > 0 and not nested:
                check(lib.bnpvi_kmeans_pass_multi(self.ctx, _ptr(self.Z), Dp, N, D, _ptr(self.phi),
                                                  int(self.phi.shape[1]), int(self.tree.L) if responsibilities else 0,
                                                  _ptr(cols_t), J,
                                                  _ptr(centres if k > 0 else None), 1 if k <= 1 else 0,
                                                  _ptr(dist2), _ptr(best_row), _ptr(best_score), self._stream()),
                      "bnpvi_kmeans_pass_multi")
            if not multi:
                centres = self.Z.index_select(0, best_row)          # device gather: no host round trip
                rows.append(best_row.clone())
            else:
                # arg-max over ranks per column: (score, global row), lowest row wins ties (np.argmax).  With
                # all-zero scores every rank reports its local row 0, so global row 0 wins (the uniform
                # fallback of normalize()); empty ranks (score -1) lose.  The owners' rows travel in one
                # all-reduce of a [J, Dp] table that is zero everywhere else.
                grow, mine = vdist.select_argmax_many(best_score, best_row + self.row_offset, self.group)
                centres = torch.zeros((J, Dp), dtype=torch.float32, device=self.device)
                if N > 0 and bool(mine.any()):
                    idx = torch.nonzero(mine).flatten()
                    centres[idx] = self.Z.index_select(0, grow[idx] - self.row_offset)
                vdist.allreduce_sum_(centres, self.group)
                rows.append(grow)
            picked.append(centres)
        rows_h = torch.stack(rows, dim=1).cpu().numpy()              # [J, num_centres]; the one wait
        cen_h = torch.stack(picked, dim=1)[:, :, :D].to(torch.float64).cpu().numpy()
        return [[int(r) for r in rr] for rr in rows_h], cen_h

    def remap_phi(self, new_L, src, scale):
        """phi_new[:, j] = scale[j] * phi_old[:, src[j]] (split / merge column surgery)."""
        new_Lp = _lib.padded_leaves(new_L)
        out = self._empty((self.N, new_Lp), torch.float32)
        src = np.ascontiguousarray(src, dtype=np.int32)
        scale = np.ascontiguousarray(scale, dtype=np.float32)
        check(lib.bnpvi_remap_columns(self.ctx, _ptr(self.phi), int(self.phi.shape[1]), _ptr(out),
                                      new_Lp, self.N, int(new_L),
                                      src.ctypes.data_as(C.c_void_p),
                                      scale.ctypes.data_as(C.c_void_p), self._stream()),
              "bnpvi_remap_columns")
        self.phi = out

    def remap_gamma(self, new_K, src):
        """gamma_new[:, k] = gamma_old[:, src[k]] or the (1, GAMMA) default for src[k] < 0."""
        src = np.ascontiguousarray(src, dtype=np.int32)
        out = self._empty((self.V, new_K, 2), torch.float64)
        check(lib.bnpvi_gamma_remap(self.ctx, _ptr(self.gamma), int(self.gamma.shape[1]), _ptr(out),
                                    int(new_K), self.V, src.ctypes.data_as(C.c_void_p), self.GAMMA,
                                    self._stream()), "bnpvi_gamma_remap")
        self.gamma = out
