"""Drop-in `VarInf` with the call surface of /root/reference/src/var_inf.py:12-491.

Host side (this file): the Node object graph, its flattening to pre-order SoA arrays, the
split / merge bookkeeping with the reference's exact control flow (including the
mutate-while-iterating quirk of merge_nodes), and lazy host views of device state.
Device side: every O(N) pass runs in the hand-written gfx950 kernels of libbnpvi.so via
`engine.DeviceVI` (C ABI in include/bnpvi.h).  There is no CPU fallback.

Method-level mapping (reference line numbers):
  update_variational_parameters :84-147    compute_sigma_alpha :299-301 (265-297)
  compute_gamma :415-416 (395-413)         compute_phi :383-393 (303-381)
  split_nodes :213-239                     weighted_kmeans :23-36
  merge_nodes :191-211                     get_matrix_from_dict :72-82
  get_true_path_mean :50-70                write_alpha/sigma/nodes :439-478
"""
from __future__ import annotations

import pickle
from copy import deepcopy

import numpy as np

from . import config
from .node import Node, _DEFAULT


class VarInf(object):
    def __init__(self, latent_code_size=None, stdev_thr=None, num_splits=None, gamma_hyper=None,
                 sigma_b_sqrinv=None, sigma_z_sqrinv=None, alpha_hyper=None, branching=None,
                 levels=None, device=None, group=None, retain_internal_phi='auto', verbose=False,
                 path_prior=False):
        # reference attributes (src/var_inf.py:13-21)
        self.root = None
        self.decay_coeff = 0.0
        self.vidid_frameid_to_idx = []
        self.vidid_to_idx = []
        self.kmeans_models = {}
        self.kmeans_labels = {}
        self.kmeans_ss = {}
        # hyper-parameters (module globals in the reference, see config.py)
        self.LATENT_CODE_SIZE = int(config.resolve('LATENT_CODE_SIZE', latent_code_size))
        self.STDEV_THR = float(config.resolve('STDEV_THR', stdev_thr))
        self.num_splits = int(config.resolve('NUM_SPLITS', num_splits))
        self.GAMMA = float(config.resolve('GAMMA', gamma_hyper))
        self.SIGMA_B_sqrinv = float(config.resolve('SIGMA_B_sqrinv', sigma_b_sqrinv))
        self.SIGMA_Z_sqrinv = float(config.resolve('SIGMA_Z_sqrinv', sigma_z_sqrinv))
        self.ALPHA = config.resolve_alpha(self.LATENT_CODE_SIZE, alpha_hyper)
        self.BF = int(config.resolve('BRANCHING_FACTOR', branching))
        self.levels = int(config.resolve('NUM_LEVELS', levels))
        self.verbose = verbose
        self.path_prior = bool(path_prior)   # run the branch behind the `continue` at src/var_inf.py:57
        self.pickle_phi = True
        self.retain_internal_phi = retain_internal_phi     # 'auto' | 'always' | 'never'
        # device state
        self._device = device
        self._group = group
        self._dev = None
        self._sig = None            # structure signature of the tree flattened on the device
        self._nodes = []            # pre-order nodes as flattened
        self._leaves = []           # DFS leaves as flattened (column order of device phi)
        self._phi_gen = 0
        self._phi_ver = 0           # bumped whenever phi content may have changed
        self._gamma_gen = 0
        self._phi_cache = {}
        self._gamma_cache = None
        self._Z_host = None
        self._vidx = None
        self._vidx_key = None
        self._row_offset = 0
        self._N_global = None
        self._mass_key = None
        self._estats_key = None
        self._estats_host = None
        self._estats_alpha = None
        self._fpos = None
        self._fpos_key = None
        self._params_stale = False   # device alpha / sigma newer than the Nodes' host values
        self._host_params_ver = 0    # bumped whenever a Node's alpha / sigmasqr_inv is assigned
        self._dev_params_ver = -1    # value of _host_params_ver the device copy was uploaded at
        self._trusted = 0            # > 0 inside _update: the flattened tree is re-derived only when flagged
        self._struct_dirty = True
        self._dev_alpha_ver = 0      # bumped whenever the device alpha / sigma change
        self._mirror = None          # host copy of the (alpha, sigma) last exchanged with the device
        self._phi_synced = False     # device phi holds exactly the current leaves' columns, in order
        self._gamma_synced = False   # same for the device gamma rows
        self._phi_all_dev = -1       # generation at which every leaf was pointed at its device column
        self._gamma_all_dev = -1

    # ================================================================== helpers
    def _log(self, *a):
        if self.verbose:
            print(*a)

    def _engine(self):
        if self._dev is None:
            from .engine import DeviceVI           # imports libbnpvi (fails loudly if missing)
            self._dev = DeviceVI(device=self._device, group=self._group,
                                 sigma_b_sqrinv=self.SIGMA_B_sqrinv,
                                 sigma_z_sqrinv=self.SIGMA_Z_sqrinv, gamma_hyper=self.GAMMA,
                                 alpha_hyper=self.ALPHA if np.any(self.ALPHA != 0) else None)
        return self._dev

    def _multi(self):
        from . import dist as vdist
        return vdist.is_distributed(self._group)

    def _num_frames_global(self):
        n = len(self.vidid_frameid_to_idx)
        if not self._multi():
            return n
        if self._N_global is None or self._N_global[0] != n:
            from . import dist as vdist
            total, _off = vdist.global_count(n, self._group, self._engine().device)
            self._N_global = (n, total)
        return self._N_global[1]

    def _new_node(self, node_id, parent):
        n = Node(node_id, len(self.vidid_to_idx), len(self.vidid_frameid_to_idx), parent,
                 self.LATENT_CODE_SIZE, self.GAMMA)
        n._owner = self
        return n

    def _preorder(self, node=None):
        out, stack = [], [self.root if node is None else node]
        while stack:
            n = stack.pop()
            out.append(n)
            if not n.isLeafNode:
                stack.extend(reversed(n.children))
        return out

    def _dfs_leaves(self, node=None):
        return [n for n in self._preorder(node) if n.isLeafNode]

    def _video_index(self):
        key = (len(self.vidid_to_idx), len(self.vidid_frameid_to_idx), id(self.vidid_frameid_to_idx))
        if self._vidx is None or self._vidx_key != key:
            pos = {v: i for i, v in enumerate(self.vidid_to_idx)}
            self._vidx = np.asarray([pos[v] for v, _f in self.vidid_frameid_to_idx], dtype=np.int32)
            self._vidx_key = key
        return self._vidx

    @property
    def phi_vidid_list(self):
        """Frame indices per video (src/var_inf.py:120-124, built there with an O(V*N) filter)."""
        vidx = self._video_index()
        order = np.argsort(vidx, kind='stable')
        bounds = np.searchsorted(vidx[order], np.arange(len(self.vidid_to_idx) + 1))
        return [list(order[bounds[i]:bounds[i + 1]]) for i in range(len(self.vidid_to_idx))]

    # ------------------------------------------------------------------ device sync
    def _load_latents(self, Z):
        dev = self._engine()
        if Z is self._Z_host and dev.Z is not None:
            return dev
        if not hasattr(Z, 'is_cuda'):            # torch tensors (host or device) pass through
            Z = np.asarray(Z)
        if Z.ndim != 2 or Z.shape[0] != len(self.vidid_frameid_to_idx):
            raise ValueError("latent_codes_matrix has shape %r but %d frames are registered"
                             % (Z.shape, len(self.vidid_frameid_to_idx)))
        dev.set_latents(Z, self._video_index(), len(self.vidid_to_idx))
        if self._multi():
            from . import dist as vdist
            _total, self._row_offset = vdist.global_count(Z.shape[0], self._group, dev.device)
        dev.row_offset = self._row_offset
        self._Z_host = Z
        self._mass_key = None
        self._estats_key = None
        return dev

    def _ensure_tree(self):
        """Flatten the Node graph (pre-order) and rebuild the device tree if it changed.

        A caller may have rearranged the Node graph by hand between calls (tests, tools), so the
        public methods re-derive the flattening every time; inside one `_update` only split_nodes /
        merge_nodes change it and they say so (`_struct_dirty`), which saves ~8 walks over all K nodes
        per call.  Host alpha / sigma are uploaded only when a Node was assigned since the last upload."""
        dev = self._engine()
        if self._trusted and not self._struct_dirty and self._sig is not None:
            if self._dev_params_ver != self._host_params_ver:
                self._push_params()
            return dev
        nodes = self._preorder()
        sig = tuple((id(n), bool(n.isLeafNode)) for n in nodes)
        changed = sig != self._sig
        if changed:
            if self._params_stale:
                self._pull_params()              # the old flattening's device values, before it goes
            index = {id(n): i for i, n in enumerate(nodes)}
            alpha = np.asarray([np.asarray(n._alpha, dtype=np.float64) for n in nodes])
            sigma = np.asarray([float(n._sigmasqr_inv) for n in nodes], dtype=np.float64)
            parent = np.asarray([-1 if n.parent is None else index[id(n.parent)] for n in nodes],
                                dtype=np.int32)
            is_leaf = np.asarray([1 if n.isLeafNode else 0 for n in nodes], dtype=np.uint8)
            dev.set_tree(parent, is_leaf, alpha, sigma)
            self._mirror = (alpha, sigma)
            self._sig = sig
            self._nodes = nodes
            self._leaves = [n for n in nodes if n.isLeafNode]
            self._mass_key = None
            self._estats_key = None
            for n in nodes:
                n._owner = self
            self._dev_params_ver = self._host_params_ver
            self._dev_alpha_ver += 1
        elif not self._params_stale:
            # public methods called directly: a caller may also have edited a Node's alpha IN PLACE.  (While
            # the device values are newer than the Nodes' -- nobody has read a Node.alpha since the last
            # M-step, the getter would have downloaded them -- there is nothing a caller could have edited.)
            self._push_params()
        self._struct_dirty = False
        return dev

    def _push_params(self):
        """Host alpha / sigma of the flattened nodes -> device (a Node was assigned)."""
        import torch
        dev = self._dev
        if self._params_stale:
            self._pull_params()
        alpha = np.asarray([np.asarray(n._alpha, dtype=np.float64) for n in self._nodes])
        sigma = np.asarray([float(n._sigmasqr_inv) for n in self._nodes], dtype=np.float64)
        self._dev_params_ver = self._host_params_ver
        m = self._mirror
        if m is not None and m[0].shape == alpha.shape and np.array_equal(m[0], alpha) and np.array_equal(m[1], sigma):
            return                               # the device already holds exactly these values
        dev.alpha.copy_(torch.from_numpy(alpha))
        dev.sigma.copy_(torch.from_numpy(sigma))
        self._mirror = (alpha, sigma)
        self._dev_alpha_ver += 1                 # statistics taken against the old alpha are no longer current

    def _pull_params(self):
        """Device alpha / sigma -> the Nodes' host values (one download for all nodes)."""
        self._params_stale = False
        dev = self._dev
        alpha = dev.alpha.cpu().numpy()
        sigma = dev.sigma.cpu().numpy()
        self._mirror = (alpha, sigma)
        for k, n in enumerate(self._nodes):
            n._alpha = alpha[k].copy()
            n._sigmasqr_inv = float(sigma[k])

    def _retire_internal_phi(self):
        """Device phi is about to be replaced: internal nodes that still point into it either
        take a host copy (the reference keeps a stale (N,1) phi on split nodes, which only
        matters when such a node loses all its children -- possible only with num_splits == 1)
        or fall back to the Node default."""
        keep = self.retain_internal_phi == 'always' or \
            (self.retain_internal_phi == 'auto' and self.num_splits == 1)
        flat = self._trusted and not self._struct_dirty and self._sig is not None
        for n in (self._nodes if flat else self._preorder()):
            if not n.isLeafNode and n._phi_dev is not None:
                if keep and n._phi_host is None:
                    got = self._fetch_phi(n)
                    n._phi_host = got
                n._phi_dev = None

    def _sync_phi(self):
        """Make the device phi matrix hold exactly the current leaves, in DFS order."""
        import torch
        dev = self._ensure_tree()
        if self._trusted and self._phi_synced:
            return dev
        leaves = self._leaves
        L = len(leaves)
        src = np.full(L, -1, dtype=np.int32)
        scale = np.zeros(L, dtype=np.float32)
        trivial = dev.phi is not None and dev.phi.shape[0] == dev.N and dev.phi.shape[1] == dev.tree.Lp
        for j, leaf in enumerate(leaves):
            pd = leaf._phi_dev
            if leaf._phi_host is None and pd is not None and pd[0] == self._phi_gen and dev.phi is not None:
                src[j], scale[j] = pd[1], pd[2]
                trivial = trivial and pd[1] == j and pd[2] == 1.0
            else:
                trivial = False
        if trivial:
            self._phi_synced = True
            return dev
        host_cols = {j: leaf.phi for j, leaf in enumerate(leaves) if src[j] < 0}
        self._retire_internal_phi()
        if dev.phi is not None and (src >= 0).any():
            dev.remap_phi(L, np.where(src >= 0, src, 0), np.where(src >= 0, scale, 0.0))
        else:
            dev.alloc_phi()
        for j, col in host_cols.items():
            dev.phi[:, j].copy_(torch.from_numpy(np.ascontiguousarray(col[:, 0], dtype=np.float32)))
        self._phi_gen += 1
        self._phi_ver += 1
        self._phi_cache = {}
        for j, leaf in enumerate(leaves):
            leaf._phi_host = None
            leaf._phi_dev = (self._phi_gen, j, 1.0)
        self._phi_all_dev = self._phi_gen
        self._phi_synced = True
        return dev

    def _sync_gamma(self):
        """Make the device gamma table hold the current nodes' gamma, in pre-order."""
        import torch
        dev = self._ensure_tree()
        if self._trusted and self._gamma_synced:
            return dev
        nodes = self._nodes
        K = len(nodes)
        src = np.full(K, -1, dtype=np.int32)
        trivial = dev.gamma is not None and dev.gamma.shape[1] == K and dev.gamma.shape[0] == dev.V
        host_rows = {}
        for k, n in enumerate(nodes):
            gd = n._gamma_dev
            if n._gamma_host is _DEFAULT and gd is not None and gd[0] == self._gamma_gen \
                    and dev.gamma is not None:
                src[k] = gd[1]
                trivial = trivial and gd[1] == k
            else:
                trivial = False
                if n._gamma_host is not _DEFAULT and n._gamma_host is not None:
                    host_rows[k] = n._gamma_host
        if trivial:
            self._gamma_synced = True
            return dev
        if dev.gamma is not None and (src >= 0).any():
            dev.remap_gamma(K, src)
        else:
            dev.default_gamma()
        for k, g in host_rows.items():
            dev.gamma[:, k, :].copy_(torch.from_numpy(np.ascontiguousarray(g, dtype=np.float64)))
        self._gamma_gen += 1
        self._gamma_cache = None
        for k, n in enumerate(nodes):
            if n._gamma_host is not None:
                n._gamma_host = _DEFAULT
                n._gamma_dev = (self._gamma_gen, k)
        self._gamma_synced = True
        return dev

    # lazy host views --------------------------------------------------------------------------
    def _note_phi_changed(self):
        self._phi_synced = False
        self._phi_ver += 1
        self._mass_key = None
        self._estats_key = None

    def _fetch_phi(self, node):
        gen, col, scale = node._phi_dev
        if gen != self._phi_gen or self._dev is None or self._dev.phi is None:
            return None
        key = (col, scale)
        if key not in self._phi_cache:
            col_host = self._dev.phi[:, col].to('cpu').numpy().astype(np.float64)
            self._phi_cache[key] = (col_host * scale).reshape(-1, 1)
        return self._phi_cache[key]

    def _fetch_gamma(self, node):
        gen, k = node._gamma_dev
        if gen != self._gamma_gen or self._dev is None or self._dev.gamma is None:
            return None
        if self._gamma_cache is None:
            self._gamma_cache = self._dev.gamma.cpu().numpy()      # (V, K, 2)
        return self._gamma_cache[:, k, :]

    # ================================================================== marshalling
    def get_matrix_from_dict(self, latent_codes):               # src/var_inf.py:72-82
        if len(self.vidid_to_idx) == 0:
            for vidid in latent_codes:
                self.vidid_to_idx.append(vidid)
                for frameid in latent_codes[vidid]:
                    self.vidid_frameid_to_idx.append((vidid, frameid))
        rows = [z for vidid in latent_codes for z in latent_codes[vidid].values()]
        if not rows:                 # a rank that holds no video (multi-rank, fewer videos than ranks)
            return np.zeros((0, self.LATENT_CODE_SIZE), dtype=np.float32)
        return np.asarray(rows)

    # ================================================================== entry point
    def update_variational_parameters(self, latent_codes):      # src/var_inf.py:84-147
        self._log('Performing variational inference...')
        self._update(self.get_matrix_from_dict(latent_codes))

    def update_variational_parameters_matrix(self, Z, keys=None):
        """The same step for callers that hold the latents as ONE (N, D) matrix -- a NumPy array or
        a torch tensor, which may already live on the GPU (no host round trip) -- instead of the
        reference's dict of dicts, whose flattening (src/var_inf.py:72-82) costs more wall-clock
        than the whole device step from a few 10^5 frames on.  Rows must be in the order
        get_matrix_from_dict would produce (videos one after the other, frames of a video
        contiguous).  `keys` = [(vidid, frameid)] per row; required on the first call, which
        freezes the row order like :73-77, ignored afterwards."""
        if len(self.vidid_to_idx) == 0:
            if keys is None:
                raise ValueError("the first call needs keys = [(vidid, frameid)] per row")
            keys = list(keys)
            seen = set()
            last = object()
            for vidid, _frameid in keys:
                if vidid != last:
                    if vidid in seen:
                        raise ValueError("frames of video %r are not contiguous rows" % (vidid,))
                    seen.add(vidid)
                    self.vidid_to_idx.append(vidid)
                    last = vidid
            self.vidid_frameid_to_idx = keys
        if int(Z.shape[0]) != len(self.vidid_frameid_to_idx):
            raise ValueError("matrix has %d rows but %d frames are registered"
                             % (int(Z.shape[0]), len(self.vidid_frameid_to_idx)))
        self._Z_host = None      # the caller may have rewritten the same matrix in place: always reload
        self._update(Z)

    def _update(self, Z):
        if self.root is None:
            self._build_initial_tree()
            self._replicate_initial_alpha()
            self._one_hot_init(Z)
        elif self.root.isLeafNode:
            # The tree has collapsed to its root (merge_nodes pruned every child, src/var_inf.py:200-201).  The
            # reference cannot go on from there: compute_gamma_node sets the leaf root's gamma to None (:397-399) and
            # compute_gamma_sum_before then indexes it (:372-373, digamma_add1) -- TypeError.  Same error here,
            # raised before anything is launched.
            raise TypeError("'NoneType' object is not subscriptable (the tree has collapsed to its root: the "
                            "reference fails in compute_gamma_sum_before, src/var_inf.py:372-373)")
        self._load_latents(Z)
        self._ensure_tree()                  # full check once; inside, only flagged changes re-flatten
        self._trusted += 1
        try:
            self._update_body(Z)
        finally:
            self._trusted -= 1

    def _vi_iteration(self, Z):
        """compute_sigma_alpha + compute_gamma + compute_phi of one call, launched the way
        DeviceVI.iteration (the benchmarked step) launches them: one statistics pass, its all-reduce, ONE
        mid-step launch (node parameters, gamma-step, prior table, leaf operands) and the E-step with its
        split / merge statistics -- same results as the three public methods
        (tests/test_gpu_parity.py::test_fused_iteration_matches_separate_calls)."""
        dev = self._load_latents(Z)
        self._sync_phi()
        dev.leaf_stats(want_q=False, want_mass=True)
        dev.allreduce_stats()
        dev.params_prior()                   # consumes (and re-zeroes) the per-video masses
        self._mass_key = None
        self._params_stale = True            # Node.alpha / .sigmasqr_inv download on first access
        self._dev_alpha_ver += 1
        self._gamma_now_on_device()
        self._retire_internal_phi()
        dev.estep(want_stats=True, operands_ready=True)
        self._after_estep(dev)

    def _update_body(self, Z):
        for _iteration in range(1):
            self._vi_iteration(Z)
            self.print_phi(self.root)
        split = self.split_nodes(self.root, Z, self.STDEV_THR + 100.0 * np.exp(-1.0 * self.decay_coeff))
        if split:
            self.decay_coeff = 0.0
            self.compute_phi(Z)
        else:
            self.decay_coeff += 1.0
        self.merge_nodes(self.root, Z, self.STDEV_THR + 100.0 * np.exp(-1.0 * self.decay_coeff))

    def _build_initial_tree(self):
        """Root + BF-ary levels created level by level (src/var_inf.py:90-118: BF=4, 3 levels);
        node creation order fixes the order of the np.random draws of Node.__init__."""
        self.root = self._new_node('0', None)
        prev = [self.root]
        for _ in range(self.levels):
            cur = []
            for n in range(len(prev) * self.BF):
                par = prev[n // self.BF]
                cur.append(self._new_node(par.node_id + '-' + str(n % self.BF), par))
            for i, par in enumerate(prev):
                par.isLeafNode = False
                par.children = cur[self.BF * i: self.BF * (i + 1)]
            prev = cur

    def _replicate_initial_alpha(self):
        """Multi-rank: the tree is replicated, so every rank must start from rank 0's draws of
        Node.alpha (the process-local NumPy streams need not be seeded alike)."""
        if not self._multi():
            return
        import torch
        from . import dist as vdist
        nodes = self._preorder()
        dev = self._engine()
        a = torch.from_numpy(np.asarray([n.alpha for n in nodes], dtype=np.float64)).to(dev.device)
        vdist.broadcast_from(a, 0, self._group)
        a = a.cpu().numpy()
        for k, n in enumerate(nodes):
            n.alpha = a[k].copy()

    def _one_hot_init(self, Z):
        """phi <- one-hot of the nearest leaf (src/var_inf.py:126-132), on the device."""
        from . import _lib
        dev = self._load_latents(Z)
        self._ensure_tree()
        self._retire_internal_phi()
        dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
        self._adopt_device_phi()

    def _adopt_device_phi(self):
        """The device phi matrix was just rewritten by an E-step: every leaf's phi IS its column now."""
        self._phi_ver += 1
        self._phi_cache = {}
        self._mass_key = None
        if self._trusted and self._phi_synced and self._phi_all_dev == self._phi_gen:
            return                           # same leaves, same columns: nothing to re-point
        self._phi_gen += 1
        for j, leaf in enumerate(self._leaves):
            leaf._phi_host = None
            leaf._phi_dev = (self._phi_gen, j, 1.0)
        self._phi_all_dev = self._phi_gen
        self._phi_synced = True

    # ================================================================== M-step
    def compute_sigma_alpha(self, latent_codes_matrix):         # src/var_inf.py:299-301
        dev = self._load_latents(latent_codes_matrix)
        self._sync_phi()
        dev.leaf_stats(want_q=False, want_mass=True)
        self._mass_key = (self._phi_gen, self._phi_ver)
        dev.allreduce_stats()
        dev.node_params()
        self._params_stale = True            # Node.alpha / .sigmasqr_inv download on first access
        self._dev_alpha_ver += 1
        self._estats_key = None

    # ================================================================== gamma-step
    def compute_gamma(self):                                    # src/var_inf.py:415-416
        dev = self._engine()
        if dev.Z is None:
            raise RuntimeError("compute_gamma: no latent codes on the device yet")
        self._sync_phi()
        if self._mass_key != (self._phi_gen, self._phi_ver):
            dev.leaf_stats(want_q=False, want_mass=True)
            self._mass_key = (self._phi_gen, self._phi_ver)
        dev.gamma_update()
        self._gamma_now_on_device()

    def _gamma_now_on_device(self):
        """The device gamma table was just rewritten by a gamma-step: every node's gamma IS its row now."""
        self._gamma_cache = None
        if not (self._trusted and self._gamma_synced and self._gamma_all_dev == self._gamma_gen):
            self._gamma_gen += 1
            for k, n in enumerate(self._nodes):
                n._gamma_host = _DEFAULT
                n._gamma_dev = (self._gamma_gen, k)
            self._gamma_all_dev = self._gamma_gen
            self._gamma_synced = True
        if len(self._nodes) == 1:
            self.root._gamma_host = None                        # root-as-leaf (:397-399)
            self.root._gamma_dev = None

    # ================================================================== E-step
    def compute_phi(self, latent_codes_matrix, want_logphi=False, want_assign=False):
        """src/var_inf.py:383-393.  Returns (log_phi, assign) device tensors when asked."""
        dev = self._load_latents(latent_codes_matrix)
        self._ensure_tree()
        self._sync_gamma()
        dev.prior_table()
        self._retire_internal_phi()
        logphi, assign = dev.estep(want_logphi=want_logphi, want_assign=want_assign, want_stats=True)
        self._after_estep(dev)
        return logphi, assign

    def _after_estep(self, dev):
        """The split / merge statistics of the E-step that just ran: S0' (and S2) summed over the ranks; for
        trees of more than 100 leaves the kernel carried S0' only and S2 is computed now, on the device, for
        the leaves that pass the reference's mass gate (src/var_inf.py:217) -- the only ones whose stdev
        split_nodes ever looks at."""
        dev.allreduce_estats()
        if dev.s2_gated:
            dev.gated_s2(self._num_frames_global())
        self._adopt_device_phi()
        self._estats_key = (self._phi_gen, self._phi_ver)
        self._estats_host = None
        self._estats_alpha = self._dev_alpha_ver

    # ================================================================== split / merge
    def _leaf_moments(self):
        """Per current leaf: S0 = sum phi, S2 = sum phi*||z-alpha||^2 (global over ranks)."""
        import torch
        dev = self._ensure_tree()
        self._sync_phi()
        # the statistics of the last E-step are current while phi and the node parameters it used are
        # (no Node.alpha assigned since, no M-step since: both are tracked, no K x D compare per call)
        fresh = self._estats_key == (self._phi_gen, self._phi_ver) and \
            self._estats_alpha == self._dev_alpha_ver
        if fresh:
            if self._estats_host is None:
                est = dev.estats.cpu().numpy()
                if dev.s2_gated:
                    # S2 exists for the gated leaves only; NaN elsewhere (split_nodes never reads those)
                    cnt = dev.gate_count.cpu().numpy()
                    if int(cnt[1]) > int(cnt[0]):
                        raise RuntimeError("split gate: %d leaves pass, %d fit one call" % (cnt[1], cnt[0]))
                    cols = dev.gate_cols.cpu().numpy()[:int(cnt[0])]
                    s2 = np.full(est.shape[0], np.nan)
                    s2[cols] = dev.gate_s2.cpu().numpy()[:int(cnt[0])]
                    est = np.stack([est[:, 0], s2], axis=1)
                self._estats_host = est
            est = self._estats_host
            return est[:, 0], est[:, 1]
        # phi / alpha did not come from the last E-step: one stats pass with Q = R^T.Z^2
        dev.leaf_stats(want_q=True, want_mass=True)
        self._mass_key = (self._phi_gen, self._phi_ver)
        dev.allreduce_stats()
        st = dev.stats.cpu().numpy()
        D = self.LATENT_CODE_SIZE
        S0, S1, Q = st[:, 0], st[:, 1:1 + D], st[:, 1 + D:1 + 2 * D]
        A = np.asarray([np.asarray(l.alpha, dtype=np.float64) for l in self._leaves])
        S2 = np.sum(Q - 2.0 * A * S1 + A * A * S0[:, None], axis=1)
        return S0, np.maximum(S2, 0.0)

    def split_nodes(self, node, latent_codes_matrix, split_thr):   # src/var_inf.py:213-239
        dev = self._load_latents(latent_codes_matrix)
        S0, S2 = self._leaf_moments()
        col = {id(l): j for j, l in enumerate(self._leaves)}
        N = float(self._num_frames_global())

        todo = []                                # (leaf, phi column) of the leaves that split, in walk order

        def rec(n):
            self._log('split_nodes', n.node_id)
            if n.isLeafNode:
                j = col[id(n)]
                if S0[j] / N < 10 ** -2:
                    return False
                stdev = np.sqrt(np.sqrt(S2[j]) / S0[j])
                self._log(n.node_id, stdev, split_thr)
                if stdev > split_thr:
                    todo.append((n, j))
                    return True
                return None
            result = False
            for c in n.children:
                result = rec(c) or result
            return result

        out = rec(node)
        if todo:
            # the reference seeds each splitting leaf as the walk reaches it (:221-224); the seedings are
            # independent of each other and of the walk, so they run batched (one pass over Z per centre
            # index for all leaves) and the children are created afterwards, in the walk's order -- which
            # also keeps the order of Node.__init__'s np.random draws
            best_K = self.num_splits
            # the fp64 ordering of the first centre's candidates (frames that tie in the fp32 matrix) assumes
            # rows that sum to 1, i.e. a phi matrix straight from an E-step; after a prune without E-step, or on
            # caller-assigned phi, the plain fp32 arg-max is used (parity on exact fp32 ties is unpinned there)
            rows_sum_to_one = self._estats_key == (self._phi_gen, self._phi_ver)
            _rows, centres = dev.kmeans_seed_multi([j for _n, j in todo], best_K, responsibilities=rows_sum_to_one)
            for (n, _j), cen in zip(todo, centres):
                n.isLeafNode = False
                for k in range(best_K):
                    child = self._new_node(n.node_id + '-' + str(k), n)
                    child.alpha = cen[k, :].copy()
                    child.sigmasqr_inv = n.sigmasqr_inv
                    gen, c, sc = n._phi_dev
                    child._phi_dev = (gen, c, sc / best_K)       # phi_parent / best_K
                    n.children.append(child)
        if out:
            self._struct_dirty = True
            self._gamma_synced = False
            self._note_phi_changed()
        return out

    def merge_nodes(self, node, latent_codes_matrix, split_thr):   # src/var_inf.py:191-211
        self._load_latents(latent_codes_matrix)
        S0, _S2 = self._leaf_moments()
        col = {id(l): j for j, l in enumerate(self._leaves)}
        N = float(self._num_frames_global())
        changed = [False]

        def rec(n):
            self._log('merge_nodes', n.node_id)
            if n.isLeafNode:
                if S0[col[id(n)]] / N < 10 ** -3:
                    self._log('Removing node ', n.node_id)
                    # mutates the list its caller iterates: the next sibling is skipped
                    n.parent.children.remove(n)
                    changed[0] = True
                return
            for c in n.children:
                rec(c)
            if len(n.children) == 0:
                n.isLeafNode = True                       # keeps whatever phi it had
                changed[0] = True
            elif len(n.children) == 1:
                self._log('Merge 1 child')
                only = n.children[0]
                n.isLeafNode = only.isLeafNode
                n.alpha = deepcopy(only.alpha)
                n.sigmasqr_inv = only.sigmasqr_inv
                n._phi_host = None if only._phi_host is None else only._phi_host.copy()
                n._phi_dev = only._phi_dev
                gh = only._gamma_host
                n._gamma_host = gh.copy() if isinstance(gh, np.ndarray) else gh
                n._gamma_dev = only._gamma_dev
                n.children = list(only.children)          # the reference deep-copies them
                for c in n.children:
                    c.parent = n
                changed[0] = True

        rec(node)
        if changed[0]:
            self._struct_dirty = True
            self._gamma_synced = False
            self._note_phi_changed()

    @staticmethod
    def weighted_kmeans(data, weights, K):                       # src/var_inf.py:23-36
        """Deterministic phi-weighted farthest-point seeding, on the device."""
        import torch
        from .engine import DeviceVI
        data = np.asarray(data)
        N, D = data.shape
        dev = DeviceVI()
        try:
            dev.set_latents(data.astype(np.float32), np.zeros(N, dtype=np.int32), 1)
            dev.phi = torch.zeros((N, 32), dtype=torch.float32, device=dev.device)
            dev.phi[:, 0].copy_(torch.from_numpy(np.asarray(weights, dtype=np.float32)))
            _rows, centres = dev.kmeans_seed(0, K, responsibilities=False)
        finally:
            dev.close()
        return centres

    @staticmethod
    def normalize(p):                                            # src/var_inf.py:418-425
        p = np.asarray(p)
        s = np.sum(p)
        if s > 0:
            return p / s
        return VarInf.normalize(np.ones(shape=np.shape(p)))

    # ================================================================== accessors
    def _frame_index(self):
        """(vidid, frameid) -> row, the dict form of the reference's `list.index` lookup (:62)."""
        key = (len(self.vidid_frameid_to_idx), id(self.vidid_frameid_to_idx))
        if self._fpos is None or self._fpos_key != key:
            self._fpos = {vf: i for i, vf in reversed(list(enumerate(self.vidid_frameid_to_idx)))}
            self._fpos_key = key
        return self._fpos

    def _leaves_bfs(self):
        """Leaves in the visiting order of get_phi_leaves (src/var_inf.py:40-47)."""
        out, queue = [], [self.root]
        while queue:
            nxt = queue.pop(0)
            if nxt.isLeafNode:
                out.append(nxt)
            else:
                queue.extend(nxt.children)
        return out

    def get_true_path_mean(self, x_annot_batch):                 # src/var_inf.py:50-70
        """Prior mean of every frame of a minibatch.

        Default (`path_prior=False`): the reference's live behaviour -- the "standard VAE" zero
        mean, because of the `continue` at :57.  With `path_prior=True` the branch behind it runs
        (:58-68): per frame one leaf is drawn with probability phi[n, .] (renormalised, :65) from
        NumPy's global stream (one `np.random.choice` per frame, :66) and that leaf's alpha is the
        mean.  The responsibilities of the whole batch come from the device in one gather.  Leaves
        are enumerated breadth-first like get_phi_leaves; the reference then goes through a dict
        keyed by Node objects, whose Python-2 order is address dependent, so which draw selects
        which leaf is not pinned by the reference -- the distribution is."""
        B = len(x_annot_batch)
        zeros = np.zeros(shape=(B, self.LATENT_CODE_SIZE))
        if not self.path_prior or B == 0:
            return zeros
        if len(self.vidid_frameid_to_idx) == 0 or self.root is None:
            return zeros                                         # "no paths initiated" (:58-60)
        pos = self._frame_index()
        rows = np.asarray([pos[(v, f)] for (v, f) in x_annot_batch], dtype=np.int64)
        leaves = self._leaves_bfs()
        phi = self._phi_rows(rows, leaves)                       # (B, L) fp64
        self.last_path_indices = []
        out = []
        for b in range(B):
            values = VarInf.normalize(phi[b])
            pick = np.random.choice(len(values), p=values)
            self.last_path_indices.append(int(pick))
            out.append(np.asarray(leaves[pick].alpha))
        self._log(self.last_path_indices)
        return np.asarray(out)

    def _phi_rows(self, rows, leaves):
        """phi[rows][:, leaves] on the host; one device gather when every leaf's column is live."""
        import torch
        dev = self._dev
        cols, scale = [], []
        for leaf in leaves:
            pd = leaf._phi_dev
            if leaf._phi_host is None and pd is not None and dev is not None and dev.phi is not None \
                    and pd[0] == self._phi_gen:
                cols.append(pd[1])
                scale.append(pd[2])
            else:
                cols = None
                break
        if cols is None:
            return np.asarray([[leaf.phi[r][0] for leaf in leaves] for r in rows], dtype=np.float64)
        r = torch.from_numpy(rows).to(dev.device)
        c = torch.tensor(cols, dtype=torch.int64, device=dev.device)
        got = dev.phi.index_select(0, r).index_select(1, c).to('cpu').numpy().astype(np.float64)
        return got * np.asarray(scale, dtype=np.float64)[None, :]

    def get_phi_leaves(self, idx):                               # src/var_inf.py:38-48
        result = {}
        unvisited = [self.root]
        while unvisited:
            nxt = unvisited[0]
            unvisited = unvisited[1:]
            if nxt.isLeafNode:
                result[nxt] = nxt.phi[idx][0]
            else:
                unvisited = unvisited + nxt.children
        return result

    def initialize_phi(self, node, init_values):                 # src/var_inf.py:167-172
        if node.isLeafNode:
            node.phi = np.reshape(init_values[node.node_id], (len(self.vidid_frameid_to_idx), 1))
        else:
            for c in node.children:
                self.initialize_phi(c, init_values)

    def get_nodes_list(self, node, result):                      # src/var_inf.py:465-472
        result.append(node)
        if not node.isLeafNode:
            for c in node.children:
                result = self.get_nodes_list(c, result)
        return result

    def get_alpha_leaves_node(self, node, result):               # src/var_inf.py:480-488
        if node.isLeafNode:
            result.update({node.node_id: node.alpha})
            return result
        for c in node.children:
            result.update(self.get_alpha_leaves_node(c, result))
        return result

    def get_alpha_leaves(self):                                  # src/var_inf.py:490-491
        return self.get_alpha_leaves_node(self.root, {})

    def print_phi(self, node):                                   # src/var_inf.py:174-179
        if not self.verbose:
            return
        for leaf in self._dfs_leaves(node):
            print('phi', leaf.node_id, np.mean(leaf.phi))

    # ================================================================== writers
    def write_alpha_node(self, node, fileptr):                   # src/var_inf.py:439-445
        fileptr.write(node.node_id + ' ' + ' '.join(str(np.round(x, 2)) for x in node.alpha) + '\n')
        if not node.isLeafNode:
            for c in node.children:
                self.write_alpha_node(c, fileptr)

    def write_alpha(self, filename):                             # src/var_inf.py:447-450
        with open(filename, 'w') as f:
            self.write_alpha_node(self.root, f)

    def write_sigma_node(self, node, fileptr):                   # src/var_inf.py:452-458
        fileptr.write(node.node_id + ' ' + ' '.join(str(np.round(x, 2)) for x in [node.sigmasqr_inv]) + '\n')
        if not node.isLeafNode:
            for c in node.children:
                self.write_sigma_node(c, fileptr)

    def write_sigma(self, filename):                             # src/var_inf.py:460-463
        with open(filename, 'w') as f:
            self.write_sigma_node(self.root, f)

    def write_nodes(self, filename):                             # src/var_inf.py:474-478
        nodes_list = self.get_nodes_list(self.root, [])
        with open(filename, 'wb') as f:
            pickle.dump(nodes_list, f, protocol=2)
