"""Consumer side of the VI step: nearest-leaf assignment of frames and the
per-node ranking of the frames below a node -- what /root/reference/src/create_hierarchy.py
(:74-91, :40-53) and src/compute_acc_frame_avg_dynamic.py (:49-52) compute from the pickled node
list with a Python loop per frame.  Here: one hard-mode E-step (arg-min over leaves), one
path-distance kernel, and a stable device sort; `class_vote` is the tree-decayed class vote of
src/compute_acc_frame_avg_dynamic.py:61-129.  The image montage of create_hierarchy.py is not reproduced.
"""
from __future__ import annotations

import numpy as np


def _flatten(nodes_list):
    """Pre-order arrays of the hierarchy that contains `nodes_list` (root = the node without a
    parent, src/create_hierarchy.py:69-73)."""
    root = None
    for n in nodes_list:
        if n.parent is None:
            root = n
            break
    if root is None:
        raise ValueError("no root node (parent is None) in the list")
    order, stack = [], [root]
    while stack:
        n = stack.pop()
        order.append(n)
        if not n.isLeafNode:
            stack.extend(reversed(n.children))
    index = {id(n): i for i, n in enumerate(order)}
    parent = np.asarray([-1 if n.parent is None else index[id(n.parent)] for n in order], dtype=np.int32)
    is_leaf = np.asarray([1 if n.isLeafNode else 0 for n in order], dtype=np.uint8)
    alpha = np.asarray([np.asarray(n.alpha, dtype=np.float64) for n in order])
    return order, parent, is_leaf, alpha


def closest_frames(nodes_list, z, K, device=None):
    """For every node of the hierarchy: the (up to) K frames closest to its alpha among the frames
    whose nearest leaf lies below it (src/create_hierarchy.py:40-53, 82-91).

    z: (N, D) latent codes (NumPy or torch).  Returns (path_assignments, node_to_frames):
      path_assignments  (N,) int  index of the nearest leaf, leaves numbered depth-first
                                  (:82-84; the reference numbers them in the Python-2 order of a
                                  dict keyed by Node objects, which is address dependent)
      node_to_frames    {node: [row indices, closest first]}; ties keep the order of the
                                  reference's candidate list (leaves depth-first, rows ascending),
                                  as its stable `sorted` does (:51).
    """
    import torch
    from . import _lib
    from .engine import DeviceVI
    order, parent, is_leaf, alpha = _flatten(nodes_list)
    dev = DeviceVI(device=device)
    try:
        N = int(z.shape[0])
        dev.set_latents(z, np.zeros(N, dtype=np.int32), 1)
        dev.set_tree(parent, is_leaf, alpha, np.ones(len(order)))
        _, assign = dev.estep(mode=_lib.ESTEP_HARD, want_assign=True, want_stats=False)
        node, dist = dev.path_distances(assign)
        depth = node.shape[1]
        # candidates in the reference's order: leaves depth-first, rows ascending within a leaf
        rows = torch.sort(assign.to(torch.int64), stable=True).indices
        cand_node = node[rows].reshape(-1).to(torch.int64)
        cand_dist = dist[rows].reshape(-1)
        cand_row = rows.repeat_interleave(depth)
        keep = cand_node >= 0
        cand_node, cand_dist, cand_row = cand_node[keep], cand_dist[keep], cand_row[keep]
        by_dist = torch.sort(cand_dist, stable=True).indices
        by_node = torch.sort(cand_node[by_dist], stable=True)
        perm = by_dist[by_node.indices]
        sorted_node, sorted_row = by_node.values, cand_row[perm]
        counts = torch.bincount(sorted_node, minlength=len(order))
        starts = torch.cumsum(counts, 0) - counts
        rank = torch.arange(sorted_node.shape[0], device=sorted_node.device) - starts[sorted_node]
        top = rank < int(K)
        top_node = sorted_node[top].cpu().numpy()
        top_row = sorted_row[top].cpu().numpy()
        bounds = np.searchsorted(top_node, np.arange(len(order) + 1))
        result = {n: [int(r) for r in top_row[bounds[k]:bounds[k + 1]]] for k, n in enumerate(order)}
        return assign.cpu().numpy().astype(np.int64), result
    finally:
        dev.close()


def nearest_leaf(nodes_list, z, device=None):
    """(N,) index of the nearest leaf (leaves depth-first) of every row of z:
    src/compute_acc_frame_avg_dynamic.py:49-52 / :127-128 on the device (hard-mode E-step)."""
    from . import _lib
    from .engine import DeviceVI
    order, parent, is_leaf, alpha = _flatten(nodes_list)
    dev = DeviceVI(device=device)
    try:
        N = int(z.shape[0])
        dev.set_latents(z, np.zeros(N, dtype=np.int32), 1)
        dev.set_tree(parent, is_leaf, alpha, np.ones(len(order)))
        _, assign = dev.estep(mode=_lib.ESTEP_HARD, want_assign=True, want_stats=False)
        return assign.cpu().numpy().astype(np.int64)
    finally:
        dev.close()


def class_vote(nodes_list, z_train, train_labels, z_test, decay_factor=0.0, device=None):
    """Tree-decayed class vote of src/compute_acc_frame_avg_dynamic.py:49-129.

    Frames vote for the class of their nearest leaf (:61-63); votes are summed up the tree
    (:66-82); a leaf's score is its own votes plus decay_factor^j times the votes of its j-th
    ancestor (:89-108); leaves without any score borrow the class of the nearest leaf (by alpha)
    that has one (:114-122); test frames get the class of their nearest leaf (:124-129).
    train_labels: (N,) integer classes 0..C-1.  Returns (path2class [L], predicted [N_test]);
    leaves are numbered depth-first.  The O(N) parts (both nearest-leaf passes, the vote
    histogram) run on the device, the O(L*C) tree arithmetic on the host."""
    import torch
    order, parent, is_leaf, alpha = _flatten(nodes_list)
    leaves = [k for k in range(len(order)) if is_leaf[k]]
    L = len(leaves)
    labels = np.asarray(train_labels).astype(np.int64)
    C = len(np.unique(labels))                                          # :47
    assign = nearest_leaf(nodes_list, z_train, device)
    flat = torch.from_numpy(assign * C + labels)
    counts = torch.bincount(flat, minlength=L * C).numpy().reshape(L, C).astype(np.float64)
    cumulative = np.zeros((len(order), C))
    for j, k in enumerate(leaves):
        cumulative[k] = counts[j]
    for k in range(len(order) - 1, 0, -1):                              # pre-order: children after parents
        cumulative[parent[k]] += cumulative[k]
    total = np.zeros((L, C))
    for j, k in enumerate(leaves):
        mult, a = 1.0, k
        while a >= 0:
            total[j] += mult * cumulative[a]
            a = parent[a]
            mult *= decay_factor
    path2class = np.argmax(total, axis=1)
    leaf_alpha = alpha[leaves]
    for j in range(L):
        if np.max(total[j]) == 0:
            dist = np.linalg.norm(leaf_alpha[j] - leaf_alpha, axis=1)
            m = int(np.argmin(dist))
            while np.max(total[m]) == 0:
                dist[m] = np.inf
                m = int(np.argmin(dist))
            path2class[j] = path2class[m]
    predicted = path2class[nearest_leaf(nodes_list, z_test, device)]
    return path2class, predicted
