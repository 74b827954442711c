"""CPU restatement (TEST INFRASTRUCTURE ONLY, never imported by the product) of the frame
ranking in /root/reference/src/create_hierarchy.py: nearest-leaf assignment (:82-84), frames of
a leaf (:87-90), frames of an internal node = concatenation over its children (:44-49), ranking
by ||alpha - z|| with a stable sort and truncation to K (:50-51).  Leaves are numbered
depth-first where the reference uses the Python-2 order of a dict keyed by Node objects (address
dependent; under Python 3 it is insertion = depth-first order).  Parity: PINNED -- both reference tools
(create_hierarchy.py and compute_acc_frame_avg_dynamic.py) were run unmodified in the build
container on files in their own input formats by oracle/make_golden.py (offline_tools), and
tests/test_oracle_golden.py replays this restatement against tests/golden/offline_tools.npz
(assignments, vote table, path->class map, predictions, precision, per-node closest frames).
"""
import numpy as np


def dfs_leaves(root):
    out, stack = [], [root]
    while stack:
        n = stack.pop()
        if n.isLeafNode:
            out.append(n)
        else:
            stack.extend(reversed(n.children))
    return out


def closest_frames(root, z, K):
    leaves = dfs_leaves(root)
    alpha_values = np.asarray([l.alpha for l in leaves])
    path_assignments = []
    for zi in z:                                                     # :82-84
        path_assignments.append(int(np.argmin(np.linalg.norm(alpha_values - zi, axis=1))))
    path_assignments = np.asarray(path_assignments)
    result = {}
    for path_id, leaf in enumerate(leaves):                          # :87-90
        result[leaf] = list(np.where(path_assignments == path_id)[0])
    top = {}

    def rec(node):                                                   # :40-59
        if node.isLeafNode:
            frames_assigned = result[node]
        else:
            for c in node.children:
                rec(c)
            frames_assigned = []
            for c in node.children:
                frames_assigned += list(result[c])
            result[node] = frames_assigned
        if len(frames_assigned):
            dist_to_path = np.linalg.norm(node.alpha - z[frames_assigned], axis=1)
        else:
            dist_to_path = np.zeros(0)
        top_k = sorted(range(len(dist_to_path)), key=lambda i: dist_to_path[i])[:K]
        top[node] = [int(frames_assigned[i]) for i in top_k]

    rec(root)
    return path_assignments, top


def get_internal_nodes(node, result):                                # src/create_hierarchy.py:21-28, same in
    if not node.isLeafNode:                                           # compute_acc_frame_avg_dynamic.py
        result.append(node)
        for c in node.children:
            result = get_internal_nodes(c, result)
    return result


def class_vote(root, z_train, train_labels, z_test, decay_factor=0.0):
    """src/compute_acc_frame_avg_dynamic.py:41-129 with the reference's index bookkeeping
    (internal nodes first, then leaves; parents found by node_id lookups)."""
    leaf_nodes_list = dfs_leaves(root)
    alpha_values = np.asarray([l.alpha for l in leaf_nodes_list])
    num_paths = len(leaf_nodes_list)
    train_labels = np.asarray(train_labels)
    num_classes = len(np.unique(train_labels))
    path_assignments = [int(np.argmin(np.linalg.norm(alpha_values - z, axis=1))) for z in z_train]
    counts = np.zeros(shape=(num_paths, num_classes))
    for i, l in enumerate(path_assignments):
        counts[l][int(train_labels[i])] += 1
    internal_nodes_list = get_internal_nodes(root, [])
    num_internal_nodes = len(internal_nodes_list)
    counts_cumulative = np.concatenate((np.zeros(shape=(num_internal_nodes, num_classes)), counts.copy()), axis=0)
    leaf_ids = [x.node_id for x in leaf_nodes_list]
    internal_ids = [x.node_id for x in internal_nodes_list]
    for i, node in reversed(list(enumerate(internal_nodes_list))):
        for c in node.children:
            idx = num_internal_nodes + leaf_ids.index(c.node_id) if c.isLeafNode else internal_ids.index(c.node_id)
            counts_cumulative[i] += counts_cumulative[idx]
    counts_total = np.zeros(shape=(num_paths, num_classes))
    for i in range(num_paths):
        j = i + num_internal_nodes
        mult = 1.0
        while j >= 0:
            counts_total[i] += mult * counts_cumulative[j, :]
            node = leaf_nodes_list[j - num_internal_nodes] if j >= num_internal_nodes else internal_nodes_list[j]
            j = internal_ids.index(node.parent.node_id) if node.parent is not None else -1
            mult *= decay_factor
    path2class = np.argmax(counts_total, axis=1)
    for i in range(len(path2class)):
        if np.max(counts_total[i]) == 0:
            dist = np.linalg.norm(alpha_values[i] - alpha_values, axis=1)
            min_idx = np.argmin(dist)
            while np.max(counts_total[min_idx]) == 0:
                dist[min_idx] = np.inf
                min_idx = np.argmin(dist)
            path2class[i] = path2class[min_idx]
    predicted = [path2class[int(np.argmin(np.linalg.norm(alpha_values - z, axis=1)))] for z in z_test]
    class_vote.last = dict(path_assignments=np.asarray(path_assignments), counts_total=counts_total,
                           leaf_ids=leaf_ids)
    return path2class, np.asarray(predicted)


def precision_percent(predicted, test_labels):
    """The number the tool prints last (src/compute_acc_frame_avg_dynamic.py:131-146): correct
    predictions over all frames whose predicted class is one of the 15 MED classes, in percent."""
    predicted, test_labels = np.asarray(predicted), np.asarray(test_labels)
    corr = (predicted == test_labels).astype(int)
    aggr_corr = aggr_total = 0.0
    for class_id in range(15):
        idx = np.where(predicted == class_id)[0]
        aggr_corr += np.sum(corr[idx])
        aggr_total += len(idx)
    return 100.0 * aggr_corr / aggr_total
