"""CPU restatement (TEST INFRASTRUCTURE ONLY, never imported by the product) of the frame
ranking in /root/reference/src/create_hierarchy.py: nearest-leaf assignment (:82-84), frames of
a leaf (:87-90), frames of an internal node = concatenation over its children (:44-49), ranking
by ||alpha - z|| with a stable sort and truncation to K (:50-51).  Leaves are numbered
depth-first where the reference uses the Python-2 order of a dict keyed by Node objects (address
dependent, so unpinned).  Parity: pinned by a hand-computed case in tests/test_oracle_golden.py
only -- the reference tool itself needs image files and is not run.
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
