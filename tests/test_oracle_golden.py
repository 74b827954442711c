"""The CPU oracle replayed against the committed golden vectors (outputs of the REAL
reference, see oracle/make_golden.py).  Runs everywhere, including the GPU box where
/root/reference does not exist.  fp64 throughout; tolerance 1e-12 (the oracle is bit-exact
against the reference with the NumPy/SciPy versions recorded in each fixture's `meta`).
"""
import os

import numpy as np
import pytest

from oracle.vi_oracle import (OracleVarInf, ONode, snapshot, restore, make_latent_dict)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-12


def load(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def assert_snap(got, want, tol=TOL, phi_tol=None):
    assert list(got['ids']) == list(want['ids'])
    np.testing.assert_array_equal(got['parent'], want['parent'])
    np.testing.assert_array_equal(got['is_leaf'], want['is_leaf'])
    np.testing.assert_allclose(got['alpha'], want['alpha'], rtol=tol, atol=tol)
    np.testing.assert_allclose(got['sigma'], want['sigma'], rtol=tol, atol=tol)
    np.testing.assert_allclose(got['gamma'], want['gamma'], rtol=tol, atol=tol, equal_nan=True)
    np.testing.assert_allclose(got['phi'], want['phi'], rtol=0, atol=tol if phi_tol is None else phi_tol)
    assert float(got['decay']) == float(want['decay'])


def meta(d):
    import json
    return json.loads(str(d['meta']))


def rows_for(fpv, V, N):
    lc = make_latent_dict(np.zeros((N, 1), np.float32), list(fpv))
    vids = list(lc.keys())
    pairs = [(v, f) for v in vids for f in lc[v].keys()]
    assert len(vids) == V
    return vids, pairs


@pytest.mark.parametrize("name", ["traj_ref_shape.npz", "traj_dynamic.npz", "traj_single_video.npz"])
def test_trajectory(name):
    g = load(name)
    m = meta(g)
    vi = OracleVarInf(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    for ep in range(m['epochs']):
        lc = make_latent_dict(g['Z%d' % ep], list(g['fpv']))
        np.random.seed(13 + ep)
        vi.update_variational_parameters(lc)
        assert_snap(snapshot(vi), sub(g, 'e%d_' % ep))


@pytest.mark.parametrize("name", ["phases_irregular.npz", "phases_ref_shape.npz"])
def test_phases(name):
    g = load(name)
    m = meta(g)
    fpv = list(g['fpv'])
    N, V = int(sum(fpv)), len(fpv)
    vids, pairs = rows_for(fpv, V, N)
    vi = OracleVarInf(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    fac = lambda nid, par: ONode(nid, V, N, par, m['D'], 10.0, rng=np.random.RandomState(0))
    restore(vi, sub(g, 'in_'), fac, vids, pairs)
    assert_snap(snapshot(vi), sub(g, 'in_'))
    Z = g['Z']
    vi.compute_sigma_alpha(Z)
    assert_snap(snapshot(vi), sub(g, 'm_'))
    vi.compute_gamma()
    assert_snap(snapshot(vi), sub(g, 'g_'))
    on, before = vi.prior_tables()
    leaves = vi.leaves()
    np.testing.assert_allclose(np.asarray([on[l] for l in leaves]), g['on'], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(np.asarray([before[l] for l in leaves]), g['before'], rtol=TOL, atol=TOL)
    np.testing.assert_allclose(vi.logits(Z), g['logits'], rtol=TOL, atol=TOL)
    vi.compute_phi(Z)
    assert_snap(snapshot(vi), sub(g, 'e_'))
    np.testing.assert_allclose(vi.last_log_phi, g['log_phi'], rtol=0, atol=1e-10)
    np.testing.assert_allclose([np.mean(l.phi) for l in leaves], g['leaf_mean_phi'], rtol=TOL)
    np.testing.assert_allclose([vi.leaf_stdev(l.phi, Z, l.alpha) for l in leaves], g['leaf_stdev'], rtol=TOL)
    thr = float(g['split_thr'])
    np.random.seed(m.get('seed', 0))      # children alphas are overwritten by the k-means centres
    ret = vi.split_nodes(vi.root, Z, thr)
    assert bool(ret) == bool(g['split_ret'])
    assert_snap(snapshot(vi), sub(g, 's_'))
    vi.compute_phi(Z)
    vi.merge_nodes(vi.root, Z, thr)
    assert_snap(snapshot(vi), sub(g, 'x_'))


def test_faithful_loop_form_is_identical():
    """The per-frame loop + list.index form (what the CPU baseline times) == vectorised form."""
    g = load("phases_irregular.npz")
    m = meta(g)
    fpv = list(g['fpv'])
    N, V = int(sum(fpv)), len(fpv)
    vids, pairs = rows_for(fpv, V, N)
    outs = []
    for faithful in (False, True):
        vi = OracleVarInf(latent_code_size=m['D'], faithful_loops=faithful)
        fac = lambda nid, par: ONode(nid, V, N, par, m['D'], 10.0, rng=np.random.RandomState(0))
        restore(vi, sub(g, 'g_'), fac, vids, pairs)
        outs.append(vi.logits(g['Z']))
    np.testing.assert_array_equal(outs[0], outs[1])


def test_weighted_kmeans_golden():
    g = load("kmeans.npz")
    for i in range(3):
        c = OracleVarInf.weighted_kmeans(g['data%d' % i], g['w%d' % i], int(g['K%d' % i]))
        np.testing.assert_allclose(c, g['centers%d' % i], rtol=0, atol=0)


def test_hierarchy_oracle_hand_case():
    """oracle/hierarchy_oracle.py (restating src/create_hierarchy.py:40-53,82-91) on a case small
    enough to do by hand: two leaves at (1,0) and (-1,0) under a root at the origin."""
    from oracle.hierarchy_oracle import closest_frames

    class N(object):
        def __init__(self, alpha, parent, leaf):
            self.alpha, self.parent, self.isLeafNode, self.children = np.asarray(alpha, float), parent, leaf, []
            if parent is not None:
                parent.children.append(self)

    root = N([0, 0], None, False)
    a, b = N([1, 0], root, True), N([-1, 0], root, True)
    z = np.asarray([[0.9, 0], [2, 0], [-0.5, 0], [-3, 0], [0.2, 0]])
    assign, top = closest_frames(root, z, 2)
    assert list(assign) == [0, 0, 1, 1, 0]
    assert top[a] == [0, 4] and top[b] == [2, 3] and top[root] == [4, 2]


def test_class_vote_oracle_hand_case():
    """oracle/hierarchy_oracle.class_vote (src/compute_acc_frame_avg_dynamic.py:41-129) by hand:
    leaf (5,0) gets no vote and borrows the class of its nearest voted neighbour (1,0)."""
    from oracle.hierarchy_oracle import class_vote

    class N(object):
        def __init__(self, node_id, alpha, parent, leaf):
            self.node_id, self.alpha, self.parent = node_id, np.asarray(alpha, float), parent
            self.isLeafNode, self.children = leaf, []
            if parent is not None:
                parent.children.append(self)

    r = N('0', [0, 0], None, False)
    N('0-0', [1, 0], r, True); N('0-1', [-1, 0], r, True); N('0-2', [5, 0], r, True)
    z = np.asarray([[0.9, 0], [2, 0], [-0.5, 0], [-3, 0], [0.2, 0]])
    p2c, pred = class_vote(r, z, [0, 0, 1, 1, 0], np.asarray([[4.0, 0], [-2, 0]]), 0.0)
    assert list(p2c) == [0, 1, 0] and list(pred) == [0, 1]
    p2c, _ = class_vote(r, z, [0, 0, 1, 1, 0], z[:1], 1.0)        # decay 1: own votes + all 5 votes of the root:
    assert list(p2c) == [0, 1, 0]                                 # (6,2), (3,4), (3,2)


def test_trajectory_config5_shape():
    """BASELINE config-5 shape at fixture size (D = 128, num_splits = 4, splits and prunes in every epoch):
    the oracle against what the real reference did (tests/golden/traj_config5_shape.npz)."""
    from oracle.vi_oracle import drifting_latents
    g = load("traj_config5_shape.npz")
    m = meta(g)
    fpv = list(g['fpv'])
    N = int(g['N'])
    vi = OracleVarInf(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    ks = []
    for ep in range(m['epochs']):
        Z = drifting_latents(N, m['D'], m['ncomp'], m['seed'], ep, m['spread'])
        want = sub(g, 'e%d_' % ep)
        assert float(np.sum(Z.astype(np.float64))) == float(want['z_checksum'])
        np.random.seed(13 + ep)
        vi.update_variational_parameters(make_latent_dict(Z, fpv))
        got = snapshot(vi)
        assert list(got['ids']) == list(want['ids'])
        np.testing.assert_array_equal(got['is_leaf'], want['is_leaf'])
        np.testing.assert_allclose(got['alpha'], want['alpha'], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(got['sigma'], want['sigma'], rtol=TOL, atol=TOL)
        np.testing.assert_allclose(got['gamma'], want['gamma'], rtol=TOL, atol=TOL, equal_nan=True)
        np.testing.assert_allclose(got['phi'].sum(axis=0), want['phi_colsum'], rtol=1e-11)
        np.testing.assert_allclose(got['phi'][:: max(1, N // 64)], want['phi_rows'], rtol=0, atol=TOL)
        ks.append(len(got['ids']))
    assert len(set(ks)) > 2 and max(ks) > 85          # the tree both grew past its initial 85 nodes and shrank


def _offline_tree(g):
    from oracle.vi_oracle import tree_from_arrays
    return tree_from_arrays(g['tree_ids'], g['tree_parent'], g['tree_is_leaf'], g['tree_alpha'])


@pytest.mark.parametrize("tag,decay,sparse", [("vote0_", 0.0, False), ("vote5_", 0.5, False),
                                              ("vote0s_", 0.0, True), ("vote5s_", 0.5, True)])
def test_class_vote_oracle_against_reference_tool(tag, decay, sparse):
    """oracle/hierarchy_oracle.class_vote against the reference's compute_acc_frame_avg_dynamic.main run
    unmodified (tests/golden/offline_tools.npz): assignments, vote table, path -> class, predictions and
    the printed precision.  The sparse cases leave leaves without a vote (nearest-voted-leaf fallback)."""
    from oracle.hierarchy_oracle import class_vote, precision_percent
    g = load("offline_tools.npz")
    nodes = _offline_tree(g)
    n = int(g['n_sparse']) if sparse else len(g['z_train'])
    p2c, pred = class_vote(nodes[0], g['z_train'][:n], g['train_labels'][:n], g['z_test'], decay)
    assert class_vote.last['leaf_ids'] == list(g[tag + 'leaf_ids'])
    np.testing.assert_array_equal(class_vote.last['path_assignments'], g[tag + 'path_assignments'])
    np.testing.assert_array_equal(class_vote.last['counts_total'], g[tag + 'counts_total'])
    np.testing.assert_array_equal(p2c, g[tag + 'path2class'])
    np.testing.assert_array_equal(pred, g[tag + 'predicted'])
    assert precision_percent(pred, g['test_labels']) == float(g[tag + 'precision'])
    if sparse and decay == 0.0:
        assert (g[tag + 'counts_total'].max(axis=1) == 0).any()


@pytest.mark.parametrize("K", [1, 4])
def test_closest_frames_oracle_against_reference_tool(K):
    """oracle/hierarchy_oracle.closest_frames against create_hierarchy.main run unmodified: the frames
    it lists per node (its printed lines), ties included (duplicated training frames)."""
    from oracle.hierarchy_oracle import closest_frames
    g = load("offline_tools.npz")
    nodes = _offline_tree(g)
    assign, top = closest_frames(nodes[0], g['z_train'], K)
    np.testing.assert_array_equal(assign, g['top%d_assign' % K])
    by_id = {n.node_id: n for n in nodes}
    assert sorted(by_id) == sorted(g['top%d_ids' % K])
    for nid, frames in zip(g['top%d_ids' % K], g['top%d_frames' % K]):
        assert top[by_id[str(nid)]] == [int(f) for f in frames if f >= 0], nid
