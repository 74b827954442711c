"""GPU parity tests: the HIP path (through the C ABI, via the drop-in VarInf / DeviceVI)
against the CPU oracle and the committed golden vectors (outputs of the real reference).

Tolerances are BASELINE.json's: node parameters rel 1e-4, log-responsibilities abs 1e-4,
hard assignments equal except at near-ties (|dlogp| < 1e-5).
"""
import os

import numpy as np
import pytest

from oracle.vi_oracle import (OracleVarInf, ONode, snapshot, restore, make_latent_dict,
                              gaussian_mixture, structure)

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAM_RTOL = 1e-4
LOGP_ATOL = 1e-4
PHI_ATOL = 1e-4


def load(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def meta(d):
    import json
    return json.loads(str(d['meta']))


def rows_for(fpv, N):
    lc = make_latent_dict(np.zeros((N, 1), np.float32), list(fpv))
    vids = list(lc.keys())
    return vids, [(v, f) for v in vids for f in lc[v].keys()]


def check_gated_s2(S0, S2, want_S2, N):
    """S2 = sum phi ||z - alpha||^2 against the oracle.  Trees of more than 100 leaves carry it only for the
    leaves behind the reference's gate mean(phi) >= 1e-2 (src/var_inf.py:217), NaN elsewhere."""
    S0, S2, want_S2 = np.asarray(S0), np.asarray(S2), np.asarray(want_S2)
    if len(S0) > 100:
        gated = S0 / float(N) >= 1e-2
        near = np.abs(S0 / float(N) - 1e-2) < 1e-6          # a mass on the gate may fall either way in fp32
        assert np.array_equal(np.isnan(S2)[~near], ~gated[~near])
        ok = ~np.isnan(S2)
        np.testing.assert_allclose(S2[ok], want_S2[ok], rtol=1e-4, atol=1e-3)
    else:
        np.testing.assert_allclose(S2, want_S2, rtol=1e-4, atol=1e-3)


def close_params(got, want, what):
    """MAX-RELATIVE closeness: every element within PARAM_RTOL of the array's LARGEST magnitude (atol =
    1e-4 * max|want|), not of its own.  That is the right yardstick for alpha, whose entries straddle zero
    (an fp32 sum of O(1) terms cannot hold an entry of 1e-6 to four digits); sigma, which is >= 1 everywhere,
    is additionally held to 1e-4 of each element by close_sigma."""
    scale = max(1.0, float(np.max(np.abs(want))))
    np.testing.assert_allclose(got, want, rtol=PARAM_RTOL, atol=PARAM_RTOL * scale, err_msg=what)


def close_sigma(got, want, what):
    """elementwise relative 1e-4, no absolute slack (sigmasqr_inv >= 1: src/var_inf.py:282-297)"""
    np.testing.assert_allclose(got, want, rtol=PARAM_RTOL, atol=0.0, err_msg=what)


def assert_snap_close(got, want, phi_atol=PHI_ATOL):
    assert list(got['ids']) == list(want['ids'])
    np.testing.assert_array_equal(got['parent'], want['parent'])
    np.testing.assert_array_equal(got['is_leaf'], want['is_leaf'])
    close_params(got['alpha'], want['alpha'], 'alpha')
    close_sigma(got['sigma'], want['sigma'], 'sigma')
    np.testing.assert_allclose(got['gamma'], want['gamma'], rtol=PARAM_RTOL, atol=1e-3, equal_nan=True)
    np.testing.assert_allclose(got['phi'], want['phi'], rtol=0, atol=phi_atol)


def make_vi(**kw):
    from bnp_vae_amd import VarInf
    return VarInf(**kw)


def product_from_snapshot(snap, D, thr, ns, fpv):
    N = int(sum(fpv))
    vids, pairs = rows_for(fpv, N)
    vi = make_vi(latent_code_size=D, stdev_thr=thr, num_splits=ns)
    vi.vidid_to_idx, vi.vidid_frameid_to_idx = list(vids), list(pairs)
    restore(vi, snap, lambda nid, par: vi._new_node(nid, par), vids, pairs)
    return vi


def oracle_from_snapshot(snap, D, thr, ns, fpv):
    N, V = int(sum(fpv)), len(fpv)
    vids, pairs = rows_for(fpv, N)
    vi = OracleVarInf(latent_code_size=D, stdev_thr=thr, num_splits=ns)
    restore(vi, snap, lambda nid, par: ONode(nid, V, N, par, D, 10.0, rng=np.random.RandomState(0)),
            vids, pairs)
    return vi


# --------------------------------------------------------------------------------------------
def test_library_loads_on_gpu():
    from bnp_vae_amd import _lib
    assert _lib.lib.bnpvi_abi_version() == 1


@pytest.mark.parametrize("name", ["phases_irregular.npz", "phases_ref_shape.npz"])
def test_phases_against_golden(name):
    g = load(name)
    m = meta(g)
    fpv = list(g['fpv'])
    Z = g['Z']
    vi = product_from_snapshot(sub(g, 'in_'), m['D'], m['thr'], m['num_splits'], fpv)
    assert_snap_close(snapshot(vi), sub(g, 'in_'))

    vi.compute_sigma_alpha(Z)                                    # M-step
    s = snapshot(vi)
    close_params(s['alpha'], g['m_alpha'], 'alpha after M-step')
    close_sigma(s['sigma'], g['m_sigma'], 'sigma after M-step')

    vi.compute_gamma()                                           # gamma-step
    np.testing.assert_allclose(snapshot(vi)['gamma'], g['g_gamma'], rtol=PARAM_RTOL, atol=1e-4,
                               equal_nan=True)

    # prior tables (device digamma + prefix sums) from the device gamma
    dev = vi._dev
    vi._ensure_tree(); vi._sync_gamma()
    on, before = dev.prior_table(want_tables=True)
    np.testing.assert_allclose(on.cpu().numpy().T, g['on'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(before.cpu().numpy().T, g['before'], rtol=0, atol=2e-5)

    logphi, assign = vi.compute_phi(Z, want_logphi=True, want_assign=True)   # E-step
    L = g['logits'].shape[1]
    lp = logphi.cpu().numpy()[:, :L]
    np.testing.assert_allclose(lp, g['log_phi'], rtol=0, atol=LOGP_ATOL)
    assert_snap_close(snapshot(vi), sub(g, 'e_'))
    # hard assignments: equal except at near-ties
    want = np.argmax(g['logits'], axis=1)
    got = assign.cpu().numpy()
    bad = np.nonzero(got != want)[0]
    for i in bad:
        assert abs(g['logits'][i, got[i]] - g['logits'][i, want[i]]) < 1e-5

    # split diagnostics and decisions
    S0, S2 = vi._leaf_moments()
    N = Z.shape[0]
    np.testing.assert_allclose(S0 / N, g['leaf_mean_phi'], rtol=1e-4, atol=1e-7)
    heavy = g['leaf_mean_phi'] >= 1e-2
    np.testing.assert_allclose(np.sqrt(np.sqrt(S2[heavy]) / S0[heavy]), g['leaf_stdev'][heavy], rtol=1e-4)
    np.random.seed(0)
    ret = vi.split_nodes(vi.root, Z, float(g['split_thr']))
    assert bool(ret) == bool(g['split_ret'])
    assert_snap_close(snapshot(vi), sub(g, 's_'))
    vi.compute_phi(Z)
    vi.merge_nodes(vi.root, Z, float(g['split_thr']))
    assert_snap_close(snapshot(vi), sub(g, 'x_'))


@pytest.mark.parametrize("name", ["traj_ref_shape.npz", "traj_dynamic.npz", "traj_single_video.npz"])
def test_trajectory_against_golden(name):
    """Several epochs of update_variational_parameters with a new Z each (drop-in flow)."""
    g = load(name)
    m = meta(g)
    vi = make_vi(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    for ep in range(m['epochs']):
        lc = make_latent_dict(g['Z%d' % ep], list(g['fpv']))
        np.random.seed(13 + ep)
        vi.update_variational_parameters(lc)
        want = sub(g, 'e%d_' % ep)
        got = snapshot(vi)
        assert_snap_close(got, want, phi_atol=2e-4)
        assert float(got['decay']) == float(want['decay'])


def test_weighted_kmeans_golden():
    from bnp_vae_amd import VarInf
    g = load("kmeans.npz")
    for i in range(3):
        c = VarInf.weighted_kmeans(g['data%d' % i], g['w%d' % i], int(g['K%d' % i]))
        np.testing.assert_allclose(c, g['centers%d' % i], rtol=0, atol=0)


def _regular_case(N, D, bf, levels, fpv, seed):
    """Oracle + product on the same regular tree, one-hot init, same Z."""
    Z = gaussian_mixture(N, D, bf ** levels, seed=seed, spread=4.0)
    lc = make_latent_dict(Z, fpv)
    np.random.seed(seed)
    ora = OracleVarInf(latent_code_size=D, branching=bf, levels=levels, stdev_thr=1e9)
    Zm = ora.get_matrix_from_dict(lc)
    ora.build_initial_tree(); ora.build_video_index(); ora.one_hot_init(Zm)
    np.random.seed(seed)
    vi = make_vi(latent_code_size=D, branching=bf, levels=levels, stdev_thr=1e9)
    Zp = vi.get_matrix_from_dict(lc)
    vi._build_initial_tree(); vi._one_hot_init(Zp)
    return ora, vi, Zm


@pytest.mark.parametrize("N,D,bf,levels,fpv", [
    (6000, 32, 5, 2, [256]),          # BASELINE config 1 shape (K=31), scaled N
    (20000, 64, 8, 2, [256]),         # BASELINE config 2 shape (K=73, L=64)
    (3000, 64, 8, 3, [100, 57]),      # BASELINE config 3 shape (K=585, L=512): streamed regime
    (2500, 40, 3, 2, [300]),          # D not a kernel width (padded 40 -> 64), L=9
    (4000, 128, 4, 3, [500]),         # reference tree shape, D=128
    (1500, 200, 7, 3, [300, 77]),     # L=343, D=200 -> 256: streamed regime, 128 KiB operand chunks staged in LDS
    (1501, 128, 4, 5, [256, 100]),    # L=1024, D=128: workgroup-wide M-step tiles (mstep_wide.hip), ragged N
    (700, 130, 6, 4, [512]),          # L=1296 (5.06 leaf blocks of 256), D=130 -> 256: partial last leaf block
    (1800, 128, 8, 3, [256, 61]),     # L=512, D=128: one 512-leaf x 128-dim workgroup tile (mstep_wide.hip)
    (900, 256, 23, 2, [128]),         # L=529 (2.07 blocks of 256), D=256
    (512, 256, 6, 5, [100, 57, 200]), # BASELINE config 4's exact tree (K=9331, L=7776, D=256), videos of unequal length
])
def test_iteration_against_oracle(N, D, bf, levels, fpv):
    ora, vi, Z = _regular_case(N, D, bf, levels, fpv, seed=7)
    # one-hot init: identical except near-ties of the two nearest leaves
    po, pp = snapshot(ora)['phi'], snapshot(vi)['phi']
    mism = np.nonzero(np.argmax(po, 1) != np.argmax(pp, 1))[0]
    assert len(mism) <= max(1, N // 2000), "one-hot init differs on %d frames" % len(mism)
    if len(mism):
        # continue from identical state
        for leaf, col in zip(vi._dfs_leaves(), po.T):
            leaf.phi = col.reshape(-1, 1)
    for it in range(2):
        ora.compute_sigma_alpha(Z); vi.compute_sigma_alpha(Z)
        so, sp = snapshot(ora), snapshot(vi)
        close_params(sp['alpha'], so['alpha'], 'alpha it%d' % it)
        close_sigma(sp['sigma'], so['sigma'], 'sigma it%d' % it)
        ora.compute_gamma(); vi.compute_gamma()
        np.testing.assert_allclose(snapshot(vi)['gamma'], snapshot(ora)['gamma'], rtol=1e-4, atol=1e-4)
        ora.compute_phi(Z)
        logphi, assign = vi.compute_phi(Z, want_logphi=True, want_assign=True)
        L = ora.last_log_phi.shape[1]
        np.testing.assert_allclose(logphi.cpu().numpy()[:, :L], ora.last_log_phi, rtol=0, atol=LOGP_ATOL)
        np.testing.assert_allclose(snapshot(vi)['phi'], snapshot(ora)['phi'], rtol=0, atol=PHI_ATOL)
        got, want = assign.cpu().numpy(), np.argmax(ora.last_logits, axis=1)
        for i in np.nonzero(got != want)[0]:
            assert abs(ora.last_logits[i, got[i]] - ora.last_logits[i, want[i]]) < 1e-5
        # size-independent properties
        phi = snapshot(vi)['phi']
        np.testing.assert_allclose(phi.sum(axis=1), 1.0, atol=1e-5)
        S0, _ = vi._leaf_moments()
        np.testing.assert_allclose(S0.sum(), N, rtol=1e-6)


@pytest.mark.parametrize("N,D,bf,levels,fpv", [
    (3000, 64, 8, 3, [100, 57]),      # BASELINE config 3 shape (L=512): two-pass row kernel
    (2000, 32, 5, 3, [256]),          # L=125 (padded 128)
    (1500, 128, 6, 3, [300, 41]),     # L=216, D=128: leaf operands just fit the LDS
    (2100, 64, 9, 2, [64]),           # L=81 (padded 96): last 64-leaf chunk half empty
    (2300, 64, 16, 2, [128, 77]),     # L=256: single-pass variant with 4 chunks, all full
    (1900, 48, 6, 3, [90]),           # L=216 (padded 224), D=48 -> 64: 4 chunks, last one ragged
    (1300, 200, 7, 3, [256, 33]),     # L=343, D=200 -> 256: operands exceed the LDS -> streamed regime,
                                      # column-window normalise pass (no arg-max asked)
])
def test_row_kernel_against_oracle(N, D, bf, levels, fpv):
    """Trees with more than 64 leaves take the row kernel (estep_rows.h) when neither hard mode nor
    the arg-max is asked for -- two passes, or one pass with register-resident logits when no
    statistics are wanted and L <= 256, D <= 64: phi, log phi and the split/merge statistics."""
    ora, vi, Z = _regular_case(N, D, bf, levels, fpv, seed=9)
    for leaf, col in zip(vi._dfs_leaves(), snapshot(ora)['phi'].T):
        leaf.phi = col.reshape(-1, 1)
    for it in range(2):
        ora.compute_sigma_alpha(Z); vi.compute_sigma_alpha(Z)
        ora.compute_gamma(); vi.compute_gamma()
        ora.compute_phi(Z)
        logphi, _ = vi.compute_phi(Z, want_logphi=True)
        L = ora.last_log_phi.shape[1]
        np.testing.assert_allclose(logphi.cpu().numpy()[:, :L], ora.last_log_phi, rtol=0, atol=LOGP_ATOL)
        po, pp = snapshot(ora)['phi'], snapshot(vi)['phi']
        np.testing.assert_allclose(pp, po, rtol=0, atol=PHI_ATOL)
        np.testing.assert_allclose(pp.sum(axis=1), 1.0, atol=1e-5)
        S0, S2 = vi._leaf_moments()
        d2 = ((Z[:, None, :].astype(np.float64) - np.asarray([l.alpha for l in ora.leaves()])[None]) ** 2).sum(-1)
        np.testing.assert_allclose(S0, po.sum(0), rtol=1e-4, atol=1e-4)
        check_gated_s2(S0, S2, (po * d2).sum(0), len(Z))
    # without log phi (the production call: the one-pass kernel for 101 .. 512 leaves, estep_rows1.h) the same
    # phi must come out, to the rounding of two different summation orders
    vi.compute_phi(Z)
    np.testing.assert_allclose(snapshot(vi)['phi'], pp, rtol=0, atol=3e-6)
    # and without statistics
    dev = vi._dev
    two_pass = dev.phi.clone()
    dev.estep(want_stats=False)
    import torch
    torch.cuda.synchronize()
    assert float((dev.phi - two_pass).abs().max()) <= 3e-6
    logphi2, _ = dev.estep(want_stats=False, want_logphi=True)
    np.testing.assert_allclose(logphi2.cpu().numpy()[:, :L], ora.last_log_phi, rtol=0, atol=LOGP_ATOL)


@pytest.mark.parametrize("bf,levels", [(8, 2), (8, 3)])
def test_fused_iteration_matches_separate_calls(bf, levels):
    """DeviceVI.iteration (one mid-step call: node parameters + gamma + prior table + leaf
    operands, small trees in ONE launch) against the same iteration made of the separate entry
    points, and a second iteration on top (the mass buffer it leaves zeroed, operand reuse)."""
    import torch
    N, D = 5000, 64
    ora, vi, Z = _regular_case(N, D, bf, levels, [256], seed=21)
    for leaf, col in zip(vi._dfs_leaves(), snapshot(ora)['phi'].T):
        leaf.phi = col.reshape(-1, 1)
    vi._load_latents(Z); vi._sync_phi(); vi._sync_gamma()
    dev = vi._dev
    state = [t.clone() for t in (dev.phi, dev.alpha, dev.sigma, dev.gamma)]
    outs = []
    for fused in (False, True):
        dev.phi, dev.alpha, dev.sigma, dev.gamma = [t.clone() for t in state]
        for _ in range(2):
            if fused:
                dev.iteration(want_estats=True)
            else:
                dev.leaf_stats(want_q=False, want_mass=True)
                dev.node_params()
                dev.gamma_update()
                dev.prior_table()
                dev.estep(want_stats=True)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (dev.phi, dev.alpha, dev.sigma, dev.gamma, dev.prior, dev.estats)])
    for a, b, name in zip(outs[0], outs[1], ("phi", "alpha", "sigma", "gamma", "prior", "estats")):
        fin = torch.isfinite(a)
        assert torch.equal(fin, torch.isfinite(b)), name
        np.testing.assert_allclose(b[fin].cpu().numpy(), a[fin].cpu().numpy(), rtol=1e-6, atol=1e-6, err_msg=name)


def test_frames_coinciding_with_leaf_means():
    """Right after a split the new leaf means ARE data rows (weighted_kmeans): d = 0 exactly.
    The fp32 expansion cancels catastrophically there; the kernel must fall back to direct
    differences (kRefineRel path)."""
    N, D, bf, levels = 2048, 64, 8, 2
    ora, vi, Z = _regular_case(N, D, bf, levels, [256], seed=11)
    rows = np.arange(0, 64) * 7
    for vobj, leaves in ((ora, ora.leaves()), (vi, vi._dfs_leaves())):
        for r, leaf in zip(rows, leaves):
            leaf.alpha = Z[r].astype(np.float64)
    ora.compute_phi(Z)
    logphi, _ = vi.compute_phi(Z, want_logphi=True)
    np.testing.assert_allclose(logphi.cpu().numpy()[:, :64], ora.last_log_phi, rtol=0, atol=LOGP_ATOL)


def test_frames_coinciding_with_leaf_means_row_kernel():
    """Same cancellation case through the two-pass row kernel (L = 512)."""
    N, D, bf, levels = 2048, 64, 8, 3
    ora, vi, Z = _regular_case(N, D, bf, levels, [256], seed=11)
    rows = np.arange(0, 512) * 3
    for vobj, leaves in ((ora, ora.leaves()), (vi, vi._dfs_leaves())):
        for r, leaf in zip(rows, leaves):
            leaf.alpha = Z[r].astype(np.float64)
    ora.compute_phi(Z)
    logphi, _ = vi.compute_phi(Z, want_logphi=True)
    np.testing.assert_allclose(logphi.cpu().numpy()[:, :512], ora.last_log_phi, rtol=0, atol=LOGP_ATOL)


def test_ragged_and_tiny_inputs():
    """N not a multiple of the 32-frame tile, single-frame videos, a 2-frame data set."""
    for N, fpv in ((33, [1, 2, 30]), (2, [1, 1]), (65, [64, 1])):
        ora, vi, Z = _regular_case(N, 8, 2, 2, fpv, seed=3)
        for leaf, col in zip(vi._dfs_leaves(), snapshot(ora)['phi'].T):
            leaf.phi = col.reshape(-1, 1)
        ora.compute_sigma_alpha(Z); vi.compute_sigma_alpha(Z)
        ora.compute_gamma(); vi.compute_gamma()
        ora.compute_phi(Z); vi.compute_phi(Z)
        assert_snap_close(snapshot(vi), snapshot(ora))


def test_pickle_and_writers(tmp_path):
    import pickle
    g = load("traj_dynamic.npz")
    m = meta(g)
    vi = make_vi(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    np.random.seed(13)
    vi.update_variational_parameters(make_latent_dict(g['Z0'], list(g['fpv'])))
    vi.write_nodes(str(tmp_path / "nodes.pkl"))
    vi.write_alpha(str(tmp_path / "alpha.txt"))
    vi.write_sigma(str(tmp_path / "sigma.txt"))
    with open(str(tmp_path / "nodes.pkl"), "rb") as f:
        nodes = pickle.load(f)
    assert [n.node_id for n in nodes] == list(g['e0_ids'])
    root = [n for n in nodes if n.parent is None][0]
    assert root.node_id == '0' and not root.isLeafNode
    leaves = [n for n in nodes if n.isLeafNode]
    np.testing.assert_allclose(np.concatenate([l.phi for l in leaves], axis=1), g['e0_phi'], atol=2e-4)
    lines = (tmp_path / "alpha.txt").read_text().splitlines()
    assert [l.split(' ')[0] for l in lines] == list(g['e0_ids'])
    assert vi.get_true_path_mean([('a', 'b')] * 5).shape == (5, m['D'])


def test_mstep_bit_reproducible_behind_the_estep():
    """E-step then M-step, 400 times on fixed parameters (config-3 shape): phi is rewritten with
    the same values each time, so every M-step result must equal the first bit for bit.  Guards
    the S0 accumulation of mstep_dma.hip, which once lost k-steps in ~1 of 1000 launches."""
    import torch
    from bench import CONFIGS, regular_tree, make_latents
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    _n, D, bf, levels = CONFIGS["config3"]
    N = 262_144
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    dev = DeviceVI(device=0)
    Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    for _ in range(3):
        dev.iteration(want_estats=True)
    first, differ = None, 0
    for _ in range(400):
        dev.estep(want_stats=True)
        dev.leaf_stats(want_q=False, want_mass=True)
        if first is None:
            first = dev.stats.clone()
        else:
            differ += int(not torch.equal(dev.stats, first))
    col = dev.phi[:, :L].sum(dim=0, dtype=torch.float64)
    assert differ == 0
    np.testing.assert_allclose(first[:, 0].cpu().numpy(), col.cpu().numpy(), rtol=2e-6, atol=1e-4)
    dev.close()


def test_closest_frames_per_node_against_oracle():
    """bnp_vae_amd.hierarchy.closest_frames (hard E-step + path-distance kernel + stable device
    sort) against the restatement of src/create_hierarchy.py:40-53,82-91 on a tree grown by
    split / merge, with duplicated frames so that the stable tie order matters."""
    from bnp_vae_amd.hierarchy import closest_frames
    from oracle.hierarchy_oracle import closest_frames as oracle_closest, dfs_leaves
    g = load("traj_dynamic.npz")
    m = meta(g)
    vi = make_vi(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    np.random.seed(13)
    for e in range(3):
        vi.update_variational_parameters(make_latent_dict(g['Z%d' % e], list(g['fpv'])))
    nodes = vi.get_nodes_list(vi.root, [])
    Z = np.asarray(g['Z2'], dtype=np.float32)
    z = np.concatenate([Z, Z[:40], Z[5:25]]).astype(np.float32)       # exact duplicates: distance ties
    for K in (1, 4, 1000):
        assign, top = closest_frames(nodes, z, K)
        want_assign, want_top = oracle_closest(vi.root, z.astype(np.float64), K)
        d = np.linalg.norm(np.asarray([l.alpha for l in dfs_leaves(vi.root)])[None] - z[:, None].astype(np.float64), axis=2)
        for i in np.nonzero(assign != want_assign)[0]:                  # near-ties of the two nearest leaves only
            assert abs(d[i, assign[i]] - d[i, want_assign[i]]) < 1e-5
        if np.array_equal(assign, want_assign):
            assert set(id(n) for n in top) == set(id(n) for n in want_top)
            for n in want_top:
                assert top[n] == want_top[n], n.node_id
    assert sum(len(v) for v in top.values()) >= len(z)                  # K=1000: every frame under its leaf


@pytest.mark.parametrize("decay", [0.0, 0.5])
def test_class_vote_against_oracle(decay):
    """hierarchy.class_vote (nearest-leaf passes and vote histogram on the device) against the
    restatement of src/compute_acc_frame_avg_dynamic.py:41-129, on a split/merged tree; with
    decay 0 some leaves get no vote and borrow a neighbour's class."""
    from bnp_vae_amd.hierarchy import class_vote
    from oracle.hierarchy_oracle import class_vote as oracle_vote
    g = load("traj_dynamic.npz")
    m = meta(g)
    vi = make_vi(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
    np.random.seed(13)
    for e in range(3):
        vi.update_variational_parameters(make_latent_dict(g['Z%d' % e], list(g['fpv'])))
    nodes = vi.get_nodes_list(vi.root, [])
    z_train = np.asarray(g['Z2'], dtype=np.float32)[:60]                # few frames: some leaves stay empty
    z_test = np.asarray(g['Z3'], dtype=np.float32)
    labels = np.random.RandomState(3).randint(0, 4, size=len(z_train))
    p2c, pred = class_vote(nodes, z_train, labels, z_test, decay_factor=decay)
    want_p2c, want_pred = oracle_vote(vi.root, z_train.astype(np.float64), labels, z_test.astype(np.float64), decay)
    np.testing.assert_array_equal(p2c, want_p2c)
    # a prediction may differ only where the frame's two nearest leaves are an fp32 near-tie: the class
    # the product chose must belong to a leaf no further than 1e-5 (relative) beyond the nearest one
    from oracle.hierarchy_oracle import dfs_leaves
    alpha = np.asarray([l.alpha for l in dfs_leaves(vi.root)], dtype=np.float64)
    for i in np.nonzero(pred != want_pred)[0]:
        d = np.linalg.norm(alpha - z_test[i].astype(np.float64), axis=1)
        tied = d - d.min() <= 1e-5 * max(1.0, d.min())
        assert pred[i] in set(p2c[tied]), (i, d.min(), np.sort(d)[:3])
    assert (pred != want_pred).mean() <= 0.01


@pytest.mark.parametrize("cfg,N", [("config2", 262_144), ("config3", 65_536)])
def test_estep_bit_reproducible(cfg, N):
    """300 E-steps on fixed parameters: phi must come out bit-identical every time (fused kernel
    for config 2, two-pass row kernel for config 3), the statistics to fp32-atomic-order accuracy."""
    import torch
    from bench import CONFIGS, regular_tree, make_latents
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    _n, D, bf, levels = CONFIGS[cfg]
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    dev = DeviceVI(device=0)
    Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    for _ in range(2):
        dev.iteration(want_estats=True)
    first, est0, differ = None, None, 0
    for _ in range(300):
        dev.estep(want_stats=True)
        if first is None:
            first, est0 = dev.phi.clone(), dev.estats.clone()
        else:
            differ += int(not torch.equal(dev.phi, first))
            assert float(((dev.estats - est0).abs() / (est0.abs() + 1.0)).max()) < 1e-12
    assert differ == 0
    dev.close()


def test_matrix_entry_point_matches_dict_entry_point():
    """update_variational_parameters_matrix (NumPy matrix, device tensor) walks the same
    trajectory -- splits and merges included -- as the reference-shaped dict entry point."""
    import torch
    g = load("traj_dynamic.npz")
    m = meta(g)
    fpv = list(g['fpv'])
    epochs = [g["Z%d" % e] for e in range(4)]
    runs = []
    for kind in ("dict", "numpy", "cuda"):
        vi = make_vi(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'])
        np.random.seed(13)
        snaps = []
        for Z in epochs:
            lc = make_latent_dict(Z, fpv)
            if kind == "dict":
                vi.update_variational_parameters(lc)
            else:
                keys = [(v, f) for v in lc for f in lc[v]]
                rows = np.asarray([lc[v][f] for v, f in keys])
                if kind == "cuda":
                    rows = torch.from_numpy(rows).to("cuda:0")
                vi.update_variational_parameters_matrix(rows, keys=keys)
            snaps.append(snapshot(vi))
        runs.append(snaps)
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert list(a['ids']) == list(b['ids'])
            np.testing.assert_array_equal(a['alpha'], b['alpha'])
            np.testing.assert_array_equal(a['phi'], b['phi'])
            np.testing.assert_array_equal(a['gamma'], b['gamma'])
    vi = make_vi(latent_code_size=m['D'])
    with pytest.raises(ValueError):
        vi.update_variational_parameters_matrix(epochs[0])                      # no keys on first call
    with pytest.raises(ValueError):
        vi.update_variational_parameters_matrix(epochs[0][:3], keys=[(0, 0), (1, 0), (0, 1)])


def test_path_prior_branch_of_get_true_path_mean():
    """src/var_inf.py:58-68 (dead behind a `continue` in the reference, optional here): same
    draws as the oracle from the same global NumPy seed, on the same responsibilities."""
    g = load("traj_dynamic.npz")
    m = meta(g)
    fpv = list(g['fpv'])
    lc = make_latent_dict(g['Z0'], fpv)
    vi = make_vi(latent_code_size=m['D'], stdev_thr=m['thr'], num_splits=m['num_splits'], path_prior=True)
    assert np.all(vi.get_true_path_mean([('x', 0)] * 3) == 0)          # no paths initiated (:58-60)
    np.random.seed(13)
    vi.update_variational_parameters(lc)
    ora = oracle_from_snapshot(snapshot(vi), m['D'], m['thr'], m['num_splits'], fpv)
    pairs = list(vi.vidid_frameid_to_idx)
    rs = np.random.RandomState(5)
    batch = [pairs[i] for i in rs.randint(0, len(pairs), size=64)]
    np.random.seed(99)
    got = vi.get_true_path_mean(batch)
    np.random.seed(99)
    want = ora.get_true_path_mean(batch, path_prior=True)
    # the device phi is fp32: a draw can differ only where the uniform variate falls within
    # rounding of a cumulative-probability boundary
    same = np.asarray(vi.last_path_indices) == np.asarray(ora.last_path_indices)
    assert same.mean() >= 0.95
    # ... and every differing draw IS such a boundary case: np.random.choice draws one uniform variate per
    # call and takes cdf.searchsorted(u, 'right'); replay the variates and look at the boundaries between
    # the two picks in the oracle's fp64 cdf
    np.random.seed(99)
    us = [np.random.random_sample() for _ in batch]
    bfs, queue = [], [ora.root]
    while queue:
        n, queue = queue[0], queue[1:]
        if n.isLeafNode:
            bfs.append(n)
        else:
            queue = queue + n.children
    for b in np.nonzero(~same)[0]:
        idx = ora.vidid_frameid_to_idx.index(batch[b])
        p = np.asarray([leaf.phi[idx][0] for leaf in bfs], dtype=np.float64)
        cdf = np.cumsum(p / p.sum())
        lo, hi = sorted((vi.last_path_indices[b], ora.last_path_indices[b]))
        assert np.all(np.abs(cdf[lo:hi] - us[b]) < 1e-5), (b, us[b], cdf[lo:hi])
    np.testing.assert_allclose(got[same], want[same], rtol=PARAM_RTOL, atol=1e-6)
    assert got.shape == (64, m['D']) and len(set(vi.last_path_indices)) > 1
    plain = make_vi(latent_code_size=m['D'])
    assert np.all(plain.get_true_path_mean(batch) == 0)                # the reference's live path


# --------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: no oracle run is affordable there, so the checks are size-independent
# properties of one VI iteration (and of its parts) on the synthetic benchmark workload.
@pytest.mark.parametrize("cfg,N", [("config2", 999_936), ("config3", 262_144), ("config4", 32_768)])
def test_full_size_properties(cfg, N):
    import torch
    from bench import CONFIGS, regular_tree, make_latents
    from bnp_vae_amd import _lib
    from bnp_vae_amd.engine import DeviceVI
    _n, D, bf, levels = CONFIGS[cfg]
    parent, is_leaf = regular_tree(bf, levels)
    K, L = len(parent), int(is_leaf.sum())
    dev = DeviceVI(device=0)
    Z, vid, V = make_latents(torch, dev.device, N, D, L, 0)
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, np.random.RandomState(13).normal(size=(K, D)), np.ones(K))
    dev.default_gamma()
    dev.estep(mode=_lib.ESTEP_HARD, want_stats=False)
    phi0 = dev.phi[:, :L]
    assert torch.equal(phi0.sum(dim=1), torch.ones(N, device=dev.device))           # one-hot rows
    assert torch.all((phi0 == 0) | (phi0 == 1))
    # nearest leaf indeed (spot check against a torch fp32 distance on a slice)
    leaves = torch.from_numpy(dev.tree.node_of_leaf.astype(np.int64)).to(dev.device)
    A = dev.alpha[leaves].to(torch.float32)
    sl = slice(0, 20000)
    d2 = torch.cdist(dev.Z[sl, :D], A) ** 2
    pick = phi0[sl].argmax(dim=1)
    best = d2.min(dim=1).values
    assert torch.all(d2.gather(1, pick[:, None])[:, 0] <= best * (1 + 1e-5) + 1e-4)

    for it in range(3):
        dev.iteration(want_estats=True)
    torch.cuda.synchronize()
    phi = dev.phi[:, :L]
    rows = phi.sum(dim=1, dtype=torch.float64)
    assert float((rows - 1).abs().max()) < 1e-5                                      # softmax rows
    assert float(phi.min()) >= 0.0 and bool(torch.all(dev.phi[:, L:] == 0))          # padding stays 0
    S0e = dev.estats[:, 0]
    np.testing.assert_allclose(float(S0e.sum()), N, rtol=1e-6)                       # total mass
    col = phi.sum(dim=0, dtype=torch.float64)
    np.testing.assert_allclose(S0e.cpu().numpy(), col.cpu().numpy(), rtol=2e-5, atol=1e-3)

    # M-step statistics of that phi: S0 = column sums, per-video masses add up to S0, S1 against a
    # torch fp64 product on a slice, linearity over a split of the frames
    dev.leaf_stats(want_q=False, want_mass=True)
    torch.cuda.synchronize()
    st, mass = dev.stats.clone(), dev.mass.clone()
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), col.cpu().numpy(), rtol=2e-6, atol=1e-4)
    # (both are fp64 sums of fp32 partials, grouped differently: per video vs per frame chunk)
    np.testing.assert_allclose(mass.sum(dim=0).cpu().numpy(), st[:, 0].cpu().numpy(), rtol=2e-6, atol=1e-4)
    ref = phi.to(torch.float64).T @ dev.Z[:, :D].to(torch.float64)
    scale = float(ref.abs().max())
    assert float((st[:, 1:1 + D] - ref).abs().max()) <= 2e-6 * scale
    half = (N // 512) * 256
    import ctypes as C

    def part(lo, hi):
        out = torch.empty_like(st)
        _lib.check(_lib.lib.bnpvi_leaf_stats(
            dev.ctx, C.c_void_p(dev.Z.data_ptr() + 4 * lo * dev.Dp), dev.Dp,
            C.c_void_p(dev.phi.data_ptr() + 4 * lo * dev.tree.Lp), dev.tree.Lp,
            C.c_void_p(dev.vid.data_ptr() + 4 * lo), hi - lo, D, L, dev.V, 0,
            C.c_void_p(out.data_ptr()), C.c_void_p(0), dev._stream()))
        return out
    both = part(0, half) + part(half, N)
    torch.cuda.synchronize()
    # fp32 partials over <=8192-frame chunks fall on different chunk boundaries in the split run
    assert float((both - st).abs().max()) <= 1e-6 * max(1.0, float(st.abs().max()))
    # bit-reproducible run to run (fixed-order reductions)
    dev.leaf_stats(want_q=False, want_mass=False)
    torch.cuda.synchronize()
    assert torch.equal(dev.stats, st)
    dev.close()


@pytest.mark.parametrize("N,D,L,fpv", [
    (40_000, 128, 1024, 100),      # 512-leaf x 128-dim workgroup tiles, video boundaries inside k-steps
    (24_577, 200, 1500, 77),       # 256 x 256 tiles, D and L padded, ragged last k-step
])
def test_wide_mstep_against_fp64(N, D, L, fpv):
    """mstep_wide.hip on long chunks (hundreds of k-steps through the 3-slot ring) with video
    boundaries that fall inside k-steps: S0, S1 and the per-video masses against torch fp64."""
    import torch
    from bnp_vae_amd.engine import DeviceVI
    parent = np.concatenate([[-1], np.zeros(L, np.int32)]).astype(np.int32)      # root + L leaves
    is_leaf = np.concatenate([[0], np.ones(L, np.uint8)]).astype(np.uint8)
    dev = DeviceVI(device=0)
    g = torch.Generator(device=dev.device)
    g.manual_seed(5)
    Z = torch.randn((N, D), generator=g, device=dev.device) * 3.0
    vid = (torch.arange(N, device=dev.device) // fpv).to(torch.int32)
    V = int(vid[-1]) + 1
    dev.set_latents(Z, vid, V)
    dev.set_tree(parent, is_leaf, np.zeros((L + 1, D)), np.ones(L + 1))
    logits = torch.randn((N, L), generator=g, device=dev.device) * 4.0
    dev.alloc_phi()
    dev.phi[:, :L] = torch.softmax(logits, dim=1)
    del logits
    assert dev.tree.Lp >= 1024 and dev.Dp % 128 == 0                              # the wide kernel's regime
    dev.leaf_stats(want_q=False, want_mass=True)
    torch.cuda.synchronize()
    phi64 = dev.phi[:, :L].to(torch.float64)
    s0 = phi64.sum(dim=0)
    s1 = phi64.T @ Z.to(torch.float64)
    mass = torch.zeros((V, L), dtype=torch.float64, device=dev.device).index_add_(0, vid.long(), phi64)
    st = dev.stats
    np.testing.assert_allclose(st[:, 0].cpu().numpy(), s0.cpu().numpy(), rtol=2e-6, atol=1e-6)
    assert float((st[:, 1:1 + D] - s1).abs().max()) <= 2e-6 * float(s1.abs().max())
    np.testing.assert_allclose(dev.mass[:, :L].cpu().numpy(), mass.cpu().numpy(), rtol=2e-6, atol=1e-7)
    first, first_mass = st.clone(), dev.mass.clone()
    dev.leaf_stats(want_q=False, want_mass=False)
    torch.cuda.synchronize()
    assert torch.equal(dev.stats, first)                                          # fixed-order reductions
    # run to run: statistics bit-identical over many launches (S0 and S1 come from fixed-order sums;
    # the masses are fp64 atomics of per-wave partials, at most a few per entry, so only close)
    for _ in range(200):
        dev.leaf_stats(want_q=False, want_mass=True)
        assert torch.equal(dev.stats, first)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dev.mass.cpu().numpy(), first_mass.cpu().numpy(), rtol=1e-12, atol=1e-12)
    dev.close()


def test_phi_beyond_32bit_element_indices():
    """Config-3 shape with 8.5M frames: phi has 4.35e9 elements, so every row offset past the middle
    needs 64-bit arithmetic (row kernel, wide M-step, slab reductions).  Checks in tools/bign_check.py:
    row sums at both ends and the middle, total mass, S1 of the last 1M frames against torch fp64."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "bign_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bign_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main("config3", 8_500_224)
